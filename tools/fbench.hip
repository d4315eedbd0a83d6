// tools/fbench.hip -- developer microbenchmark for the FIRST-layer encoder kernels (not part of the product).
// usage: fbench B [lds_kb_fwd per_cu_fwd lds_kb_wg per_cu_wg]
#include "../augcl_amd/csrc/encoder_kernels.hip"
#include <algorithm>
#include <functional>
#include <string>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)
using namespace augcl;
struct V { std::string name; double bytes; std::function<void()> run; };

template <bool BF>
static void fwd(const float *in, const float *w, const float *bias, float *out, int B, int lds_kb, int per_cu)
{
    Conv1Args a{}; a.in = in; a.w = w; a.bias = bias; a.out = out; a.B = B; a.C = 9; a.Hi = 84; a.Wi = 84; a.stride = 2; a.scale = 255.f; a.relu = 1;
    a.Ho = 41; a.Wo = 41;
    int rt = (lds_kb * 1024 / 4 / (10 * 84) - 3) / 2 + 1; if (rt > 41) rt = 41;
    a.S = (41 + rt - 1) / rt; a.RT = (41 + a.S - 1) / a.S; a.n_items = B * a.S;
    const size_t lds = (size_t)10 * (2 * (a.RT - 1) + 3) * 84 * 4;
    set_lds(k_conv_first<5, BF>, lds, "x");
    hipLaunchKernelGGL((k_conv_first<5, BF>), dim3(grid_for(a.n_items, per_cu)), dim3(256), lds, 0, a);
}
template <bool BF>
static void wg(const float *x, const float *dz, float *ws, int B, int lds_kb, int per_cu)
{
    Wgrad1Args a{}; a.x = x; a.dz = dz; a.partial = ws; a.B = B; a.C = 9; a.Hi = 84; a.Wi = 84; a.stride = 2; a.scale = 255.f; a.Ho = 41; a.Wo = 41;
    int rt = (lds_kb * 1024 / 4 - 3 * 9 * 84 - 64) / (9 * 2 * 84 + 32 * 41); if (rt > 41) rt = 41;
    a.S = (41 + rt - 1) / rt; a.RT = (41 + a.S - 1) / a.S; a.n_items = B * a.S;
    const int planeX = (2 * (a.RT - 1) + 3) * 84, planeZ = (a.RT * 41) | 1;
    size_t lds = (size_t)(9 * planeX + 32 * planeZ) * 4; const size_t red = (size_t)((3 * 16 + 1) * 64) * 4; if (lds < red) lds = red;
    set_lds(k_wgrad_first<3, BF>, lds, "x");
    hipLaunchKernelGGL((k_wgrad_first<3, BF>), dim3(grid_for(a.n_items, per_cu)), dim3(256), lds, 0, a);
}
int main(int argc, char **argv)
{
    const int B = argc > 1 ? atoi(argv[1]) : 256;
    const size_t nin = (size_t)B * 9 * 84 * 84, nout = (size_t)B * 32 * 41 * 41;
    float *in, *out, *w, *bias, *ws, *dw, *db;
    CK(hipMalloc(&in, nin * 4)); CK(hipMalloc(&out, nout * 4)); CK(hipMalloc(&w, 2592 * 4)); CK(hipMalloc(&bias, 128));
    CK(hipMalloc(&dw, 2592 * 4)); CK(hipMalloc(&db, 128));
    const size_t wsb = augcl_enc_wgrad_workspace_bytes(9) * 4; CK(hipMalloc(&ws, wsb));
    std::vector<float> h(std::max(nin, nout)); for (size_t i = 0; i < h.size(); ++i) h[i] = (float)((i * 2654435761u >> 9) & 255);
    CK(hipMemcpy(in, h.data(), nin * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(out, h.data(), nout * 4, hipMemcpyHostToDevice));
    for (auto &v : h) v = v / 512.f - 0.2f;
    CK(hipMemcpy(w, h.data(), 2592 * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(bias, h.data(), 128, hipMemcpyHostToDevice));
    const double b_in = (double)nin * 4, b_f32 = (double)nout * 4, b_bf = (double)nout * 2;
    std::vector<V> vs;
    vs.push_back({"fwd f32 product", b_in + b_f32, [&] { augcl_enc_conv_first_forward(in, w, bias, out, B, 9, 84, 84, 2, 255.f, 1, 0); }});
    vs.push_back({"fwd bf16 product", b_in + b_bf, [&] { augcl_enc_bf16_conv_first_forward(in, w, bias, out, B, 9, 84, 84, 2, 255.f, 1, 0); }});
    vs.push_back({"wgrad f32 product", b_in + b_f32, [&] { augcl_enc_conv_first_backward_weights(in, out, dw, db, ws, wsb, B, 9, 84, 84, 2, 255.f, 0); }});
    vs.push_back({"wgrad bf16 product", b_in + b_bf, [&] { augcl_enc_bf16_conv_first_backward_weights(in, out, dw, db, ws, wsb, B, 9, 84, 84, 2, 255.f, 0); }});
    const int cfg_f[][2] = {{76, 2}, {50, 3}, {37, 4}, {30, 5}};
    for (auto &c : cfg_f) {
        const int kb = c[0], pc = c[1];
        vs.push_back({"fwd f32 lds" + std::to_string(kb) + " x" + std::to_string(pc), b_in + b_f32, [=] { fwd<false>(in, w, bias, out, B, kb, pc); }});
        vs.push_back({"fwd bf16 lds" + std::to_string(kb) + " x" + std::to_string(pc), b_in + b_bf, [=] { fwd<true>(in, w, bias, out, B, kb, pc); }});
    }
    const int cfg_w[][2] = {{150, 1}, {76, 2}, {50, 3}, {37, 4}};
    for (auto &c : cfg_w) {
        const int kb = c[0], pc = c[1];
        vs.push_back({"wgrad f32 lds" + std::to_string(kb) + " x" + std::to_string(pc), b_in + b_f32, [=] { wg<false>(in, out, ws, B, kb, pc); }});
        vs.push_back({"wgrad bf16 lds" + std::to_string(kb) + " x" + std::to_string(pc), b_in + b_bf, [=] { wg<true>(in, out, ws, B, kb, pc); }});
    }
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    printf("B=%d\n%-28s %9s %9s\n", B, "variant", "med us", "GB/s");
    std::vector<std::vector<double>> us(vs.size());
    for (int r = 0; r < 5; ++r) for (size_t v = 0; v < vs.size(); ++v) {
        for (int i = 0; i < 3; ++i) vs[v].run();
        CK(hipEventRecord(e0, 0)); for (int i = 0; i < 20; ++i) vs[v].run(); CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1)); CK(hipGetLastError());
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); us[v].push_back(ms * 1000.0 / 20);
    }
    for (size_t v = 0; v < vs.size(); ++v) { std::sort(us[v].begin(), us[v].end()); const double m = us[v][2]; printf("%-28s %9.1f %9.1f\n", vs[v].name.c_str(), m, vs[v].bytes / m / 1e3); }
    return 0;
}
