// tools/bbench.hip -- developer microbenchmark for the bf16 encoder kernels (not part of the product).
// usage: bbench B H
#include "../augcl_amd/csrc/encoder_bf16.hip"
#include <functional>
#include <string>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)
struct V { std::string name; double bytes; std::function<void()> run; };
int main(int argc, char **argv)
{
    const int B = argc > 1 ? atoi(argv[1]) : 512;
    const int H = argc > 2 ? atoi(argv[2]) : 41;
    const size_t nbig = (size_t)B * 32 * (H + 2) * (H + 2);
    void *in, *out, *mask; float *w, *bias, *ws, *dw, *db;
    CK(hipMalloc(&in, nbig * 2)); CK(hipMalloc(&out, nbig * 4)); CK(hipMalloc(&mask, nbig * 2));
    CK(hipMalloc(&w, 9216 * 4)); CK(hipMalloc(&bias, 128)); CK(hipMalloc(&dw, 9216 * 4)); CK(hipMalloc(&db, 128));
    const size_t wsb = augcl_enc_wgrad_workspace_bytes(32); CK(hipMalloc(&ws, wsb));
    std::vector<uint16_t> h(nbig); for (size_t i = 0; i < nbig; ++i) h[i] = (uint16_t)(0x3c00 + ((i * 2654435761u >> 9) & 0x1ff) + ((i & 4) ? 0x8000 : 0));
    CK(hipMemcpy(in, h.data(), nbig * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(mask, h.data(), nbig * 2, hipMemcpyHostToDevice));
    std::vector<float> hw(9216); for (int i = 0; i < 9216; ++i) hw[i] = ((i * 37) % 101) / 300.f - 0.15f;
    CK(hipMemcpy(w, hw.data(), 9216 * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(bias, hw.data(), 128, hipMemcpyHostToDevice));
    const double pin = (double)B * H * H * 64, pout = (double)B * (H - 2) * (H - 2) * 64;
    std::vector<V> vs;
    vs.push_back({"fwd bf16", pin + pout, [&] { augcl_enc_bf16_conv32_forward(in, w, bias, out, B, H, H, 1, 0, 0); }});
    vs.push_back({"bwd-data bf16 (dz HxH -> H+2)", pin + 2.0 * B * (H + 2) * (H + 2) * 64, [&] { augcl_enc_bf16_conv32_backward_data(in, w, mask, out, B, H, H, 0); }});
    vs.push_back({"wgrad bf16 mfma", pin + pout, [&] { augcl_enc_bf16_conv32_backward_weights_mfma(in, out, dw, db, ws, wsb, B, H, H, 0); }});
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    printf("B=%d H=%d\n%-32s %9s %9s\n", B, H, "variant", "med us", "GB/s");
    std::vector<std::vector<double>> us(vs.size());
    for (int r = 0; r < 5; ++r) for (size_t v = 0; v < vs.size(); ++v) {
        for (int i = 0; i < 3; ++i) vs[v].run();
        CK(hipEventRecord(e0, 0)); for (int i = 0; i < 20; ++i) vs[v].run(); CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1)); CK(hipGetLastError());
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); us[v].push_back(ms * 1000.0 / 20);
    }
    for (size_t v = 0; v < vs.size(); ++v) { std::sort(us[v].begin(), us[v].end()); const double m = us[v][2]; printf("%-32s %9.1f %9.1f\n", vs[v].name.c_str(), m, vs[v].bytes / m / 1e3); }
    return 0;
}
