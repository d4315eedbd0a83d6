// tools/kbench.hip -- developer microbenchmark for kernel VARIANTS (not part of the product).
// Build:  hipcc -O3 --offload-arch=gfx950 -std=c++17 -ffp-contract=off -Iinclude tools/kbench.hip \
//               augcl_amd/csrc/capi_core.cpp -x hip -o tools/kbench
// It #includes the product kernels and times experimental variants against them on a ring of
// batches larger than the 256 MiB Infinity Cache, interleaved rounds in ONE process.
#include "../augcl_amd/csrc/aug_kernels.hip"

#include <algorithm>
#include <functional>
#include <string>
#include <vector>

using namespace augcl;

#define CK(x)                                                                    \
    do {                                                                         \
        hipError_t e_ = (x);                                                     \
        if (e_ != hipSuccess) {                                                  \
            printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
            exit(1);                                                             \
        }                                                                        \
    } while (0)

// ---------------------------------------------------------------- reference copies
// S3: plain copy (upper bound for this access shape): same decomposition, no shift
template <int H, int W, int PARTS>
__global__ __launch_bounds__(256) void k_copy_parts(const float *__restrict__ in, float *__restrict__ out)
{
    constexpr int N4 = H * W / 4 / PARTS;
    const float4 *__restrict__ src = reinterpret_cast<const float4 *>(in) + (size_t)blockIdx.x * N4;
    float4 *__restrict__ dst = reinterpret_cast<float4 *>(out) + (size_t)blockIdx.x * N4;
#pragma unroll
    for (int j = threadIdx.x; j < N4; j += 256) dst[j] = src[j];
}

// S4: grid-stride copy with 2048 blocks (persistent-style)
__global__ __launch_bounds__(256) void k_copy_gs(const float4 *__restrict__ in, float4 *__restrict__ out, size_t n4)
{
    for (size_t j = (size_t)blockIdx.x * 256 + threadIdx.x; j < n4; j += (size_t)gridDim.x * 256) out[j] = in[j];
}

struct Variant {
    std::string name;
    double bytes;
    std::function<void(int)> launch; // arg = ring slot
};

int main(int argc, char **argv)
{
    const int B = argc > 1 ? atoi(argv[1]) : 256;
    const int RING = 8, ROUNDS = 5, ITERS = 100;
    const size_t n = (size_t)B * 9 * 84 * 84;
    std::vector<float *> fin(RING), fout(RING), fout2(2);
    std::vector<uint8_t *> uin(RING);
    std::vector<float> h(n);
    for (size_t i = 0; i < n; ++i) h[i] = (float)((i * 2654435761u >> 13) & 255);
    std::vector<uint8_t> hu(n);
    for (size_t i = 0; i < n; ++i) hu[i] = (uint8_t)h[i];
    for (int k = 0; k < RING; ++k) {
        CK(hipMalloc(&fin[k], n * 4));
        CK(hipMalloc(&fout[k], n * 4));
        CK(hipMalloc(&uin[k], n));
        CK(hipMemcpy(fin[k], h.data(), n * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(uin[k], hu.data(), n, hipMemcpyHostToDevice));
    }
    for (int k = 0; k < 2; ++k) CK(hipMalloc(&fout2[k], n * 4));
    std::vector<int32_t> hoy(B), hox(B);
    for (int b = 0; b < B; ++b) { hoy[b] = (b * 7 + 3) % 9; hox[b] = (b * 5 + 1) % 9; }
    int32_t *oy, *ox;
    CK(hipMalloc(&oy, B * 4)); CK(hipMalloc(&ox, B * 4));
    CK(hipMemcpy(oy, hoy.data(), B * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(ox, hox.data(), B * 4, hipMemcpyHostToDevice));
    std::vector<float> hw(B * 81);
    for (int i = 0; i < B * 81; ++i) hw[i] = (float)((i * 40503u >> 7) & 1023) / 512.f - 1.f;
    float *w;
    CK(hipMalloc(&w, B * 81 * 4));
    CK(hipMemcpy(w, hw.data(), B * 81 * 4, hipMemcpyHostToDevice));
    float *ov;
    CK(hipMalloc(&ov, (size_t)B * 3 * 84 * 84 * 4));
    CK(hipMemset(ov, 0, (size_t)B * 3 * 84 * 84 * 4));

    const double f2f = 2.0 * n * 4;
    std::vector<Variant> vs;
    vs.push_back({"shift_f32 product", f2f, [&](int k) { augcl_random_shift_f32(fin[k], fout[k], B, 9, 84, 84, 4, oy, ox, 0, 0, 0, 0); }});
    vs.push_back({"conv R=6 NT=128", f2f, [&](int k) { hipLaunchKernelGGL((k_conv<84, 84, 6, 128, float, false>), dim3(B * 3 * 14), dim3(128), 0, 0, fin[k], (float *)nullptr, fout[k], B * 3, 3, 0, (const int32_t *)nullptr, (const int32_t *)nullptr, w, 0, 0, 0); }});
    vs.push_back({"shift_conv_u8 R=6 NT=128", (double)n * 9, [&](int k) { hipLaunchKernelGGL((k_conv<84, 84, 6, 128, uint8_t, true>), dim3(B * 3 * 14), dim3(128), 0, 0, uin[k], fout[k], fout2[k & 1], B * 3, 3, 4, oy, ox, w, 0, 0, 0); }});
    vs.push_back({"conv ABL1 no sigmoid", f2f, [&](int k) { hipLaunchKernelGGL((k_conv<84, 84, 12, 256, float, false, 1>), dim3(B * 3 * 7), dim3(256), 0, 0, fin[k], (float *)nullptr, fout[k], B * 3, 3, 0, (const int32_t *)nullptr, (const int32_t *)nullptr, w, 0, 0, 0); }});
    vs.push_back({"conv ABL2 1/3 fma", f2f, [&](int k) { hipLaunchKernelGGL((k_conv<84, 84, 12, 256, float, false, 2>), dim3(B * 3 * 7), dim3(256), 0, 0, fin[k], (float *)nullptr, fout[k], B * 3, 3, 0, (const int32_t *)nullptr, (const int32_t *)nullptr, w, 0, 0, 0); }});
    vs.push_back({"conv ABL3 stage+copy only", f2f, [&](int k) { hipLaunchKernelGGL((k_conv<84, 84, 12, 256, float, false, 3>), dim3(B * 3 * 7), dim3(256), 0, 0, fin[k], (float *)nullptr, fout[k], B * 3, 3, 0, (const int32_t *)nullptr, (const int32_t *)nullptr, w, 0, 0, 0); }});
    {
        static JitterParams jp; jp.order[0] = 1; jp.order[1] = 2; jp.order[2] = 3; jp.order[3] = 0;
        jp.c_r = 1.2f; jp.c_inv = -0.2f; jp.s_r = 0.8f; jp.s_inv = 0.2f; jp.hue = 0.1f; jp.bound = 1.0f; jp.scale = 255.0f;
        const int n = 84 * 84;
        vs.push_back({"jitter product", f2f, [&](int k) { static int32_t ord[4] = {1, 2, 3, 0}; augcl_color_jitter_f32(fout[k], B, 3, 84, 84, nullptr, 0, ord, 1.2, 0.8, 0.1, 1.0f, 255.0f, 0); }});
        vs.push_back({"jitter reg<1024,2,4>", f2f, [&, n](int k) { hipLaunchKernelGGL((k_color_jitter_reg<1024, 2, 4>), dim3(B * 3), dim3(1024), 0, 0, fout[k], n, (const int32_t *)nullptr, 0, jp); }});
        vs.push_back({"jitter reg<1024,2,8> (spills)", f2f, [&, n](int k) { hipLaunchKernelGGL((k_color_jitter_reg<1024, 2, 8>), dim3(B * 3), dim3(1024), 0, 0, fout[k], n, (const int32_t *)nullptr, 0, jp); }});
        vs.push_back({"jitter reg<512,4,2>", f2f, [&, n](int k) { hipLaunchKernelGGL((k_color_jitter_reg<512, 4, 2>), dim3(B * 3), dim3(512), 0, 0, fout[k], n, (const int32_t *)nullptr, 0, jp); }});
        vs.push_back({"jitter reg<256,7,1>", f2f, [&, n](int k) { hipLaunchKernelGGL((k_color_jitter_reg<256, 7, 1>), dim3(B * 3), dim3(256), 0, 0, fout[k], n, (const int32_t *)nullptr, 0, jp); }});
        vs.push_back({"jitter reg<256,7,2>", f2f, [&, n](int k) { hipLaunchKernelGGL((k_color_jitter_reg<256, 7, 2>), dim3(B * 3), dim3(256), 0, 0, fout[k], n, (const int32_t *)nullptr, 0, jp); }});
    }
    vs.push_back({"copy parts=1", f2f, [&](int k) { hipLaunchKernelGGL((k_copy_parts<84, 84, 1>), dim3(B * 9), dim3(256), 0, 0, fin[k], fout[k]); }});
    vs.push_back({"copy parts=3", f2f, [&](int k) { hipLaunchKernelGGL((k_copy_parts<84, 84, 3>), dim3(B * 9 * 3), dim3(256), 0, 0, fin[k], fout[k]); }});
    vs.push_back({"copy parts=7", f2f, [&](int k) { hipLaunchKernelGGL((k_copy_parts<84, 84, 7>), dim3(B * 9 * 7), dim3(256), 0, 0, fin[k], fout[k]); }});
    vs.push_back({"copy gridstride 2048", f2f, [&](int k) { hipLaunchKernelGGL(k_copy_gs, dim3(2048), dim3(256), 0, 0, (const float4 *)fin[k], (float4 *)fout[k], n / 4); }});
    vs.push_back({"copy gridstride 4096", f2f, [&](int k) { hipLaunchKernelGGL(k_copy_gs, dim3(4096), dim3(256), 0, 0, (const float4 *)fin[k], (float4 *)fout[k], n / 4); }});
    vs.push_back({"hipMemcpyAsync d2d", f2f, [&](int k) { CK(hipMemcpyAsync(fout[k], fin[k], n * 4, hipMemcpyDeviceToDevice, 0)); }});
    vs.push_back({"conv_f32 product", f2f, [&](int k) { augcl_random_conv_f32(fin[k], fout[k], B, 3, 84, 84, w, 0, 0, 0, 0); }});
    vs.push_back({"shift_conv_u8 product", (double)n * 9, [&](int k) { augcl_shift_conv_u8_f32(uin[k], fout[k], fout2[k & 1], B, 3, 84, 84, 4, oy, ox, w, 0, 0, 0, 0); }});
    vs.push_back({"shift_u8_f32 product", (double)n * 5, [&](int k) { augcl_random_shift_u8_f32(uin[k], fout[k], B, 9, 84, 84, 4, oy, ox, 0, 0, 0, 0); }});
    vs.push_back({"overlay product", f2f + (double)B * 3 * 84 * 84 * 4, [&](int k) { augcl_random_overlay_f32(fin[k], ov, fout[k], B, 3, 84, 84, 0.5f, 0); }});
#ifdef KBENCH_EXTRA
#include KBENCH_EXTRA
#endif

    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    std::vector<std::vector<double>> us(vs.size());
    for (int r = 0; r < ROUNDS; ++r)
        for (size_t v = 0; v < vs.size(); ++v) {
            for (int i = 0; i < 10; ++i) vs[v].launch(i % RING);
            CK(hipEventRecord(e0, 0));
            for (int i = 0; i < ITERS; ++i) vs[v].launch(i % RING);
            CK(hipEventRecord(e1, 0));
            CK(hipEventSynchronize(e1));
            CK(hipGetLastError());
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            us[v].push_back(ms * 1000.0 / ITERS);
        }
    printf("B=%d  (%.1f MB in + out per launch for f32->f32)\n", B, f2f / 1e6);
    printf("%-34s %9s %9s %9s %8s\n", "variant", "min us", "med us", "GB/s(med)", "Mobs/s");
    for (size_t v = 0; v < vs.size(); ++v) {
        std::sort(us[v].begin(), us[v].end());
        const double med = us[v][us[v].size() / 2];
        printf("%-34s %9.2f %9.2f %9.0f %8.2f\n", vs[v].name.c_str(), us[v][0], med, vs[v].bytes / med / 1e3, B / med);
    }
    return 0;
}
