// tools/vbench.hip -- can packed-f32 VALU FMAs and f32 MFMAs run at peak side by side on gfx950?  (design probe, not product)
// build: hipcc -O3 --offload-arch=gfx950 -std=c++17 tools/vbench.hip -o tools/vbench
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

// mode 0: every wave packed FMAs; 1: every wave MFMA; 2: even waves MFMA, odd waves packed FMA; 3: every wave scalar v_fma
template <int MODE>
__global__ __launch_bounds__(512) void k(const float *__restrict__ w, float *out, int iters)
{
    const int wave = threadIdx.x >> 6;
    const bool do_mfma = MODE == 1 || (MODE == 2 && (wave & 1) == 0);
    float r = 0.f;
    if (do_mfma) {
        f32x16 acc0, acc1;
        for (int i = 0; i < 16; ++i) { acc0[i] = 0.f; acc1[i] = 0.f; }
        float a = threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-4f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(b, a, acc1, 0, 0, 0);
            }
        }
        for (int i = 0; i < 16; ++i) r += acc0[i] + acc1[i];
    } else if (MODE == 3) {
        float acc[64];
        for (int i = 0; i < 64; ++i) acc[i] = 0.f;
        const float x = threadIdx.x * 1e-3f;
        for (int it = 0; it < iters; ++it) {
            const float *wp = w + (it & 7) * 64;
#pragma unroll
            for (int c = 0; c < 64; ++c) acc[c] = __builtin_fmaf(wp[c], x, acc[c]);
        }
        for (int i = 0; i < 64; ++i) r += acc[i];
    } else {
        f32x2 acc[32];
        for (int i = 0; i < 32; ++i) acc[i] = f32x2{0.f, 0.f};
        const f32x2 x = f32x2{threadIdx.x * 1e-3f, 1.0f};
        for (int it = 0; it < iters; ++it) {
            const float *wp = w + (it & 7) * 32; // uniform address: scalar loads, SGPR splat operands
#pragma unroll
            for (int c = 0; c < 32; ++c) {
                const float ws = wp[c];
                acc[c] = __builtin_elementwise_fma(f32x2{ws, ws}, x, acc[c]);
            }
        }
        for (int i = 0; i < 32; ++i) r += acc[i].x + acc[i].y;
    }
    if (r == 123.456f) out[0] = r;
}

template <int MODE>
static int run(const char *name, const float *w, float *out, int waves_per_wg, double flop_mfma_waves, double flop_valu_waves)
{
    const int iters = 2000, grid = 256 * 2;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int rep = 0; rep < 2; ++rep) {
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(64 * waves_per_wg), 0, 0, w, out, iters);
        CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    }
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double waves = (double)grid * waves_per_wg;
    const double fl = iters * waves * (flop_mfma_waves + flop_valu_waves);
    printf("%-46s %8.3f ms  %7.1f TFLOP/s\n", name, ms, fl / ms / 1e9);
    return 0;
}

int main()
{
    float *w, *out; CK(hipMalloc(&w, 4096)); CK(hipMalloc(&out, 64)); CK(hipMemset(w, 0, 4096));
    const double MF = 32.0 * 4096, PK = 32.0 * 64 * 2 * 2, SC = 64.0 * 64 * 2; // flop per wave per iteration
    if (run<0>("packed v_pk_fma_f32, 8 waves/WG, 2 WG/CU", w, out, 8, 0, PK)) return 1;
    if (run<3>("scalar v_fma_f32,   8 waves/WG, 2 WG/CU", w, out, 8, 0, SC)) return 1;
    if (run<1>("MFMA 32x32x2 f32,   8 waves/WG, 2 WG/CU", w, out, 8, MF, 0)) return 1;
    if (run<1>("MFMA 32x32x2 f32,   4 waves/WG, 2 WG/CU", w, out, 4, MF, 0)) return 1;
    // mixed: half the waves MFMA, half packed FMA; FLOP counted for both halves
    {
        const int iters = 2000, grid = 512;
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        for (int rep = 0; rep < 2; ++rep) {
            CK(hipEventRecord(e0, 0));
            hipLaunchKernelGGL(k<2>, dim3(grid), dim3(512), 0, 0, w, out, iters);
            CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
        }
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        const double fl = (double)iters * grid * 4 * (MF + PK);
        printf("%-46s %8.3f ms  %7.1f TFLOP/s (MFMA part %.1f, packed part %.1f)\n", "mixed: 4 MFMA + 4 packed waves per WG", ms, fl / ms / 1e9,
               (double)iters * grid * 4 * MF / ms / 1e9, (double)iters * grid * 4 * PK / ms / 1e9);
    }
    return 0;
}
