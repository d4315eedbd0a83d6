// Do vector-ALU instructions hide behind matrix instructions on gfx950?  One wave per SIMD runs a chain of dependent
// v_mfma_f32_32x32x16_f16 (32 cycles each) with K independent v_fma_f32 between two of them (K = 0..16, four accumulating
// registers in turn: no dependency stall); second arm: the matrix wave runs bare and ANOTHER wave on the same SIMD issues the
// vector instructions (the conv kernels' compute / loader split).  Cycles per matrix instruction from s_memtime of wave 0.
// build: hipcc -O3 --offload-arch=gfx950 tools/mfma_valu_probe.hip -o tools/mfma_valu_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

#define VFMA(X) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(X) : "v"(m), "v"(c))

template <int K>
__device__ __forceinline__ void valu_block(float &x0, float &x1, float &x2, float &x3, float m, float c)
{
#pragma unroll
    for (int k = 0; k < K; ++k) {
        if ((k & 3) == 0) VFMA(x0);
        else if ((k & 3) == 1) VFMA(x1);
        else if ((k & 3) == 2) VFMA(x2);
        else VFMA(x3);
    }
}

// MODE 0: one wave per SIMD, K vector instructions between two matrix instructions of the SAME wave.
// MODE 1: two waves per SIMD; waves 0-3 run the bare matrix chain, waves 4-7 run K vector instructions per 32 cycles of it.
template <int K, int MODE>
__global__ __launch_bounds__(512, 1) void k_probe(const uint4 *ops, float *out, int iters, unsigned long long *clk)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    f16x8 a = __builtin_bit_cast(f16x8, ops[lane]), b = __builtin_bit_cast(f16x8, ops[64 + lane]);
    float x0 = lane, x1 = lane + 1, x2 = lane + 2, x3 = lane + 3;
    const float m = 0.999f, c = 0.001f;
    f32x16 p;
    for (int r = 0; r < 16; ++r) p[r] = 0.f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (MODE == 0 || wave < 4) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                p = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, p, 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                if (MODE == 0) valu_block<K>(x0, x1, x2, x3, m, c);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    } else {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                valu_block<K>(x0, x1, x2, x3, m, c);
                if (K < 8) __builtin_amdgcn_s_sleep(0); // (keeps the loop from being only a branch when K is small)
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float sink = x0 + x1 + x2 + x3;
    for (int r = 0; r < 16; ++r) sink += p[r];
    out[blockIdx.x * 512 + threadIdx.x] = sink;
    if (blockIdx.x == 0 && threadIdx.x == 0) clk[0] = t1 - t0;
}

template <int K, int MODE>
static void run(const uint4 *ops, float *out, unsigned long long *clk, int iters)
{
    const int threads = MODE == 0 ? 256 : 512;
    hipLaunchKernelGGL((k_probe<K, MODE>), dim3(256), dim3(threads), 0, 0, ops, out, iters, clk);
    hipDeviceSynchronize();
    hipLaunchKernelGGL((k_probe<K, MODE>), dim3(256), dim3(threads), 0, 0, ops, out, iters, clk);
    hipDeviceSynchronize();
    unsigned long long c = 0;
    hipMemcpy(&c, clk, 8, hipMemcpyDeviceToHost);
    printf("%s  K = %2d vector instructions per matrix instruction: %.1f cycles per matrix instruction (wave 0; 32 = the pipe's rate)\n",
           MODE == 0 ? "same wave    " : "other wave   ", K, (double)c / ((double)iters * 8.0));
}

int main()
{
    uint32_t h[512];
    for (int i = 0; i < 512; ++i) h[i] = 0x3c003c00u; // f16 1.0 pairs
    uint4 *ops; float *out; unsigned long long *clk;
    hipMalloc(&ops, 2048); hipMalloc(&out, 256 * 512 * 4); hipMalloc(&clk, 16);
    hipMemcpy(ops, h, 2048, hipMemcpyHostToDevice);
    const int iters = 20000; // x 8 x 32 cycles = 5 M cycles ~ 2.5 ms per launch
    run<0, 0>(ops, out, clk, iters); run<2, 0>(ops, out, clk, iters); run<4, 0>(ops, out, clk, iters); run<6, 0>(ops, out, clk, iters);
    run<8, 0>(ops, out, clk, iters); run<12, 0>(ops, out, clk, iters); run<16, 0>(ops, out, clk, iters);
    run<0, 1>(ops, out, clk, iters); run<4, 1>(ops, out, clk, iters); run<8, 1>(ops, out, clk, iters); run<16, 1>(ops, out, clk, iters);
    return 0;
}
