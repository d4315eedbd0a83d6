// Sustained bf16 MFMA throughput under the chip's power management, by instruction shape: v_mfma_f32_32x32x16_bf16 against
// v_mfma_f32_16x16x32_bf16 at equal flops (random operands from registers; 1 wave per SIMD on every CU; ~1 s per arm).
// build: hipcc -O3 --offload-arch=gfx950 tools/mfma_shape_probe.hip -o tools/mfma_shape_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

template <int SHAPE>
__global__ __launch_bounds__(256, 1) void k_loop(const uint4 *ops, float *out, int iters, unsigned long long *clk)
{
    const int lane = threadIdx.x & 63;
    bf16x8 a[6], b[6];
    for (int i = 0; i < 6; ++i) {
        a[i] = __builtin_bit_cast(bf16x8, ops[(i * 64 + lane) & 1023]);
        b[i] = __builtin_bit_cast(bf16x8, ops[((i + 6) * 64 + lane) & 1023]);
    }
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    float sink = 0.f;
    if (SHAPE == 32) {
        f32x16 p, n;
        for (int r = 0; r < 16; ++r) p[r] = n[r] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 6; ++i) p = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[i], p, 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 6; ++i) n = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[i], a[i], n, 0, 0, 0);
        }
        for (int r = 0; r < 16; ++r) sink += p[r] - n[r];
    } else if (SHAPE == 33) { // v_mfma_f32_32x32x16_f16: same shape and rate as the bf16 form, 11-bit significands (operands: the same bits)
        f32x16 p, n;
        for (int r = 0; r < 16; ++r) p[r] = n[r] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 6; ++i) p = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a[i]), __builtin_bit_cast(f16x8, b[i]), p, 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 6; ++i) n = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, b[i]), __builtin_bit_cast(f16x8, a[i]), n, 0, 0, 0);
        }
        for (int r = 0; r < 16; ++r) sink += p[r] - n[r];
    } else {
        f32x4 c[4];
        for (int j = 0; j < 4; ++j)
            for (int r = 0; r < 4; ++r) c[j][r] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 6; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) c[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[(i + j) % 6], c[j], 0, 0, 0);
        }
        for (int j = 0; j < 4; ++j)
            for (int r = 0; r < 4; ++r) sink += c[j][r];
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    out[blockIdx.x * 256 + threadIdx.x] = sink;
    if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
}

int main()
{
    std::vector<uint32_t> h(4096);
    srand(1);
    for (auto &v : h) { // random bf16 pairs in [-2, 2): sign, exponent 126..128, random mantissa
        uint32_t lo = ((rand() & 1) << 15) | ((126 + rand() % 3) << 7) | (rand() & 127);
        uint32_t hi = ((rand() & 1) << 15) | ((126 + rand() % 3) << 7) | (rand() & 127);
        v = lo | (hi << 16);
    }
    uint4 *ops; float *out; unsigned long long *clk;
    hipMalloc(&ops, 16384); hipMalloc(&out, 256 * 256 * 4); hipMalloc(&clk, 16);
    hipMemcpy(ops, h.data(), 16384, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 400000; // x 12 MFMAs of 32 cycles = 154 M cycles ~ 0.1 s; launched 10x back to back
    for (int rep = 0; rep < 2; ++rep)
        for (int shape : {32, 33, 16}) {
            for (int w = 0; w < 3; ++w) {
                if (shape == 32) hipLaunchKernelGGL(k_loop<32>, dim3(256), dim3(256), 0, 0, ops, out, iters, clk);
                else if (shape == 33) hipLaunchKernelGGL(k_loop<33>, dim3(256), dim3(256), 0, 0, ops, out, iters, clk);
                else hipLaunchKernelGGL(k_loop<16>, dim3(256), dim3(256), 0, 0, ops, out, iters, clk);
            }
            hipEventRecord(e0);
            for (int w = 0; w < 10; ++w) {
                if (shape == 32) hipLaunchKernelGGL(k_loop<32>, dim3(256), dim3(256), 0, 0, ops, out, iters, clk);
                else if (shape == 33) hipLaunchKernelGGL(k_loop<33>, dim3(256), dim3(256), 0, 0, ops, out, iters, clk);
                else hipLaunchKernelGGL(k_loop<16>, dim3(256), dim3(256), 0, 0, ops, out, iters, clk);
            }
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            unsigned long long c[2]; hipMemcpy(c, clk, 16, hipMemcpyDeviceToHost);
            const double flops = 10.0 * iters * 12.0 * 32768.0 * 1024.0;
            printf("shape %2d: %.1f ms for 10 launches  %.0f TFLOP/s  cycles/launch %llu  clock %.2f GHz\n", shape, ms, flops / ms / 1e9,
                   c[0], (double)c[0] / c[1] * 0.1);
        }
    return 0;
}
