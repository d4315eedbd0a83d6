// encoder_x2_common.h -- types, constants and the f16-pair split shared by the f16-pair conv kernels (encoder_x2.hip: one
// layer per launch; encoder_chain.hip: several layers per launch with the intermediates resident in LDS).
#pragma once
#include "augcl_common.h"

namespace augcl {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t vu4 __attribute__((ext_vector_type(4)));
typedef float vf4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

constexpr int X2_SLOT = 144;       // LDS bytes per pixel slot: 2 planes x 64 B + 16 B pad (36 p mod 64 dwords: 16 consecutive
                                   // pixels x one 16-byte chunk fall into 16 distinct bank quads)
constexpr int X2_NU = 2;           // loader rounds per strip: 256 lanes x X2_NU units of (4 pixels x 8 channels)
constexpr int X2_ROWPAD = 224;     // forward: extra bytes per LDS tile row (14 chunks of 16 B: see k_conv32_x2's row pitch)
constexpr int X2_HDR = 512;        // LDS in front of the tiles: +-bias (2 x 128 B), the maximum's two words, spare
constexpr int X2_MAXTILE = 81664;  // bytes per tile buffer: 2 x this + X2_HDR <= 160 KiB
constexpr int X2_FRAG_VU4 = 2 * 18 * 64; // 16-byte entries per layer block, then one entry: word 0 = the weights' exponent, word 1 =
                                         // max_co sum_{ci,t} |w| (float bits), word 2 = max_ci sum_{co,t} |w|
constexpr int X2_BLOCK_BYTES = 55296;    // stride of the per-layer blocks in frag_ws (= the split-bf16 kernels' block: one size function)

__device__ __forceinline__ int acc_row_h(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

// k with amax 2^k in [2^14, 2^15) (amax = 1.m 2^(e - 127)); zero / tiny / huge / non-finite maxima clamp to k in [-46, 74]
__host__ __device__ __forceinline__ int scale_exp(uint32_t amax_bits)
{
    int e = (int)((amax_bits >> 23) & 0xffu);
    e = e < 67 ? 67 : e > 187 ? 187 : e;
    return 141 - e;
}
__device__ __forceinline__ float pow2f(int k) { return __uint_as_float((uint32_t)(127 + k) << 23); } // -126 <= k <= 127

// two floats (already scaled) -> their f16 (hi, lo') pairs: hi + lo' 2^-11 == the float to 2^-24
__device__ __forceinline__ void split2h(float a, float b, uint32_t &hi, uint32_t &lo)
{
    const f16x2 h = __builtin_convertvector(f32x2{a, b}, f16x2); // RNE
    const float ra = (a - (float)h[0]) * 2048.0f, rb = (b - (float)h[1]) * 2048.0f; // residuals are exact; 2^11 lifts them to f16's range
    const f16x2 l = __builtin_convertvector(f32x2{ra, rb}, f16x2);
    hi = __builtin_bit_cast(uint32_t, h);
    lo = __builtin_bit_cast(uint32_t, l);
}
// ... with the residual left unscaled (backward-weights: all three products share one accumulator; elements below 2^-16 of
// the tensor's maximum keep an ABSOLUTE error of 2^-40 of the maximum, which is what a sum over every pixel needs)
__device__ __forceinline__ void split2u(float a, float b, uint32_t &hi, uint32_t &lo)
{
    const f16x2 h = __builtin_convertvector(f32x2{a, b}, f16x2);
    const f16x2 l = __builtin_convertvector(f32x2{a - (float)h[0], b - (float)h[1]}, f16x2);
    hi = __builtin_bit_cast(uint32_t, h);
    lo = __builtin_bit_cast(uint32_t, l);
}

__device__ __forceinline__ float wave_max(float v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

} // namespace augcl
