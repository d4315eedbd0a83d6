// augcl_common.h -- shared device/host helpers for libaugcl_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/augcl.h"

namespace augcl {

// ---- error plumbing (thread-local message, no exceptions across the ABI) ----
extern thread_local char g_err[512];
int fail(int code, const char *fmt, ...);
int check_launch(const char *what);

// Fixed-order two-level reduction of per-workgroup gradient slabs (backward-weights).  A launcher hands its
// descriptor to reduce_slabs(): run at once (2 launches), or -- while the whole-stack backward has opened a
// deferral scope on this thread -- queued, so that ALL layers are reduced by one batched pair of launches.
struct ReduceDesc {
    const float *partial; // [nslab][n]
    float *part2;         // [8][n] scratch
    float *out_w, *out_b; // out_w[0..n_w), out_b[0..n-n_w)
    int nslab, n, n_w;
};
constexpr int kReduceMaxBatch = 16;
int reduce_slabs(const ReduceDesc &d, hipStream_t s, const char *what);
void reduce_defer_begin();
int reduce_defer_flush(hipStream_t s, const char *what); // ends the scope
void reduce_defer_abort();                               // ends the scope, drops what was queued

// Strips per image for the 32-channel conv kernels: the launchers take the fewest strips that fit LDS, split evenly.
// For a few of the encoder's layer widths one to four MORE strips measured 5-14 % faster (tools/tune_strips.py, B = 256
// and 512 on 256 CUs: the 32-pixel tiles of a strip are dealt to 4 compute waves, and some heights waste most of a
// round); everything else keeps the default.  kind: 0 f32 forward, 1 f32 backward-data, 2 bf16 forward, 3 bf16
// backward-data; W = width of the kernel's INPUT tensor.  (tools/tune_strips.py re-measures the table with a library
// built with -DAUGCL_TUNING, which is the only build that reads an override from the environment.)
inline int extra_strips(int kind, int W)
{
#ifdef AUGCL_TUNING
    if (const char *e = getenv("AUGCL_EXTRA_STRIPS")) return atoi(e);
#endif
    switch (kind) {
    case 0: return W == 41 ? 4 : W == 37 ? 2 : 0;
    case 1: return W == 39 ? 4 : W == 33 ? 2 : W == 23 ? 4 : 0; // re-measured for the zero-padded dZ tile (pitch W + 2)
    case 2: return W == 41 ? 2 : W == 39 ? 4 : 0;
    case 3: return W == 39 ? 2 : W == 33 ? 2 : 0;
    }
    return 0;
}

// Persistent workgroup i of G takes items v, v+G, v+2G, ... with v = xcd_slot(i, G): workgroups are dealt round-robin
// over the 8 XCDs (one L2 each), so giving XCD x the CONTIGUOUS slots [x*G/8, (x+1)*G/8) makes neighbouring items --
// the strips of one image, which re-read each other's two halo rows -- run at the same time on one XCD, where the
// second read hits in L2 (PMC: the halo rows were fetched from HBM twice, +10 % traffic on the bf16 layers).
__device__ __forceinline__ int xcd_slot(int i, int G) { return (G & 7) == 0 ? (i & 7) * (G >> 3) + (i >> 3) : i; }

// Streaming outputs (written once, read by a LATER kernel at the earliest): non-temporal stores keep them from
// displacing the input stream in L2 / Infinity Cache (random_conv: 28.9 -> 27.0 us per B=256 launch).
typedef float nt_f4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void store_nt(float *dst, float4 v)
{
    __builtin_nontemporal_store(nt_f4{v.x, v.y, v.z, v.w}, reinterpret_cast<nt_f4 *>(dst));
}
__device__ __forceinline__ void store_nt(uint32_t *dst, uint32_t v) { __builtin_nontemporal_store(v, dst); }

// first layer fast path (encoder_first.hip): AUGCL_OK / negative error after launching, +1 if the shape is not covered
// w2 / bias2 != null: two-weight-set forward (B images = two halves over the SAME B/2 inputs); amax_out (f32 output only):
// max |out| is atomically max-ed into it (one float, two per half when dual; zeroed by the caller); gate_bits_out (f32 output
// with ReLU only): one word per output pixel (B,Ho,Wo), bit c = (channel c > 0)
int conv_first9_try(const float *in, const float *w, const float *bias, void *out, int B, int C, int Hi, int Wi, int stride,
                    float scale, int relu, bool out_bf16, hipStream_t s, const char *name, const float *w2 = nullptr,
                    const float *bias2 = nullptr, float *amax_out = nullptr, void *gate_bits_out = nullptr);

// ... and the float32-input, float32-output case on the bf16 matrix pipe (encoder_first_x3.hip: three exact bf16 pieces per operand,
// six piece products; 9 x 84 x 84, stride 2): what conv_first9_try launches for that case; same return convention
int conv_first9_x3_try(const float *in, const float *w, const float *bias, float *out, int B, int C, int Hi, int Wi, int stride,
                       float scale, int relu, hipStream_t s, const char *name, const float *w2 = nullptr, const float *bias2 = nullptr,
                       float *amax_out = nullptr, void *gate_bits_out = nullptr);
// the same layer fed with uint8 observations (encoder_first_u8.hip; 9 x 84 x 84, stride 2, f32 NCHW output only): AUGCL_OK /
// negative error after launching, +1 if the shape is not covered.  w2 != null: both weight sets over the SAME B inputs, out =
// (2B,...) with images [0,B) from set 1 and [B,2B) from set 2; amax_out: one float per set
int conv_first9_u8_try(const uint8_t *in, const float *w, const float *bias, float *out, int B, int C, int Hi, int Wi, int stride,
                       float scale, int relu, hipStream_t s, const char *name, const float *w2 = nullptr, const float *bias2 = nullptr,
                       float *amax_out = nullptr, void *gate_bits_out = nullptr);

#define AUGCL_REQUIRE(cond, ...)                                 \
    do {                                                         \
        if (!(cond)) return augcl::fail(AUGCL_E_INVALID, __VA_ARGS__); \
    } while (0)

static inline hipStream_t as_stream(augcl_stream_t s) { return reinterpret_cast<hipStream_t>(s); }
static inline bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }
static inline bool aligned4(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 3u) == 0; }

// ---- Philox4x32-10 (Salmon et al. 2011), counter-based: no state, any lane can
// compute any sample's parameters, so per-sample params are sharding-invariant.
struct u32x4 {
    uint32_t x, y, z, w;
};

__host__ __device__ inline u32x4 philox4x32_10(u32x4 ctr, uint32_t k0, uint32_t k1)
{
    const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        uint64_t p0 = (uint64_t)M0 * ctr.x;
        uint64_t p1 = (uint64_t)M1 * ctr.z;
        u32x4 n;
        n.x = (uint32_t)(p1 >> 32) ^ ctr.y ^ k0;
        n.y = (uint32_t)p1;
        n.z = (uint32_t)(p0 >> 32) ^ ctr.w ^ k1;
        n.w = (uint32_t)p0;
        ctr = n;
        k0 += W0;
        k1 += W1;
    }
    return ctr;
}

// counter layout: (global sample index, slot, offset_lo, offset_hi); key = seed
__host__ __device__ inline u32x4 philox_sample(uint64_t seed, uint64_t offset, uint32_t sample, uint32_t slot)
{
    u32x4 c{sample, slot, (uint32_t)offset, (uint32_t)(offset >> 32)};
    return philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32));
}

// slot 0: crop offsets, uniform over {0..2*pad} by multiply-high (no modulo bias worth the name: 2^32/9)
__host__ __device__ inline void philox_shift_offsets(uint64_t seed, uint64_t offset, uint32_t sample, int pad,
                                                     int &oy, int &ox)
{
    u32x4 r = philox_sample(seed, offset, sample, 0u);
    const uint32_t n = (uint32_t)(2 * pad + 1);
    oy = (int)(((uint64_t)r.x * n) >> 32);
    ox = (int)(((uint64_t)r.y * n) >> 32);
}

// Box-Muller on two 24-bit uniforms in (0,1), on the transcendental unit: v_log_f32 (log2), v_sqrt_f32 and
// v_sin_f32 / v_cos_f32, whose argument is in REVOLUTIONS -- exactly the 2*pi*u2 Box-Muller wants, no range reduction.
// ~8 instructions instead of ~200 for logf + sincosf: inside random_conv the 21 lanes that draw a sample's weights sit
// on the workgroup's critical path (the strip cannot be convolved before they are done).  ~1e-6 absolute accuracy --
// these are random draws, defined by this function (the exported streams use it too), not a parity-bound quantity.
__device__ inline void box_muller(uint32_t a, uint32_t b, float &n0, float &n1)
{
    const float u1 = ((float)(a >> 8) + 0.5f) * 5.9604644775390625e-8f; // 2^-24
    const float u2 = ((float)(b >> 8) + 0.5f) * 5.9604644775390625e-8f;
    const float rad = __builtin_amdgcn_sqrtf(-1.3862943611198906f * __builtin_amdgcn_logf(u1)); // sqrt(-2 ln u1)
    n0 = rad * __builtin_amdgcn_cosf(u2);
    n1 = rad * __builtin_amdgcn_sinf(u2);
}

// slots 1..21: the 81 N(0,1) conv weights of a sample, 4 per Philox call (j = 0..20 -> weights 4j..4j+3)
__device__ inline void philox_conv_weights4(uint64_t seed, uint64_t offset, uint32_t sample, int j, float w[4])
{
    u32x4 r = philox_sample(seed, offset, sample, 1u + (uint32_t)j);
    box_muller(r.x, r.y, w[0], w[1]);
    box_muller(r.z, r.w, w[2], w[3]);
}

} // namespace augcl
