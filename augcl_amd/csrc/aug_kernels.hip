// aug_kernels.hip -- CDNA4 (gfx950) kernels for AugCL's pixel augmentations.
//
// All four operators are HBM-bound: the design goal of every kernel
// here is ONE coalesced 16-B/lane read and ONE 16-B/lane write per element, with the
// 3x3 neighbourhoods of random_conv served from an LDS-staged row strip and the
// per-sample random parameters (crop offset, 81 conv weights) wave-uniform: they are
// derived from blockIdx only, so they live in SGPRs / are LDS-broadcast, never per lane.
//
// Compiled with -ffp-contract=off: random_overlay / color_jitter restate chains of
// separate IEEE f32 ops of the reference and must not be fused; FMAs are written
// explicitly (__builtin_fmaf) where a fused accumulate is intended (random_conv).
#include "augcl_common.h"

namespace augcl {

typedef float f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

// per-sample crop displacement (dy,dx) = (oy-pad, ox-pad): uniform per workgroup
__device__ __forceinline__ void sample_shift(const int32_t *__restrict__ oy, const int32_t *__restrict__ ox,
                                             uint64_t seed, uint64_t offset, int gsample, int b, int pad,
                                             int &dy, int &dx)
{
    int y, x;
    if (oy != nullptr) {
        y = oy[b];
        x = ox[b];
    } else {
        philox_shift_offsets(seed, offset, (uint32_t)gsample, pad, y, x);
    }
    dy = __builtin_amdgcn_readfirstlane(y - pad);
    dx = __builtin_amdgcn_readfirstlane(x - pad);
}

__device__ __forceinline__ float4 unpack_u8x4(uint32_t v)
{
    return make_float4((float)(v & 0xffu), (float)((v >> 8) & 0xffu), (float)((v >> 16) & 0xffu), (float)(v >> 24));
}

// ---- staging: 4 consecutive pixels of the (optionally shifted) source row ---------------------
// ONE load per quad: global_load_dwordx4 / global_load_dword tolerate dword / byte alignment on
// gfx950, and a wave's 64 lanes still sweep one contiguous span, so the x-shift costs no second
// load and no rotate (measured: 5.06 TB/s vs 4.69 for two aligned loads + rotate, vs 5.4 plain
// copy).  The start is clamped into the row, base = clamp(x4+dx, 0, W-4); only quads that stick out
// of the row (d = x4+dx-base != 0: the first / last quad of a row) need the per-lane
// replicate-select, out[k] = V[clamp(d+k, 0, 3)].
typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));
typedef uint32_t u32u __attribute__((aligned(1)));

template <int W, typename TIn>
struct Quad;

template <int W>
struct Quad<W, float> {
    f32x4u v;
    __device__ __forceinline__ void load(const float *__restrict__ row, int base)
    {
        v = *reinterpret_cast<const f32x4u *>(row + base);
    }

    __device__ __forceinline__ float4 get(int d) const
    {
        float4 o = make_float4(v.x, v.y, v.z, v.w);
        if (d != 0) {
            const float vv[4] = {v.x, v.y, v.z, v.w};
            float oo[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int idx = clampi(d + k, 0, 3);
                oo[k] = idx == 0 ? vv[0] : idx == 1 ? vv[1] : idx == 2 ? vv[2] : vv[3];
            }
            o = make_float4(oo[0], oo[1], oo[2], oo[3]);
        }
        return o;
    }
};

template <int W>
struct Quad<W, uint8_t> {
    uint32_t v;
    __device__ __forceinline__ void load(const uint8_t *__restrict__ row, int base)
    {
        v = *reinterpret_cast<const u32u *>(row + base);
    }

    __device__ __forceinline__ uint32_t packed(int d) const
    {
        uint32_t o = v;
        if (d != 0) {
            o = 0;
#pragma unroll
            for (int k = 0; k < 4; ++k) o |= ((v >> (8 * clampi(d + k, 0, 3))) & 0xffu) << (8 * k);
        }
        return o;
    }
    __device__ __forceinline__ float4 get(int d) const { return unpack_u8x4(packed(d)); }
};

// =========================================================================================
// random_shift: one workgroup per RP-row slab of one (sample, channel) plane, one 4-pixel quad
// per thread (RP*W/4 <= 256): 16128 short workgroups at B=256 keep every CU busy to the end of
// the launch (finer slabs measured 35 -> 26 us vs one workgroup per plane).
// =========================================================================================
template <int H, int W, int RP, typename TIn, typename TOut>
__global__ __launch_bounds__(256) void k_random_shift(const TIn *__restrict__ in, TOut *__restrict__ out, int C,
                                                      int pad, const int32_t *__restrict__ oy,
                                                      const int32_t *__restrict__ ox, uint64_t seed, uint64_t offset,
                                                      int sample0)
{
    constexpr int W4 = W / 4;
    constexpr int PARTS = H / RP;
    static_assert(RP * W4 <= 256 && H % RP == 0, "one quad per thread");
    const int plane = blockIdx.x / PARTS;
    const int part = blockIdx.x - plane * PARTS;
    const int b = plane / C;
    int dy, dx;
    sample_shift(oy, ox, seed, offset, sample0 + b, b, pad, dy, dx);
    const int j = threadIdx.x;
    if (j >= RP * W4) return;
    const int yy = j / W4;
    const int x4 = (j - yy * W4) * 4;
    const int y = part * RP + yy;
    const int sy = clampi(y + dy, 0, H - 1);
    const int base = clampi(x4 + dx, 0, W - 4);
    const TIn *__restrict__ src = in + (size_t)plane * (H * W) + sy * W;
    TOut *__restrict__ dst = out + (size_t)plane * (H * W) + y * W + x4;
    Quad<W, TIn> q;
    q.load(src, base);
    if constexpr (sizeof(TOut) == 1)
        store_nt(reinterpret_cast<uint32_t *>(dst), q.packed(x4 + dx - base));
    else
        store_nt(reinterpret_cast<float *>(dst), q.get(x4 + dx - base));
}

// any shape / alignment: one element per thread (correctness fallback, not tuned)
template <typename TIn, typename TOut>
__global__ __launch_bounds__(256) void k_random_shift_generic(const TIn *__restrict__ in, TOut *__restrict__ out,
                                                              int C, int H, int W, int pad,
                                                              const int32_t *__restrict__ oy,
                                                              const int32_t *__restrict__ ox, uint64_t seed,
                                                              uint64_t offset, int sample0)
{
    const int plane = blockIdx.y;
    const int b = plane / C;
    int dy, dx;
    sample_shift(oy, ox, seed, offset, sample0 + b, b, pad, dy, dx);
    const int n = H * W;
    const TIn *__restrict__ src = in + (size_t)plane * n;
    TOut *__restrict__ dst = out + (size_t)plane * n;
    for (int j = blockIdx.x * 256 + threadIdx.x; j < n; j += gridDim.x * 256) {
        const int y = j / W, x = j - y * W;
        dst[j] = (TOut)src[clampi(y + dy, 0, H - 1) * W + clampi(x + dx, 0, W - 1)];
    }
}

// =========================================================================================
// random_conv (+ optional fused random_shift front end, + optional gather from the replay ring)
//   work item = one R-row strip of one RGB frame; the (R+2) x 3 x W input rows are staged ONCE
//   into LDS already divided by 255 (the reference's x/255.0, one IEEE division per input element
//   instead of 27), 3x3 windows are then read from LDS: one ds_read_b128 + two edge ds_read_b32
//   per (channel,row), 81 v_pk_fma_f32 pairs per output pixel quad, sigmoid, *255, one float4 store
//   per output channel.
//   The 81 weights of the item's sample sit ONCE in LDS and are read with broadcast ds_reads one input
//   channel (27 weights) at a time; v_pk_fma_f32 takes a weight as an op_sel-splatted register half,
//   so no (w,w) pair has to exist anywhere (the duplicated-pair layout cost twice the LDS traffic:
//   more LDS time than the pixel windows).
// =========================================================================================
template <int H, int W, int R>
struct ConvSmem {
    float tile[3][R + 2][W]; // x/255, rows y0-1 .. y0+R (row-clamped = replicate pad)
    float w[84];             // the sample's 81 weights [o][i][ky][kx] (+3: the Philox draw comes in fours)
};

// x / 255.0f, correctly rounded, in 3 VALU ops instead of the ~10 of the IEEE division
// sequence: q = x*r; q' = fma(fma(-q,255,x), r, q) with r = RN(1/255).  Verified EXHAUSTIVELY
// (all 2^31 non-negative finite f32, denormals included; tests/div255_check.c, run by tests/test_div255.py) to equal the IEEE
// quotient bit for bit; +-inf would turn into NaN in the residual, so it is passed through.
__device__ __forceinline__ float div255(float x)
{
    const float r = 1.0f / 255.0f;
    const float q = x * r;
    const float e = __builtin_fmaf(-q, 255.0f, x);
    const float y = __builtin_fmaf(e, r, q);
    return __builtin_isinf(x) ? x : y;
}

// the same for FINITE x (pixel data): without the +-inf pass-through
__device__ __forceinline__ float div255_finite(float x)
{
    const float r = 1.0f / 255.0f;
    const float q = x * r;
    return __builtin_fmaf(__builtin_fmaf(-q, 255.0f, x), r, q);
}

__device__ __forceinline__ float sigmoid255(float a)
{
    // torch.sigmoid = 1/(1+exp(-a)).  exp(-a) = 2^(-a*log2 e) on the v_exp_f32 unit: the rounding of
    // the product perturbs the result by <= |a| * 2^-24 relative, i.e. s(1-s)|a| * 6e-8 <= 1.4e-8 on
    // the sigmoid (x255: < 4e-6), and v_exp_f32 / v_rcp_f32 are 1 ulp: all far inside the 1e-5 budget
    // and an order of magnitude below the conv's own summation-order noise.
    const float e = __builtin_amdgcn_exp2f(a * -1.4426950408889634f);
    return __builtin_amdgcn_rcpf(1.0f + e) * 255.0f;
}


// one thread = 4 consecutive pixels of one row, all 3 output channels.  FMAs are issued as
// v_pk_fma_f32 on the pixel pairs (0,1) and (2,3): for tap kx the operand pairs are
// (v[kx],v[kx+1]) and (v[kx+2],v[kx+3]) of the 6-wide window v = [left, m0..m3, right]; the even
// pairs come from ds_read2_b32, the odd pairs are the two halves of one aligned ds_read_b128.
// Accumulation order per output: i, ky, kx ascending (same as oracle/aug_oracle.c).
template <int H, int W, int R>
__device__ __forceinline__ void conv_compute_store(const ConvSmem<H, W, R> &sm, float *__restrict__ out_frame, int y0,
                                                   int tid)
{
    constexpr int W4 = W / 4;
    if (tid >= R * W4) return;
    const int ry = tid / W4;
    const int x4 = (tid - ry * W4) * 4;
    const int xl = x4 > 0 ? x4 - 1 : 0;
    const int xr = x4 + 4 < W ? x4 + 4 : W - 1;

    f32x2 acc[3][2];
#pragma unroll
    for (int o = 0; o < 3; ++o) acc[o][0] = acc[o][1] = f32x2{0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        f32x2 pe[3][3], po[3][2]; // even pairs (v0v1,v2v3,v4v5), odd pairs (v1v2,v3v4)
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const float *row = sm.tile[i][ry + ky];
            const float4 m = *reinterpret_cast<const float4 *>(row + x4);
            po[ky][0] = f32x2{m.x, m.y};
            po[ky][1] = f32x2{m.z, m.w};
            pe[ky][0] = f32x2{row[xl], m.x};
            pe[ky][1] = f32x2{m.y, m.z};
            pe[ky][2] = f32x2{m.w, row[xr]};
        }
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx)
#pragma unroll
                for (int o = 0; o < 3; ++o) {
                    const float ws = sm.w[((o * 3 + i) * 3 + ky) * 3 + kx]; // same address in every lane: broadcast read
                    const f32x2 w2 = f32x2{ws, ws}; // op_sel splat of one register half: no move, no second copy
                    const f32x2 in01 = (kx == 1) ? po[ky][0] : pe[ky][kx / 2];
                    const f32x2 in23 = (kx == 1) ? po[ky][1] : pe[ky][kx / 2 + 1];
                    acc[o][0] = __builtin_elementwise_fma(w2, in01, acc[o][0]);
                    acc[o][1] = __builtin_elementwise_fma(w2, in23, acc[o][1]);
                }
    }
#pragma unroll
    for (int o = 0; o < 3; ++o) {
        const float4 res = make_float4(sigmoid255(acc[o][0].x), sigmoid255(acc[o][0].y), sigmoid255(acc[o][1].x),
                                       sigmoid255(acc[o][1].y));
        store_nt(out_frame + ((size_t)o * H + (y0 + ry)) * W + x4, res);
    }
}

//   SHIFT  = false: plain random_conv (in -> out_aug);  true: out_shift = random_shift(in), out_aug = conv(out_shift)
//   GATHER = true : `in` is the replay buffer's uint8 frame ring; the frames of sample b are named by
//                   slot_tab[idx[b]][0..frames) (ReplayBuffer.sample + SVEA's front end in ONE pass over the ring)
//   One workgroup per strip, 8 workgroups per CU: every thread first issues ALL of its strip loads (NI quads in
//   flight), then converts, optionally stores the shifted rows, divides by 255 and writes the LDS tile.  (A
//   persistent, software-pipelined form -- next strip's loads issued before the current strip's FMAs -- measured
//   SLOWER, 31.4 vs 26.7 us at B=256: the prefetch registers cut occupancy to 4-5 waves per SIMD and the two barriers
//   per strip serialise each workgroup, while eight independent short workgroups per CU desynchronise by themselves
//   and keep loads, FMAs and stores of different strips overlapped; tools/kbench2.hip has both.)
template <int H, int W, int R, int NT, typename TIn, bool SHIFT, bool GATHER, int EXP = (sizeof(TIn) == 1 ? 7 : 2)>
__global__ __launch_bounds__(NT) void k_conv(const TIn *__restrict__ in, float *__restrict__ out_shift,
                                             float *__restrict__ out_aug, int n_frames, int frames, int pad,
                                             const int32_t *__restrict__ oy, const int32_t *__restrict__ ox,
                                             const float *__restrict__ weights, uint64_t seed, uint64_t offset,
                                             int sample0, const int32_t *__restrict__ slot_tab,
                                             const int64_t *__restrict__ idx, int tab_cols)
{
    constexpr int NS = H / R;
    constexpr int W4 = W / 4;
    constexpr int NQ = 3 * (R + 2) * W4;
    constexpr int NI = (NQ + NT - 1) / NT;
    static_assert(R * W4 <= NT && H % R == 0, "one compute task per thread");
    __shared__ __attribute__((aligned(16))) ConvSmem<H, W, R> sm;
    // XCD-aware item map: workgroups g, g+8, g+16, ... share an XCD (and its L2), so the NS strips of
    // one frame are given to consecutive workgroups of ONE residue class: their halo rows are then
    // L2 hits instead of fabric re-reads (PMC: FETCH 1.19x -> ~1.0x of the algorithmic bytes).
    // Pure speed heuristic: any placement is correct.  Frames beyond a multiple of 8 map plainly.
    int strip, fidx;
    {
        const int nfr8 = (n_frames / 8) * 8;
        const int bid = blockIdx.x;
        if (bid < nfr8 * NS) {
            const int g = bid & 7, q = bid >> 3;
            fidx = (q / NS) * 8 + g;
            strip = q - (q / NS) * NS;
        } else {
            strip = bid % NS;
            fidx = bid / NS;
        }
    }
    const int b = fidx / frames; // fidx = b*frames + f
    const int y0 = strip * R;
    int dy = 0, dx = 0;
    if constexpr (SHIFT) sample_shift(oy, ox, seed, offset, sample0 + b, b, pad, dy, dx);
    const TIn *__restrict__ src;
    if constexpr (GATHER) {
        const int slot = slot_tab[(size_t)idx[b] * tab_cols + (fidx - b * frames)];
        src = in + (size_t)slot * (3 * H * W);
    } else {
        src = in + (size_t)fidx * 3 * H * W;
    }

    // EXP bit 0: raise the wave priority while the strip's loads are issued (uint8 front end: 31.1 -> 30.0 us at B=256; it
    // costs the f32 kernel 0.9 us, whose prologue is 4x the bytes); bit 1: x/255 without the +-inf pass-through; bit 2:
    // branch-free load issue.  (Non-temporal LOADS of the input changed nothing: 25.4 us either way.)  Measured with
    // tools/kbench2.hip; all variants within ~3 % -- at B=256 the launch is ramp/drain-bound (plain copy: 23.9 us).
    if (EXP & 1) __builtin_amdgcn_s_setprio(3);
    Quad<W, TIn> raw[NI];
#pragma unroll
    for (int k = 0; k < NI; ++k) {
        const int j0 = threadIdx.x + k * NT;
        const int j = (EXP & 4) ? min(j0, NQ - 1) : j0; // bit 2: branch-free issue (lanes past the strip re-load its last quad)
        if ((EXP & 4) || j < NQ) {
            const int ch = j / ((R + 2) * W4);
            const int rem = j - ch * ((R + 2) * W4);
            const int rr = rem / W4;
            const int x4 = (rem - rr * W4) * 4;
            const int yc = clampi(y0 - 1 + rr, 0, H - 1); // conv's replicate pad (on the shifted image)
            const int sy = clampi(yc + dy, 0, H - 1);     // shift's replicate pad on the source
            raw[k].load(src + ((size_t)ch * H + sy) * W, clampi(x4 + dx, 0, W - 4));
        }
    }
    if (EXP & 1) __builtin_amdgcn_s_setprio(0);
    if (out_aug != nullptr) { // the sample's 81 weights -> LDS (caller's array, or 21 lanes draw 84 Philox normals)
        if (weights != nullptr) {
            if (threadIdx.x < 81) sm.w[threadIdx.x] = weights[(size_t)b * 81 + threadIdx.x];
        } else if (threadIdx.x < 21) {
            float w4[4];
            philox_conv_weights4(seed, offset, (uint32_t)(sample0 + b), threadIdx.x, w4);
#pragma unroll
            for (int k = 0; k < 4; ++k) sm.w[4 * threadIdx.x + k] = w4[k];
        }
    }
#pragma unroll
    for (int k = 0; k < NI; ++k) {
        const int j = threadIdx.x + k * NT;
        if (j < NQ) {
            const int ch = j / ((R + 2) * W4);
            const int rem = j - ch * ((R + 2) * W4);
            const int rr = rem / W4;
            const int x4 = (rem - rr * W4) * 4;
            const int ys = y0 - 1 + rr;
            float4 q = raw[k].get(SHIFT ? x4 + dx - clampi(x4 + dx, 0, W - 4) : 0);
            if (SHIFT && out_shift != nullptr && ys >= y0 && ys < y0 + R)
                store_nt(out_shift + ((size_t)(fidx * 3 + ch) * H + ys) * W + x4, q);
            if (EXP & 2) {
                q.x = div255_finite(q.x); q.y = div255_finite(q.y); q.z = div255_finite(q.z); q.w = div255_finite(q.w);
            } else {
                q.x = div255(q.x); q.y = div255(q.y); q.z = div255(q.z); q.w = div255(q.w);
            }
            *reinterpret_cast<float4 *>(&sm.tile[ch][rr][x4]) = q;
        }
    }
    if (out_aug == nullptr) return;
    __syncthreads();
    conv_compute_store<H, W, R>(sm, out_aug + (size_t)fidx * 3 * H * W, y0, threadIdx.x);
}

// any shape: one output pixel (3 channels) per thread straight from global memory
__global__ __launch_bounds__(256) void k_random_conv_generic(const float *__restrict__ in, float *__restrict__ out,
                                                             int frames, int H, int W,
                                                             const float *__restrict__ weights, uint64_t seed,
                                                             uint64_t offset, int sample0)
{
    __shared__ float w[84];
    const int fidx = blockIdx.y;
    const int b = fidx / frames;
    const int tid = threadIdx.x;
    if (weights != nullptr) {
        if (tid < 81) w[tid] = weights[(size_t)b * 81 + tid];
    } else if (tid < 21) {
        float w4[4];
        philox_conv_weights4(seed, offset, (uint32_t)(sample0 + b), tid, w4);
        for (int k = 0; k < 4; ++k) w[4 * tid + k] = w4[k];
    }
    __syncthreads();
    const int n = H * W;
    const float *__restrict__ src = in + (size_t)fidx * 3 * n;
    float *__restrict__ dst = out + (size_t)fidx * 3 * n;
    for (int j = blockIdx.x * 256 + tid; j < n; j += gridDim.x * 256) {
        const int y = j / W, x = j - y * W;
        float acc[3] = {0.f, 0.f, 0.f};
        for (int i = 0; i < 3; ++i)
            for (int ky = 0; ky < 3; ++ky)
                for (int kx = 0; kx < 3; ++kx) {
                    const float v =
                        src[(size_t)i * n + clampi(y + ky - 1, 0, H - 1) * W + clampi(x + kx - 1, 0, W - 1)] / 255.0f;
                    for (int o = 0; o < 3; ++o)
                        acc[o] = __builtin_fmaf(w[((o * 3 + i) * 3 + ky) * 3 + kx], v, acc[o]);
                }
        for (int o = 0; o < 3; ++o) dst[(size_t)o * n + j] = sigmoid255(acc[o]);
    }
}

// =========================================================================================
// random_overlay: out = ((1-a)*(x/255.0) + a*ov)*255.0, literal op order, no contraction.
//   one thread = one float4 of the (3,H,W) overlay image, reused for all frames of the sample
// =========================================================================================
__global__ __launch_bounds__(256) void k_random_overlay(const float *__restrict__ in,
                                                        const float *__restrict__ overlay, float *__restrict__ out,
                                                        int frames, int plane4 /* 3*H*W/4 */, float alpha,
                                                        float one_minus_alpha)
{
    const int b = blockIdx.y;
    const int q = blockIdx.x * 256 + threadIdx.x;
    if (q >= plane4) return;
    const float4 ov = reinterpret_cast<const float4 *>(overlay)[(size_t)b * plane4 + q];
    const float4 c = make_float4(alpha * ov.x, alpha * ov.y, alpha * ov.z, alpha * ov.w);
    for (int f = 0; f < frames; ++f) {
        const size_t idx = ((size_t)b * frames + f) * plane4 + q;
        const float4 x = reinterpret_cast<const float4 *>(in)[idx];
        float4 o;
        o.x = (one_minus_alpha * div255(x.x) + c.x) * 255.0f;
        o.y = (one_minus_alpha * div255(x.y) + c.y) * 255.0f;
        o.z = (one_minus_alpha * div255(x.z) + c.z) * 255.0f;
        o.w = (one_minus_alpha * div255(x.w) + c.w) * 255.0f;
        store_nt(out + 4 * (size_t)idx, o);
    }
}

__global__ __launch_bounds__(256) void k_random_overlay_generic(const float *__restrict__ in,
                                                                const float *__restrict__ overlay,
                                                                float *__restrict__ out, int frames, int plane,
                                                                float alpha, float one_minus_alpha)
{
    const int b = blockIdx.y;
    for (int q = blockIdx.x * 256 + threadIdx.x; q < plane; q += gridDim.x * 256) {
        const float c = alpha * overlay[(size_t)b * plane + q];
        for (int f = 0; f < frames; ++f) {
            const size_t idx = ((size_t)b * frames + f) * plane + q;
            out[idx] = (one_minus_alpha * (in[idx] / 255.0f) + c) * 255.0f;
        }
    }
}

// =========================================================================================
// color_jitter (torchvision 0.8.2 float path, restated -- blends literally, the hue op in VALU-friendly form, see
// hue_shift): one workgroup per RGB frame; the
// whole frame (3*H*W f32 = 84.7 KB at 84x84) lives in LDS, so HBM sees one read + one write
// even though adjust_contrast needs the frame's mean before it can touch a pixel.
// A frame is processed iff it appears in idx (duplicates collapse, matching the reference's
// gather-then-scatter semantics; idx == NULL selects every frame); the workgroup scans the
// short idx list itself, so no flag workspace is needed and unselected frames cost no HBM.
// =========================================================================================
struct JitterParams {
    int order[4];
    float c_r, c_inv; // contrast ratio, (1 - ratio) rounded from double
    float s_r, s_inv;
    float hue;
    float bound, scale;
};

__device__ __forceinline__ float clampf(float v, float lo, float hi) { return fminf(fmaxf(v, lo), hi); }
__device__ __forceinline__ float gray(float r, float g, float b) { return (0.2989f * r + 0.587f * g) + 0.114f * b; }
__device__ __forceinline__ float clamp01(float v) { return __builtin_amdgcn_fmed3f(v, 0.0f, 1.0f); } // folds into the
                                                                                  // producer's `clamp` output modifier
__device__ __forceinline__ float blend(float a, float b, float ratio, float inv, float bound)
{
    return clampf(ratio * a + inv * b, 0.0f, bound);
}
// torchvision's _blend with its own ceiling of 1.0 (the only one the operator is ever called with from Python)
__device__ __forceinline__ float blend1(float a, float b, float ratio, float inv) { return clamp01(ratio * a + inv * b); }

// 1/x to <= 1 ulp: v_rcp_f32 (1 ulp) + one Newton step, 3 VALU ops instead of the ~10 of the IEEE division sequence
__device__ __forceinline__ float rcp_nr(float x)
{
    const float r = __builtin_amdgcn_rcpf(x);
    return __builtin_fmaf(__builtin_fmaf(-x, r, 1.0f), r, r);
}

__device__ __forceinline__ void hue_shift(float &r, float &g, float &b, float hue)
{
    // torchvision _rgb2hsv -> h = (h + hue) % 1 -> _hsv2rgb, restated for the VALU (the IEEE-faithful form cost ~200
    // instructions per pixel and made the whole operator compute-bound at 0.37 of the HBM roofline):
    //  * the quotients cr/maxc and (maxc-c)/cr use rcp + one Newton step (<= 2 ulp from the IEEE quotient), h/6 is a
    //    multiplication by RN(1/6);
    //  * fmod(x, 1) and python's x % 1.0 are v_fract_f32 (x - floor(x)); where the reference's remainder rounds up
    //    to exactly 1.0 (x = -tiny) fract gives 1 - 2^-24, and sector 5 at f -> 1 equals sector 0 at f = 0 (HSV -> RGB
    //    is continuous), so the output moves by <= 1 ulp;
    //  * HSV -> RGB: channel c = v * (1 - s * (1 - k_c)), k_c = clamp(|fract(h + K_c) * 6 - 3| - 1, 0, 1) with
    //    K = (1, 2/3, 1/3): algebraically torchvision's sector table (k = 1 -> v, k = 0 -> p, in between -> q / t)
    //    without the five-way selects.  p, q, t are clamped to [0, 1] like the reference's, v is not.
    // All deviations are a few ulp: the parity tolerance of this operator is 1e-5 (its torchvision arithmetic is
    // unpinned in the first place: the library is absent from the container).
    const float maxc = fmaxf(r, fmaxf(g, b)), minc = fminf(r, fminf(g, b));
    const float cr = maxc - minc;
    const bool eqc = (maxc == minc);
    const float sat = cr * rcp_nr(eqc ? 1.0f : maxc);
    const float inv = rcp_nr(eqc ? 1.0f : cr);
    const float rc = (maxc - r) * inv, gc = (maxc - g) * inv, bc = (maxc - b) * inv;
    const float h6 = (maxc == r) ? (bc - gc) : (maxc == g) ? ((2.0f + rc) - bc) : ((4.0f + gc) - rc);
    const float h = __builtin_amdgcn_fractf(__builtin_fmaf(h6, 0.16666667163372040f, 1.0f));
    const float hh = __builtin_amdgcn_fractf(h + hue);
    const float v = maxc;
    auto chan = [&](float K) -> float {
        const float x = __builtin_amdgcn_fractf(hh + K);
        const float k = clamp01(fabsf(__builtin_fmaf(x, 6.0f, -3.0f)) - 1.0f);
        const float m = v * (1.0f - sat * (1.0f - k));
        return k >= 1.0f ? v : clamp01(m);
    };
    r = chan(1.0f);
    g = chan(0.66666668653488159f);
    b = chan(0.33333334326744080f);
}

// Register-resident variant (the product path for 84x84 / 100x100 frames): NT threads own the whole RGB
// frame, NQ quads of 4 pixels per thread and channel (<= 60 data VGPRs), so the only LDS traffic is the
// NT/64 partial sums of the contrast mean.  Every load of a thread is issued up front (3*NQ dwordx4 in
// flight), several workgroups fit per CU, and HBM sees one read and one write per element.
// The jitter ops on a register-resident frame (NQ quads per thread and channel); `red` = NT/64 doubles of LDS.
template <int NT, int NQ, bool UNIT> // UNIT: blend ceiling == 1.0 (clamps become free output modifiers)
__device__ __forceinline__ void jitter_frame_ops(float4 (&px)[3][NQ], const JitterParams &p, double *red, int n, int n4,
                                                 int tid)
{
    auto bl = [&](float a, float b, float ratio, float inv) { return UNIT ? blend1(a, b, ratio, inv) : blend(a, b, ratio, inv, p.bound); };
    if (p.scale == 255.0f) { // uniform branches: a per-element select would execute the IEEE division as well
#pragma unroll
        for (int c = 0; c < 3; ++c)
#pragma unroll
            for (int j = 0; j < NQ; ++j) {
                float *v = &px[c][j].x;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = div255_finite(v[e]); // pixel data: finite
            }
    } else if (p.scale != 1.0f) {
#pragma unroll
        for (int c = 0; c < 3; ++c)
#pragma unroll
            for (int j = 0; j < NQ; ++j) {
                float *v = &px[c][j].x;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = v[e] / p.scale;
            }
    }
    for (int s = 0; s < 4; ++s) {
        const int op = p.order[s];
        if (op == 1) {
            double acc = 0.0; // fixed order: own pixels, wave tree, waves in order => deterministic
#pragma unroll
            for (int j = 0; j < NQ; ++j)
                if (tid + j * NT < n4) {
                    const float *r = &px[0][j].x, *gg = &px[1][j].x, *b = &px[2][j].x;
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc += (double)gray(r[e], gg[e], b[e]);
                }
            for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o, 64);
            __syncthreads(); // red[] may still be read from an earlier contrast pass
            if ((tid & 63) == 0) red[tid >> 6] = acc;
            __syncthreads();
            double tot = 0.0;
            for (int w = 0; w < NT / 64; ++w) tot += red[w];
            const float mean = (float)(tot / (double)n);
#pragma unroll
            for (int c = 0; c < 3; ++c)
#pragma unroll
                for (int j = 0; j < NQ; ++j) {
                    float *v = &px[c][j].x;
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = bl(v[e], mean, p.c_r, p.c_inv);
                }
        } else if (op == 2) {
#pragma unroll
            for (int j = 0; j < NQ; ++j) {
                float *r = &px[0][j].x, *gg = &px[1][j].x, *b = &px[2][j].x;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float gy = gray(r[e], gg[e], b[e]);
                    r[e] = bl(r[e], gy, p.s_r, p.s_inv);
                    gg[e] = bl(gg[e], gy, p.s_r, p.s_inv);
                    b[e] = bl(b[e], gy, p.s_r, p.s_inv);
                }
            }
        } else if (op == 3) {
#pragma unroll
            for (int j = 0; j < NQ; ++j) {
                float *r = &px[0][j].x, *gg = &px[1][j].x, *b = &px[2][j].x;
#pragma unroll
                for (int e = 0; e < 4; ++e) hue_shift(r[e], gg[e], b[e], p.hue);
            }
        }
    }
    if (p.scale != 1.0f) {
#pragma unroll
        for (int c = 0; c < 3; ++c)
#pragma unroll
            for (int j = 0; j < NQ; ++j) {
                float4 &v = px[c][j];
                v.x = v.x * p.scale; v.y = v.y * p.scale; v.z = v.z * p.scale; v.w = v.w * p.scale;
            }
    }
}

template <int NT, int NQ, int WPS, bool UNIT = true>
__global__ __launch_bounds__(NT, WPS) void k_color_jitter_reg(float *__restrict__ inout, int n /* H*W, %4==0 */,
                                                         const int32_t *__restrict__ idx, int n_idx, JitterParams p)
{
    __shared__ double red[NT / 64];
    const int frame = blockIdx.x;
    const int tid = threadIdx.x;
    if (idx != nullptr) {
        int hit = 0;
        for (int k = tid; k < n_idx; k += NT) hit |= (idx[k] == frame);
        if (!__syncthreads_or(hit)) return;
    }
    float4 *__restrict__ g = reinterpret_cast<float4 *>(inout + (size_t)frame * 3 * n);
    const int n4 = n / 4;
    float4 px[3][NQ];
#pragma unroll
    for (int j = 0; j < NQ; ++j) {
        const int q = tid + j * NT;
#pragma unroll
        for (int c = 0; c < 3; ++c) px[c][j] = q < n4 ? g[c * n4 + q] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    jitter_frame_ops<NT, NQ, UNIT>(px, p, red, n, n4, tid);
#pragma unroll
    for (int j = 0; j < NQ; ++j) {
        const int q = tid + j * NT;
        if (q < n4) {
#pragma unroll
            for (int c = 0; c < 3; ++c) g[c * n4 + q] = px[c][j];
        }
    }
}

// (Splitting the work into two streaming phases around the contrast mean -- per-quad operators as the loads land, per-quad
// stores behind the arithmetic -- measured SLOWER, 36.1 vs 32.4 us at B=256: the operator chain is then instantiated
// per quad and phase with run-time trip counts and the scheduler loses the cross-quad interleaving it has here.)

// LDS-resident variant for other frame sizes (whole frame in LDS, up to 160 KiB)
template <int NT>
__global__ __launch_bounds__(NT) void k_color_jitter(float *__restrict__ inout, int n /* H*W, %4==0 */,
                                                     const int32_t *__restrict__ idx, int n_idx, JitterParams p)
{
    extern __shared__ __attribute__((aligned(16))) float fr[]; // [3][n] then [NT/64] partial sums (double)
    const int frame = blockIdx.x;
    const int tid = threadIdx.x;
    if (idx != nullptr) {
        int hit = 0;
        for (int k = tid; k < n_idx; k += NT) hit |= (idx[k] == frame);
        if (!__syncthreads_or(hit)) return;
    }
    float *__restrict__ g = inout + (size_t)frame * 3 * n;
    const int n4 = n / 4;
    for (int q = tid; q < 3 * n4; q += NT) {
        float4 v = reinterpret_cast<const float4 *>(g)[q];
        if (p.scale != 1.0f) {
            v.x = v.x / p.scale; v.y = v.y / p.scale; v.z = v.z / p.scale; v.w = v.w / p.scale;
        }
        reinterpret_cast<float4 *>(fr)[q] = v;
    }
    __syncthreads();
    float *R = fr, *G = fr + n, *Bc = fr + 2 * n;
    double *red = reinterpret_cast<double *>(fr + 3 * n);
    for (int s = 0; s < 4; ++s) {
        const int op = p.order[s];
        if (op == 1) {
            // frame mean of the grayscale image; fixed-order tree (f64 partials) => deterministic
            double acc = 0.0;
            for (int k = tid; k < n; k += NT) acc += (double)gray(R[k], G[k], Bc[k]);
            for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o, 64);
            if ((tid & 63) == 0) red[tid >> 6] = acc;
            __syncthreads();
            double tot = 0.0;
            for (int w = 0; w < NT / 64; ++w) tot += red[w];
            const float mean = (float)(tot / (double)n);
            __syncthreads();
            for (int k = tid; k < n; k += NT) {
                R[k] = blend(R[k], mean, p.c_r, p.c_inv, p.bound);
                G[k] = blend(G[k], mean, p.c_r, p.c_inv, p.bound);
                Bc[k] = blend(Bc[k], mean, p.c_r, p.c_inv, p.bound);
            }
        } else if (op == 2) {
            for (int k = tid; k < n; k += NT) {
                const float gy = gray(R[k], G[k], Bc[k]);
                R[k] = blend(R[k], gy, p.s_r, p.s_inv, p.bound);
                G[k] = blend(G[k], gy, p.s_r, p.s_inv, p.bound);
                Bc[k] = blend(Bc[k], gy, p.s_r, p.s_inv, p.bound);
            }
        } else if (op == 3) {
            for (int k = tid; k < n; k += NT) {
                float r = R[k], gg = G[k], b = Bc[k];
                hue_shift(r, gg, b, p.hue);
                R[k] = r; G[k] = gg; Bc[k] = b;
            }
        }
        // each thread touches only its own pixels k = tid (mod NT) across ops; only the mean needs a barrier
    }
    __syncthreads();
    for (int q = tid; q < 3 * n4; q += NT) {
        float4 v = reinterpret_cast<const float4 *>(fr)[q];
        if (p.scale != 1.0f) {
            v.x = v.x * p.scale; v.y = v.y * p.scale; v.z = v.z * p.scale; v.w = v.w * p.scale;
        }
        reinterpret_cast<float4 *>(g)[q] = v;
    }
}

// ---- Philox parameter export (same device functions the kernels inline) ------------------
__global__ void k_philox_shift(uint64_t seed, uint64_t offset, int sample0, int B, int pad, int32_t *oy, int32_t *ox)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b < B) {
        int y, x;
        philox_shift_offsets(seed, offset, (uint32_t)(sample0 + b), pad, y, x);
        oy[b] = y;
        ox[b] = x;
    }
}

__global__ void k_philox_conv_w(uint64_t seed, uint64_t offset, int sample0, int B, float *w)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const int b = t / 21, j = t - b * 21;
    if (b < B) {
        float w4[4];
        philox_conv_weights4(seed, offset, (uint32_t)(sample0 + b), j, w4);
        for (int k = 0; k < 4; ++k)
            if (4 * j + k < 81) w[(size_t)b * 81 + 4 * j + k] = w4[k];
    }
}

__global__ void k_philox_replay_idx(uint64_t seed, uint64_t offset, int sample0, uint32_t n_valid, int64_t *idx, int B)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b < B) {
        const u32x4 r = philox_sample(seed, offset, (uint32_t)(sample0 + b), 32u);
        idx[b] = (int64_t)(((uint64_t)r.x * n_valid) >> 32);
    }
}

__global__ void k_philox_normal(uint64_t seed, uint64_t offset, size_t n, float *out)
{
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; // one Philox call = 4 normals
    if (4 * t >= n) return;
    u32x4 c{(uint32_t)t, (uint32_t)(t >> 32), (uint32_t)offset, (uint32_t)(offset >> 32)};
    u32x4 r = philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32) ^ 0x5eedu);
    float v[4];
    box_muller(r.x, r.y, v[0], v[1]);
    box_muller(r.z, r.w, v[2], v[3]);
    for (int k = 0; k < 4; ++k)
        if (4 * t + k < n) out[4 * t + k] = v[k];
}

} // namespace augcl

// =========================================================================================
// C ABI
// =========================================================================================
using namespace augcl;

#define LAUNCH_CHECK(name) return augcl::check_launch(name)

static bool fast_shape(int H, int W) { return (H == 84 && W == 84) || (H == 100 && W == 100); }

template <typename TIn, typename TOut>
static int shift_dispatch(const TIn *in, TOut *out, int B, int C, int H, int W, int pad, const int32_t *oy,
                          const int32_t *ox, uint64_t seed, uint64_t offset, int sample0, augcl_stream_t stream,
                          const char *name)
{
    AUGCL_REQUIRE(B >= 0 && C > 0 && H > 0 && W > 0 && pad >= 0, "%s: bad shape B=%d C=%d H=%d W=%d pad=%d", name, B,
                  C, H, W, pad);
    if (B == 0) return AUGCL_OK; // an empty batch has no storage: nothing to check, nothing to do
    AUGCL_REQUIRE(in && out, "%s: null tensor", name);
    AUGCL_REQUIRE((oy == nullptr) == (ox == nullptr), "%s: oy/ox must both be given or both NULL", name);
    AUGCL_REQUIRE((const void *)in != (const void *)out, "%s: in-place not supported", name);
    if (B == 0) return AUGCL_OK;
    hipStream_t s = as_stream(stream);
    const bool vec_ok = fast_shape(H, W) && aligned16(in) && aligned16(out);
    if (vec_ok && H == 84)
        hipLaunchKernelGGL((k_random_shift<84, 84, 12, TIn, TOut>), dim3(B * C * 7), dim3(256), 0, s, in, out, C, pad,
                           oy, ox, seed, offset, sample0);
    else if (vec_ok && H == 100)
        hipLaunchKernelGGL((k_random_shift<100, 100, 10, TIn, TOut>), dim3(B * C * 10), dim3(256), 0, s, in, out, C,
                           pad, oy, ox, seed, offset, sample0);
    else {
        const int gx = (H * W + 255) / 256 < 64 ? (H * W + 255) / 256 : 64;
        hipLaunchKernelGGL((k_random_shift_generic<TIn, TOut>), dim3(gx, B * C), dim3(256), 0, s, in, out, C, H, W,
                           pad, oy, ox, seed, offset, sample0);
    }
    LAUNCH_CHECK(name);
}

template <int H, int W, int R, typename TIn, bool SHIFT, bool GATHER>
static void launch_conv(const TIn *in, float *out_shift, float *out_aug, int B, int frames, int pad, const int32_t *oy,
                        const int32_t *ox, const float *weights, uint64_t seed, uint64_t offset, int sample0,
                        const int32_t *slot_tab, const int64_t *idx, int tab_cols, hipStream_t s)
{
    const int n_frames = B * frames;
    hipLaunchKernelGGL((k_conv<H, W, R, 256, TIn, SHIFT, GATHER>), dim3(n_frames * (H / R)), dim3(256), 0, s, in, out_shift,
                       out_aug, n_frames, frames, pad, oy, ox, weights, seed, offset, sample0, slot_tab, idx, tab_cols);
}

template <typename TIn, bool GATHER = false>
static int shift_conv_dispatch(const TIn *in, float *out_shift, float *out_aug, int B, int frames, int H, int W,
                               int pad, const int32_t *oy, const int32_t *ox, const float *weights, uint64_t seed,
                               uint64_t offset, int sample0, augcl_stream_t stream, const char *name,
                               const int32_t *slot_tab = nullptr, const int64_t *idx = nullptr, int tab_cols = 0)
{
    AUGCL_REQUIRE(B >= 0 && frames > 0 && pad >= 0, "%s: bad shape", name);
    if (B == 0) return AUGCL_OK;
    AUGCL_REQUIRE(in && (out_shift || out_aug), "%s: null tensor", name);
    AUGCL_REQUIRE((oy == nullptr) == (ox == nullptr), "%s: oy/ox must both be given or both NULL", name);
    AUGCL_REQUIRE(fast_shape(H, W), "%s: only 84x84 / 100x100 frames are supported (got %dx%d)", name, H, W);
    AUGCL_REQUIRE(aligned16(in) && aligned16(out_shift) && aligned16(out_aug), "%s: tensors must be 16-byte aligned",
                  name);
    AUGCL_REQUIRE(!GATHER || (slot_tab && idx && tab_cols >= frames), "%s: slot table / index list missing", name);
    hipStream_t s = as_stream(stream);
    if (H == 84)
        launch_conv<84, 84, 12, TIn, true, GATHER>(in, out_shift, out_aug, B, frames, pad, oy, ox, weights, seed, offset,
                                                   sample0, slot_tab, idx, tab_cols, s);
    else
        launch_conv<100, 100, 10, TIn, true, GATHER>(in, out_shift, out_aug, B, frames, pad, oy, ox, weights, seed, offset,
                                                     sample0, slot_tab, idx, tab_cols, s);
    LAUNCH_CHECK(name);
}

extern "C" {

int augcl_random_shift_f32(const float *in, float *out, int B, int C, int H, int W, int pad, const int32_t *oy,
                           const int32_t *ox, uint64_t seed, uint64_t offset, int sample0, augcl_stream_t stream)
{
    return shift_dispatch<float, float>(in, out, B, C, H, W, pad, oy, ox, seed, offset, sample0, stream,
                                        "augcl_random_shift_f32");
}

int augcl_random_shift_u8(const uint8_t *in, uint8_t *out, int B, int C, int H, int W, int pad, const int32_t *oy,
                          const int32_t *ox, uint64_t seed, uint64_t offset, int sample0, augcl_stream_t stream)
{
    return shift_dispatch<uint8_t, uint8_t>(in, out, B, C, H, W, pad, oy, ox, seed, offset, sample0, stream,
                                            "augcl_random_shift_u8");
}

int augcl_random_shift_u8_f32(const uint8_t *in, float *out, int B, int C, int H, int W, int pad, const int32_t *oy,
                              const int32_t *ox, uint64_t seed, uint64_t offset, int sample0, augcl_stream_t stream)
{
    return shift_dispatch<uint8_t, float>(in, out, B, C, H, W, pad, oy, ox, seed, offset, sample0, stream,
                                          "augcl_random_shift_u8_f32");
}

int augcl_random_conv_f32(const float *in, float *out, int B, int frames, int H, int W, const float *weights,
                          uint64_t seed, uint64_t offset, int sample0, augcl_stream_t stream)
{
    const char *name = "augcl_random_conv_f32";
    AUGCL_REQUIRE(B >= 0 && frames > 0 && H > 0 && W > 0, "%s: bad shape", name);
    if (B == 0) return AUGCL_OK;
    AUGCL_REQUIRE(in && out, "%s: null tensor", name);
    AUGCL_REQUIRE(in != out, "%s: in-place not supported", name);
    hipStream_t s = as_stream(stream);
    const bool vec_ok = fast_shape(H, W) && aligned16(in) && aligned16(out);
    if (vec_ok && H == 84)
        launch_conv<84, 84, 12, float, false, false>(in, nullptr, out, B, frames, 0, nullptr, nullptr, weights, seed, offset,
                                                     sample0, nullptr, nullptr, 0, s);
    else if (vec_ok && H == 100)
        launch_conv<100, 100, 10, float, false, false>(in, nullptr, out, B, frames, 0, nullptr, nullptr, weights, seed,
                                                       offset, sample0, nullptr, nullptr, 0, s);
    else {
        const int gx = (H * W + 255) / 256 < 64 ? (H * W + 255) / 256 : 64;
        hipLaunchKernelGGL(k_random_conv_generic, dim3(gx, B * frames), dim3(256), 0, s, in, out, frames, H, W,
                           weights, seed, offset, sample0);
    }
    LAUNCH_CHECK(name);
}

int augcl_shift_conv_u8_f32(const uint8_t *in, float *out_shift, float *out_aug, int B, int frames, int H, int W,
                            int pad, const int32_t *oy, const int32_t *ox, const float *weights, uint64_t seed,
                            uint64_t offset, int sample0, augcl_stream_t stream)
{
    return shift_conv_dispatch<uint8_t>(in, out_shift, out_aug, B, frames, H, W, pad, oy, ox, weights, seed, offset,
                                        sample0, stream, "augcl_shift_conv_u8_f32");
}

int augcl_shift_conv_f32(const float *in, float *out_shift, float *out_aug, int B, int frames, int H, int W, int pad,
                         const int32_t *oy, const int32_t *ox, const float *weights, uint64_t seed, uint64_t offset,
                         int sample0, augcl_stream_t stream)
{
    return shift_conv_dispatch<float>(in, out_shift, out_aug, B, frames, H, W, pad, oy, ox, weights, seed, offset,
                                      sample0, stream, "augcl_shift_conv_f32");
}

int augcl_replay_shift_conv_u8_f32(const uint8_t *ring, const int32_t *slot_tab, const int64_t *idx, int tab_cols,
                                   float *out_shift, float *out_aug, int B, int frames, int H, int W, int pad,
                                   const int32_t *oy, const int32_t *ox, const float *weights, uint64_t seed,
                                   uint64_t offset, int sample0, augcl_stream_t stream)
{
    return shift_conv_dispatch<uint8_t, true>(ring, out_shift, out_aug, B, frames, H, W, pad, oy, ox, weights, seed,
                                              offset, sample0, stream, "augcl_replay_shift_conv_u8_f32", slot_tab, idx,
                                              tab_cols);
}

int augcl_random_overlay_f32(const float *in, const float *overlay, float *out, int B, int frames, int H, int W,
                             float alpha, augcl_stream_t stream)
{
    const char *name = "augcl_random_overlay_f32";
    AUGCL_REQUIRE(B >= 0 && frames > 0 && H > 0 && W > 0, "%s: bad shape", name);
    if (B == 0) return AUGCL_OK;
    AUGCL_REQUIRE(in && overlay && out, "%s: null tensor", name);
    hipStream_t s = as_stream(stream);
    const float om = (float)(1.0 - (double)alpha);
    const int plane = 3 * H * W;
    if ((plane & 3) == 0 && aligned16(in) && aligned16(out) && aligned16(overlay)) {
        const int plane4 = plane / 4;
        hipLaunchKernelGGL(k_random_overlay, dim3((plane4 + 255) / 256, B), dim3(256), 0, s, in, overlay, out, frames,
                           plane4, alpha, om);
    } else {
        const int gx = (plane + 255) / 256 < 128 ? (plane + 255) / 256 : 128;
        hipLaunchKernelGGL(k_random_overlay_generic, dim3(gx, B), dim3(256), 0, s, in, overlay, out, frames, plane,
                           alpha, om);
    }
    LAUNCH_CHECK(name);
}

int augcl_color_jitter_f32(float *inout, int B, int frames, int H, int W, const int32_t *idx, int n_idx,
                           const int32_t *order_host, double contrast, double saturation, double hue, float bound,
                           float scale, augcl_stream_t stream)
{
    const char *name = "augcl_color_jitter_f32";
    AUGCL_REQUIRE(B >= 0 && frames > 0 && H > 0 && W > 0 && n_idx >= 0, "%s: bad shape", name);
    AUGCL_REQUIRE((inout || B == 0) && order_host, "%s: null argument", name);
    AUGCL_REQUIRE(idx != nullptr || n_idx == 0, "%s: n_idx > 0 with idx == NULL", name);
    AUGCL_REQUIRE(scale > 0.0f && bound > 0.0f, "%s: scale and bound must be positive", name);
    JitterParams p;
    int seen = 0;
    for (int k = 0; k < 4; ++k) {
        AUGCL_REQUIRE(order_host[k] >= 0 && order_host[k] < 4 && !(seen & (1 << order_host[k])),
                      "%s: order must be a permutation of 0..3", name);
        seen |= 1 << order_host[k];
        p.order[k] = order_host[k];
    }
    // python-float scalars meet f32 tensors: ratio and (1.0 - ratio) are each rounded to f32 (torchvision _blend)
    p.c_r = (float)contrast;
    p.c_inv = (float)(1.0 - contrast);
    p.s_r = (float)saturation;
    p.s_inv = (float)(1.0 - saturation);
    p.hue = (float)hue;
    p.bound = bound;
    p.scale = scale;
    const int n = H * W;
    constexpr int NT = 1024;
    const size_t smem = (size_t)3 * n * sizeof(float) + (NT / 64) * sizeof(double);
    AUGCL_REQUIRE((n & 3) == 0 && aligned16(inout), "%s: H*W must be a multiple of 4 and the tensor 16-B aligned", name);
    if (smem > 160 * 1024 - 512)
        return augcl::fail(AUGCL_E_UNSUPPORTED, "%s: a %dx%d RGB frame (%zu B) exceeds the 160 KiB LDS of a CU", name, H,
                           W, smem);
    if (B == 0 || (idx != nullptr && n_idx == 0)) return AUGCL_OK;
    if (n / 4 <= 1024 * 2) { // 84x84 (1764 quads): register-resident, two 16-wave workgroups per CU (61 VGPRs)
        // (a persistent one-workgroup-per-CU form with the next frame's loads issued early was the faster one while the
        // hue arithmetic needed 128 registers; with the lean arithmetic two independent workgroups per CU overlap
        // better: 32.4 vs 39.0 us at B=256, tools/kbench2.hip)
        if (bound == 1.0f)
            hipLaunchKernelGGL((k_color_jitter_reg<1024, 2, 4, true>), dim3(B * frames), dim3(1024), 0, as_stream(stream),
                               inout, n, idx, n_idx, p);
        else
            hipLaunchKernelGGL((k_color_jitter_reg<1024, 2, 4, false>), dim3(B * frames), dim3(1024), 0, as_stream(stream),
                               inout, n, idx, n_idx, p);
        LAUNCH_CHECK(name);
    }
    if (n / 4 <= 1024 * 3) { // 100x100 (2500 quads)
        if (bound == 1.0f)
            hipLaunchKernelGGL((k_color_jitter_reg<1024, 3, 4, true>), dim3(B * frames), dim3(1024), 0, as_stream(stream),
                               inout, n, idx, n_idx, p);
        else
            hipLaunchKernelGGL((k_color_jitter_reg<1024, 3, 4, false>), dim3(B * frames), dim3(1024), 0, as_stream(stream),
                               inout, n, idx, n_idx, p);
        LAUNCH_CHECK(name);
    }
    static size_t attr_max = 48 * 1024; // idempotent; a benign race at worst repeats the call
    if (smem > attr_max) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_color_jitter<NT>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        if (e != hipSuccess)
            return augcl::fail(AUGCL_E_LAUNCH, "%s: cannot raise dynamic LDS limit to %zu: %s", name, smem,
                               hipGetErrorString(e));
        attr_max = smem;
    }
    hipLaunchKernelGGL((k_color_jitter<NT>), dim3(B * frames), dim3(NT), smem, as_stream(stream), inout, n, idx, n_idx,
                       p);
    LAUNCH_CHECK(name);
}

int augcl_philox_shift_offsets(uint64_t seed, uint64_t offset, int sample0, int B, int pad, int32_t *oy, int32_t *ox,
                               augcl_stream_t stream)
{
    AUGCL_REQUIRE(oy && ox && B >= 0 && pad >= 0, "augcl_philox_shift_offsets: bad argument");
    if (B == 0) return AUGCL_OK;
    hipLaunchKernelGGL(k_philox_shift, dim3((B + 255) / 256), dim3(256), 0, as_stream(stream), seed, offset, sample0,
                       B, pad, oy, ox);
    LAUNCH_CHECK("augcl_philox_shift_offsets");
}

int augcl_philox_conv_weights(uint64_t seed, uint64_t offset, int sample0, int B, float *weights,
                              augcl_stream_t stream)
{
    AUGCL_REQUIRE(weights && B >= 0, "augcl_philox_conv_weights: bad argument");
    if (B == 0) return AUGCL_OK;
    hipLaunchKernelGGL(k_philox_conv_w, dim3((B * 21 + 255) / 256), dim3(256), 0, as_stream(stream), seed, offset,
                       sample0, B, weights);
    LAUNCH_CHECK("augcl_philox_conv_weights");
}

int augcl_philox_replay_indices(uint64_t seed, uint64_t offset, int sample0, int n_valid, int64_t *idx, int B,
                                augcl_stream_t stream)
{
    AUGCL_REQUIRE(B >= 0 && (B == 0 || (idx && n_valid >= 1)), "augcl_philox_replay_indices: bad argument (empty buffer?)");
    if (B == 0) return AUGCL_OK;
    hipLaunchKernelGGL(k_philox_replay_idx, dim3((B + 255) / 256), dim3(256), 0, as_stream(stream), seed, offset, sample0,
                       (uint32_t)n_valid, idx, B);
    LAUNCH_CHECK("augcl_philox_replay_indices");
}

int augcl_philox_normal(uint64_t seed, uint64_t offset, size_t n, float *out, augcl_stream_t stream)
{
    AUGCL_REQUIRE(out || n == 0, "augcl_philox_normal: null output");
    if (n == 0) return AUGCL_OK;
    const size_t calls = (n + 3) / 4;
    hipLaunchKernelGGL(k_philox_normal, dim3((unsigned)((calls + 255) / 256)), dim3(256), 0, as_stream(stream), seed,
                       offset, n, out);
    LAUNCH_CHECK("augcl_philox_normal");
}

} // extern "C"
