// encoder_first.hip -- first encoder layer for the shapes the training loop uses (9 stacked-RGB channels ->
// 32 filters, 3x3, stride 2, input divided by 255), built like the 32-channel kernels: 4 LOADER waves run two strips
// ahead of 4 COMPUTE waves through a double-buffered LDS tile, every MFMA operand is one ds_read at a row base
// plus a compile-time constant, operand reads run eight MFMAs ahead.  Other shapes keep the generic kernels of
// encoder_kernels.hip.
//
// LDS tile: [input row r][channel c][WI] floats, already divided by the scale.  With k = (ky*9 + c)*3 + kx the tap of
// k sits at off(k) = (k/3)*WI + k%3 past the lane's pixel base (2*yy*9*WI + 2*xx): consecutive k differ by 1 (inside
// a row) or by WI-2 (next row block), so the two k of one f32 MFMA (lane half h takes k = 2j + h) need just two
// lane bases, base + h and base + (WI-2)h, plus immediates.  K = 81 is padded to 82 with a zero weight; the pad's
// operand is a real (finite) float from the slack row block.
#include "augcl_common.h"

#include <algorithm>

namespace augcl {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float vf4 __attribute__((ext_vector_type(4)));
typedef uint32_t vu2 __attribute__((ext_vector_type(2)));
typedef short s16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ int acc_row_f(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

__device__ __forceinline__ float div255x(float x)
{
    const float r = 1.0f / 255.0f;
    const float q = x * r;
    return __builtin_fmaf(__builtin_fmaf(-q, 255.0f, x), r, q); // == x / 255.0f for every finite x (see aug_kernels)
}

struct First9Args {
    const float *in;   // (B,9,Hi,WI) raw 0..255
    const float *w;    // (32,9,3,3)
    const float *bias; // (32) or null
    void *out;         // f32 NCHW (B,32,Ho,Wo) or bf16 NHWC (B,Ho,Wo,32)
    int B, Hi, Ho, Wo, RT, S, n_items;
    float scale;
    uint32_t magic_wo, magic_run_full, magic_run_last; // floor(2^32/d)+1 for d = Wo, chunks per channel run
    // two-weight-set forward over ONE input batch: image b >= B/2 reads input b - B/2 and uses w2 / bias2
    const float *w2, *bias2;
    int dual;
    // f32 output only: max |out| is atomically max-ed into it (float bits, zeroed by the caller; dual: [0] / [1] per half), or
    // null -- the f16-pair kernels of the next layer derive their operand scale from it (encoder_x2.hip)
    uint32_t *amax_out;
    // f32 output with ReLU only: gate bits, one word per output pixel (B,Ho,Wo), bit c = (channel c > 0), or null -- the next
    // layer's backward-data reads them instead of this layer's output (encoder_x2.hip)
    uint32_t *bits_out;
};

constexpr int F9_U = 12;  // 16-byte chunks per loader lane per strip
constexpr int F9_NP = 41; // k pairs (K = 81 padded to 82)
constexpr int F9_AHEAD = 8;
constexpr int F9B_NM = 6; // bf16 mode: MFMAs (K = 16 each) per 32-pixel tile

template <int WI, bool OUT_BF16, bool RELU>
__global__ __launch_bounds__(512, 2) void k_conv_first9(First9Args a)
{
    extern __shared__ __attribute__((aligned(16))) float ldsf[];
    constexpr int C = 9, ROWF = C * WI, W4 = WI / 4;
    static_assert(WI % 4 == 0, "row must be whole 16-byte chunks");
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool loader = wave >= 4;
    const int cw = wave & 3;
    const int l31 = lane & 31, h = lane >> 5;
    const int rows_in_max = 2 * a.RT + 1;
    const int bufwords = (rows_in_max + 1) * ROWF; // + one slack row block (the K pad reads into it)
    float *lbias = ldsf + 2 * bufwords;
    // two weight sets: both halves read the SAME input image, so the two workgroups of an input sit on ONE XCD (workgroup i
    // runs on XCD i % 8), next to each other in dispatch order -- i and i + 8 -- and the second read of the image hits in that
    // XCD's L2 (with even / odd workgroups taking the halves, the two readers sat on different XCDs and the image crossed
    // the fabric twice: PMC FETCH 1.17x the algorithmic bytes).  Grids that are not a multiple of 16 keep the even / odd map.
    const bool pair_xcd = a.dual && (gridDim.x & 15) == 0;
    const int grp = a.dual ? (pair_xcd ? (int)((blockIdx.x >> 3) & 1) : (int)(blockIdx.x & 1)) : 0, nhalf = a.n_items >> 1;
    // (... and XCD x takes the contiguous slots [x G/16, (x + 1) G/16): neighbouring strips, which share a halo row, too)
    const int worker = pair_xcd ? (int)(blockIdx.x & 7) * (int)(gridDim.x >> 4) + (int)(blockIdx.x >> 4) : (int)(blockIdx.x >> 1);
    const int first = a.dual ? grp * nhalf + worker : xcd_slot((int)blockIdx.x, (int)gridDim.x);
    const int NG = a.dual ? (int)(gridDim.x >> 1) : (int)gridDim.x;
    const int n = a.dual ? (grp + 1) * nhalf : a.n_items;

    // Both tile buffers start as finite zeros: the K pad (zero weight) reads one row block past the taps, and a
    // short last strip leaves rows the loader never wrote (0 * NaN from stale LDS would poison the MFMA).
    for (int i = tid; i < 2 * bufwords; i += 512) ldsf[i] = 0.0f;
    __syncthreads();
    // weights -> LDS (coalesced), rows padded to 85 floats: lanes = filters read conflict-free.  Lives in buffer 1,
    // which the loader first writes after the next-but-one barrier.
    {
        float *wst = ldsf + bufwords;
        const float *wsrc = grp ? a.w2 : a.w, *bsrc = grp ? a.bias2 : a.bias;
        for (int i = tid; i < 32 * 81; i += 512) wst[(i / 81) * 85 + (i % 81)] = wsrc[i];
        if (tid < 32) lbias[tid] = bsrc ? bsrc[tid] : 0.0f;
        if (tid < 2) reinterpret_cast<uint32_t *>(lbias)[32 + tid] = 0u; // the workgroup's max |out| and its ticket counter
    }
    __syncthreads();

    if (loader) {
        const int lt = cw * 64 + lane;
        struct Regs {
            vf4 d[F9_U];
        };
        struct Geo {
            int b, y0, rows_in, run4, nchunk;
            uint32_t magic;
        };
        auto geom = [&](int item) {
            Geo g;
            g.b = item / a.S;
            const int s = item - g.b * a.S;
            g.y0 = s * a.RT;
            const int rows = min(a.RT, a.Ho - g.y0);
            g.rows_in = 2 * rows + 1;
            g.run4 = g.rows_in * W4; // chunks per channel run
            g.nchunk = C * g.run4;
            g.magic = rows == a.RT ? a.magic_run_full : a.magic_run_last;
            return g;
        };
        auto issue = [&](int item, Regs &R) {
            const Geo g = geom(item);
            const int bsrc_img = (a.dual && g.b >= (a.B >> 1)) ? g.b - (a.B >> 1) : g.b; // both halves read the same input
            const float *src = a.in + ((size_t)bsrc_img * C * a.Hi + 2 * g.y0) * WI;
#pragma unroll
            for (int u = 0; u < F9_U; ++u) {
                const int q = min(u * 256 + lt, g.nchunk - 1);
                const int c = (int)__umulhi((uint32_t)q, g.magic);
                const int rem = q - c * g.run4;
                R.d[u] = *reinterpret_cast<const vf4 *>(src + (size_t)c * a.Hi * WI + rem * 4);
            }
        };
        auto commit = [&](int item, int bi, const Regs &R) {
            const Geo g = geom(item);
            float *dst = ldsf + bi * bufwords;
#pragma unroll
            for (int u = 0; u < F9_U; ++u) {
                const int q = min(u * 256 + lt, g.nchunk - 1);
                const int c = (int)__umulhi((uint32_t)q, g.magic);
                const int rem = q - c * g.run4;
                const int r = rem / W4, x4 = rem - r * W4;
                vf4 v = R.d[u];
                if (a.scale == 255.0f) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = div255x(v[e]);
                } else if (a.scale != 1.0f) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = v[e] / a.scale;
                }
                *reinterpret_cast<vf4 *>(dst + (r * C + c) * WI + 4 * x4) = v;
            }
        };
        Regs ra, rb;
        if (first < n) {
            issue(first, ra);
            commit(first, 0, ra);
        }
        if (first + NG < n) issue(first + NG, ra);
        if (first + 2 * NG < n) issue(first + 2 * NG, rb);
        __syncthreads();
        for (int item = first; item < n; item += 2 * NG) {
            if (item + NG < n) commit(item + NG, 1, ra);
            if (item + 3 * NG < n) issue(item + 3 * NG, ra);
            __syncthreads();
            if (item + NG >= n) break;
            if (item + 2 * NG < n) commit(item + 2 * NG, 0, rb);
            if (item + 4 * NG < n) issue(item + 4 * NG, rb);
            __syncthreads();
        }
        return;
    }

    // ---- compute waves: A operand of pair j = W[co = l31][k = 2j + h], k = (ky*9 + c)*3 + kx
    float wreg[F9_NP];
    {
        const float *wst = ldsf + bufwords + l31 * 85;
#pragma unroll
        for (int j = 0; j < F9_NP; ++j) {
            const int k0 = 2 * j; // this lane takes k0 + h; both are compile-time once h is resolved by the select
            const int ka = k0, kb = k0 + 1;
            const int ia = ((ka / 3) % 9) * 9 + ((ka / 3) / 9) * 3 + ka % 3; // (c, ky, kx) -> c*9 + ky*3 + kx
            const int ib = kb < 81 ? ((kb / 3) % 9) * 9 + ((kb / 3) / 9) * 3 + kb % 3 : 0;
            const float wa = wst[ia], wb = kb < 81 ? wst[ib] : 0.0f;
            wreg[j] = h ? wb : wa;
        }
    }
    f32x16 bias16;
#pragma unroll
    for (int r = 0; r < 16; ++r) bias16[r] = lbias[acc_row_f(r, h)];
    const __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc(a.out, 0, -1, 0x00020000);
    // bf16 mode: the products run on v_mfma_f32_32x32x16_bf16 -- K = 81 padded to 96 = 6 MFMAs instead of 41.  A fragment
    // of MFMA i: W[co = l31][k = 16 i + 8 h + j], j = 0..7, rounded to bf16 (RNE); k >= 81 carries a zero weight.
    float vmax = 0.0f; // max |output| of this lane (f32 output)
    bf16x8 wfrag[F9B_NM];
    if (OUT_BF16) {
        const float *wst = ldsf + bufwords + l31 * 85;
#pragma unroll
        for (int i = 0; i < F9B_NM; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int ka = 16 * i + j, kb = ka + 8;
                const int ia = ka < 81 ? ((ka / 3) % 9) * 9 + ((ka / 3) / 9) * 3 + ka % 3 : 0;
                const int ib = kb < 81 ? ((kb / 3) % 9) * 9 + ((kb / 3) / 9) * 3 + kb % 3 : 0;
                const float wa = ka < 81 ? wst[ia] : 0.0f, wb = kb < 81 ? wst[ib] : 0.0f;
                wfrag[i][j] = (__bf16)(h ? wb : wa);
            }
    }

    auto compute = [&](int item, int bi) {
        const int b = item / a.S, s = item - b * a.S;
        const int y0 = s * a.RT;
        const int rows = min(a.RT, a.Ho - y0);
        const int npix = rows * a.Wo;
        const char *tile = reinterpret_cast<const char *>(ldsf + bi * bufwords);
        const int ntiles = (npix + 31) >> 5;
        if (cw >= ntiles) return; // wave-uniform
        if constexpr (OUT_BF16) {
            // B operand of MFMA i: this lane's pixel, taps k = 16 i + 8 h + j.  off(k) = (k/3) WI + k%3 floats past the pixel
            // base; off(k + 8) - off(k) takes two values (k%3 == 0: 2 WI + 2, else 3 WI - 1), so two lane bases shifted by
            // h times those make every read "base + compile-time immediate".  Pads (k > 80) re-use tap 80's value.
            constexpr int D0 = (2 * WI + 2) * 4, D1 = (3 * WI - 1) * 4;
            auto locb = [&](int tl, bool &valid_, int &yy_, int &xx_, int &ba_, int &bb_) {
                int m_ = tl * 32 + l31;
                valid_ = m_ < npix;
                if (!valid_) m_ = npix - 1;
                yy_ = (int)__umulhi((uint32_t)m_, a.magic_wo);
                xx_ = m_ - yy_ * a.Wo;
                const int base = (2 * yy_ * ROWF + 2 * xx_) * 4;
                ba_ = base + h * D0;
                bb_ = base + h * D1;
            };
            auto rdb = [&](int ba_, int bb_, int i, int j) {
                const int k = 16 * i + j; // the h = 0 lanes' tap; h = 1 lanes read k + 8 through their shifted base
                if (k + 8 > 80) { // i = 5: only k = 80 is real (h = 0, j = 0); every other slot meets a zero weight
                    const int off80 = ((80 / 3) * WI + 80 % 3) * 4;
                    return *reinterpret_cast<const float *>(tile + (ba_ - h * D0) + off80);
                }
                const int off = ((k / 3) * WI + k % 3) * 4;
                return *reinterpret_cast<const float *>(tile + ((k % 3 == 0) ? ba_ : bb_) + off);
            };
            int yy, xx, ba, bb;
            bool valid;
            locb(cw, valid, yy, xx, ba, bb);
            float raw[F9B_NM][8];
#pragma unroll
            for (int i = 0; i < F9B_NM - 1; ++i)
#pragma unroll
                for (int j = 0; j < 8; ++j) raw[i][j] = rdb(ba, bb, i, j);
            raw[F9B_NM - 1][0] = rdb(ba, bb, F9B_NM - 1, 0);
            for (int tl = cw; tl < ntiles; tl += 4) {
                int yy2, xx2, ba2, bb2;
                bool valid2;
                locb(tl + 4 < ntiles ? tl + 4 : tl, valid2, yy2, xx2, ba2, bb2);
                f32x16 acc;
#pragma unroll
                for (int i = 0; i < F9B_NM; ++i) {
                    bf16x8 bop;
#pragma unroll
                    for (int j = 0; j < 8; j += 2) {
                        const float v0 = raw[i][i == F9B_NM - 1 ? 0 : j], v1 = raw[i][i == F9B_NM - 1 ? 0 : j + 1];
                        const bf16x2 two = __builtin_convertvector(f32x2{v0, v1}, bf16x2);
                        bop[j] = two[0];
                        bop[j + 1] = two[1];
                    }
                    // the next tile's operands of this group are issued as soon as its registers are free
                    if (i < F9B_NM - 1) {
#pragma unroll
                        for (int j = 0; j < 8; ++j) raw[i][j] = rdb(ba2, bb2, i, j);
                    } else {
                        raw[i][0] = rdb(ba2, bb2, i, 0);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wfrag[i], bop, i == 0 ? bias16 : acc, 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
                if (valid) {
                    vu2 *o = reinterpret_cast<vu2 *>(a.out) + ((size_t)(b * a.Ho + y0 + yy) * a.Wo + xx) * 8 + h;
#pragma unroll
                    for (int g4 = 0; g4 < 4; ++g4) {
                        vu2 pk;
#pragma unroll
                        for (int d = 0; d < 2; ++d) {
                            const bf16x2 two = __builtin_convertvector(f32x2{acc[g4 * 4 + 2 * d], acc[g4 * 4 + 2 * d + 1]}, bf16x2);
                            uint32_t bits = __builtin_bit_cast(uint32_t, two);
                            if (RELU)
                                bits = __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(s16x2, bits), s16x2{0, 0}));
                            pk[d] = bits;
                        }
                        o[2 * g4] = pk;
                    }
                }
                valid = valid2; yy = yy2; xx = xx2; ba = ba2; bb = bb2;
            }
            return;
        }
        auto locate = [&](int tl, int &m_, bool &valid_, int &yy_, int &xx_, int &b1_, int &bw_) {
            m_ = tl * 32 + l31;
            valid_ = m_ < npix;
            if (!valid_) m_ = npix - 1;
            yy_ = (int)__umulhi((uint32_t)m_, a.magic_wo);
            xx_ = m_ - yy_ * a.Wo;
            const int base = 2 * yy_ * ROWF + 2 * xx_;
            b1_ = (base + h) * 4;             // pairs inside a row: k+1 is the next float
            bw_ = (base + (WI - 2) * h) * 4;  // pairs across a row-block boundary: k+1 is WI-2 floats on
        };
        auto rd = [&](int b1_, int bw_, int j) {
            const int k0 = 2 * j;
            const int off = ((k0 / 3) * WI + k0 % 3) * 4;
            return *reinterpret_cast<const float *>(tile + ((k0 % 3 != 2) ? b1_ : bw_) + off);
        };
        int m, yy, xx, b1, bw;
        bool valid;
        locate(cw, m, valid, yy, xx, b1, bw);
        float raw[F9_NP];
#pragma unroll
        for (int j = 0; j < F9_AHEAD; ++j) raw[j] = rd(b1, bw, j);
        for (int tl = cw; tl < ntiles; tl += 4) {
            int m2, yy2, xx2, b12, bw2;
            bool valid2;
            locate(tl + 4 < ntiles ? tl + 4 : tl, m2, valid2, yy2, xx2, b12, bw2);
            f32x16 acc;
#pragma unroll
            for (int j = 0; j < F9_NP; ++j) {
                if (j + F9_AHEAD < F9_NP) raw[j + F9_AHEAD] = rd(b1, bw, j + F9_AHEAD);
                else raw[j + F9_AHEAD - F9_NP] = rd(b12, bw2, j + F9_AHEAD - F9_NP);
                __builtin_amdgcn_sched_barrier(0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wreg[j], raw[j], j == 0 ? bias16 : acc, 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
            if (valid) {
                const int y = y0 + yy;
                if (OUT_BF16) {
                    vu2 *o = reinterpret_cast<vu2 *>(a.out) + ((size_t)(b * a.Ho + y) * a.Wo + xx) * 8 + h;
#pragma unroll
                    for (int g4 = 0; g4 < 4; ++g4) {
                        vu2 pk;
#pragma unroll
                        for (int d = 0; d < 2; ++d) {
                            const bf16x2 two = __builtin_convertvector(f32x2{acc[g4 * 4 + 2 * d], acc[g4 * 4 + 2 * d + 1]}, bf16x2);
                            uint32_t bits = __builtin_bit_cast(uint32_t, two);
                            if (RELU)
                                bits = __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(s16x2, bits), s16x2{0, 0}));
                            pk[d] = bits;
                        }
                        o[2 * g4] = pk;
                    }
                } else { // buffer stores: lane byte offset + a wave-uniform row offset in an SGPR (no vector address chain)
                    const int chw = a.Ho * a.Wo;
                    const unsigned boff = (unsigned)(((b * 32 + 4 * h) * a.Ho + y) * a.Wo + xx) * 4u; // host: B*32*Ho*Wo < 2^30
                    uint32_t gbits = 0u;
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        float v = acc[r];
                        if (RELU) v = __int_as_float(max(__float_as_int(v), 0)); // one v_max_i32: negative floats are negative ints
                        vmax = fmaxf(vmax, fabsf(v));
                        if (RELU) gbits |= ((__float_as_uint(v) + 0x7fffffffu) >> 31) << ((r & 3) + 8 * (r >> 2)); // v >= +0: positive <=> bits >= 1
                        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rs_out, boff,
                                                              (unsigned)(((r & 3) + 8 * (r >> 2)) * chw) * 4u, 0);
                    }
                    if (RELU && a.bits_out != nullptr) { // the two half-waves hold the two channel sets of the same 32 pixels
                        const uint32_t mine = gbits << (4 * h);
                        const uint32_t word = mine | (uint32_t)__shfl_xor((int)mine, 32, 64);
                        if (h == 0) a.bits_out[(size_t)(b * a.Ho + y) * a.Wo + xx] = word;
                    }
                }
            }
            m = m2; valid = valid2; yy = yy2; xx = xx2; b1 = b12; bw = bw2;
        }
    };
    __syncthreads();
    for (int item = first; item < n; item += 2 * NG) {
        compute(item, 0);
        __syncthreads();
        if (item + NG >= n) break;
        compute(item + NG, 1);
        __syncthreads();
    }
    if (!OUT_BF16 && a.amax_out != nullptr) { // one global atomic per workgroup: the last of the 4 compute waves sends the maximum on
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) vmax = fmaxf(vmax, __shfl_xor(vmax, o, 64));
        uint32_t *lred = reinterpret_cast<uint32_t *>(lbias) + 32;
        if (lane == 0) {
            atomicMax(lred, __float_as_uint(vmax));
            if (atomicAdd(lred + 1, 1u) == 3u) atomicMax(a.amax_out + grp, lred[0]);
        }
    }
}

template <int WI, bool OUT_BF16, bool RELU>
static int launch_first9_t(First9Args &a, int cus, hipStream_t s, const char *name)
{
    constexpr int ROWF = 9 * WI;
    const int bufwords = (2 * a.RT + 2) * ROWF;
    const size_t lds = (size_t)2 * bufwords * 4 + 34 * 4;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_conv_first9<WI, OUT_BF16, RELU>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return fail(AUGCL_E_LAUNCH, "%s: cannot set dynamic LDS %zu: %s", name, lds, hipGetErrorString(e));
    int grid = a.n_items < cus ? a.n_items : cus;
    if (a.dual) grid &= ~1;
    hipLaunchKernelGGL((k_conv_first9<WI, OUT_BF16, RELU>), dim3(grid), dim3(512), lds, s, a);
    return check_launch(name);
}

// returns AUGCL_OK (or a negative error) after launching, or +1 when the shape is not covered (the caller falls back
// to the generic kernel)
int conv_first9_try(const float *in, const float *w, const float *bias, void *out, int B, int C, int Hi, int Wi, int stride,
                    float scale, int relu, bool out_bf16, hipStream_t s, const char *name, const float *w2,
                    const float *bias2, float *amax_out, void *gate_bits_out)
{
    if (C != 9 || Wi != 84 || stride != 2 || Hi < 3) return 1;
    if (w2 != nullptr && (B & 1)) return 1;
    if (!out_bf16) { // float32 activations: the three-piece bf16 kernel (encoder_first_x3.hip) where it covers the shape
        const int rc = conv_first9_x3_try(in, w, bias, static_cast<float *>(out), B, C, Hi, Wi, stride, scale, relu, s, name, w2, bias2,
                                          amax_out, gate_bits_out);
        if (rc <= 0) return rc;
    }
    if ((size_t)B * 32 * ((Hi - 3) / 2 + 1) * ((Wi - 3) / 2 + 1) >= ((size_t)1 << 30)) return 1; // 32-bit byte offsets in the epilogue
    if ((reinterpret_cast<uintptr_t>(in) & 15u) != 0 || ((size_t)Hi * Wi) % 4 != 0) return 1;
    First9Args a{};
    a.in = in; a.w = w; a.bias = bias; a.out = out;
    a.w2 = w2; a.bias2 = bias2; a.dual = w2 != nullptr;
    a.amax_out = reinterpret_cast<uint32_t *>(amax_out);
    a.bits_out = reinterpret_cast<uint32_t *>(gate_bits_out);
    a.B = B; a.Hi = Hi; a.Ho = (Hi - 3) / 2 + 1; a.Wo = (Wi - 3) / 2 + 1; a.scale = scale;
    // one strip = one round of loads: 9 * (2RT+1) * WI/4 chunks <= 256 * F9_U
    int rt = ((256 * F9_U) / (9 * (Wi / 4)) - 1) / 2;
    if (rt < 1) return 1;
    if (rt > a.Ho) rt = a.Ho;
    // A strip's 32-pixel tiles are dealt to the 4 compute waves, so a strip costs ceil(tiles / 4) tile-times: pick the
    // strip height with the fewest tile-times per image (41 x 41 outputs: 6 rows = 8 tiles = 2 rounds, 7 strips = 14
    // rounds; 7 rows = 9 tiles = 3 rounds, 6 strips = 18), a little per-strip overhead breaking ties.
    int best = rt;
    double best_cost = 1e30;
    for (int r = 1; r <= rt; ++r) {
        const int S = (a.Ho + r - 1) / r, last = a.Ho - (S - 1) * r;
        const double cost = (S - 1) * (double)((((r * a.Wo + 31) / 32) + 3) / 4) + (double)((((last * a.Wo + 31) / 32) + 3) / 4) + 0.15 * S;
        if (cost <= best_cost) { best_cost = cost; best = r; }
    }
    a.RT = best;
#ifdef AUGCL_TUNING // tools/tune_f9.py, tuning build only
    if (const char *e = getenv("AUGCL_F9_RT")) a.RT = std::max(1, std::min(rt, atoi(e)));
#endif
    a.S = (a.Ho + a.RT - 1) / a.RT;
    a.n_items = B * a.S;
    const int rows_last = a.Ho - (a.S - 1) * a.RT;
    a.magic_wo = (uint32_t)(0x100000000ull / (unsigned)a.Wo) + 1u;
    a.magic_run_full = (uint32_t)(0x100000000ull / (unsigned)((2 * a.RT + 1) * (Wi / 4))) + 1u;
    a.magic_run_last = (uint32_t)(0x100000000ull / (unsigned)((2 * rows_last + 1) * (Wi / 4))) + 1u;
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) {
        int v = 0;
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) cus = v;
    }
    if (out_bf16) return relu ? launch_first9_t<84, true, true>(a, cus, s, name) : launch_first9_t<84, true, false>(a, cus, s, name);
    return relu ? launch_first9_t<84, false, true>(a, cus, s, name) : launch_first9_t<84, false, false>(a, cus, s, name);
}

} // namespace augcl
