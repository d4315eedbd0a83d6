// encoder_kernels.hip -- the shared conv encoder (src/algorithms/modules.py:145-165) on CDNA4 matrix cores.
//
// Every 3x3 convolution of the stack (forward, backward-data, backward-weights) is a dense
// contraction on v_mfma_f32_32x32x2_f32: exact f32 products / f32 accumulation (bitwise a k-ordered
// fmaf chain), so the stack meets the 1e-5 parity budget while running on the MFMA pipe and leaving
// the VALU free for staging and epilogues.
//
//   forward (32->32):   D[co][m] = sum_k W[co][k] * X[k][m],   k = (ci,ky,kx), m = output pixel
//     A operand = weights, held in 144 VGPRs per lane for the whole launch (lane l: row co = l&31,
//     k-half h = l>>5 takes input channels 16h..16h+15);  B operand = pixels, one ds_read_b32 per MFMA
//     out of an LDS tile [32 ci][RT+2 rows][Wv] staged once per (image, row-strip).  The accumulator
//     comes out with PIXELS on the lanes, so each store instruction writes two 128-B runs of one
//     channel row: coalesced NCHW stores, bias / ReLU / ReLU-mask fused in the epilogue.
//   backward-data:      the same kernel on the zero-padded dZ with W'[ci][co][ky][kx] = W[co][ci][2-ky][2-kx]
//     (gathered straight from the forward weights) and the ReLU mask (a_prev > 0) in the epilogue.
//   backward-weights:   D_t[co][ci] = sum_p dZ[co][p] * A[ci][p + tap_t] for the 9 taps (144 accumulators),
//     persistent workgroups accumulate over their share of (image, strip) items in registers, write
//     one partial slab each; a second kernel sums the slabs in a FIXED order (no atomics: bitwise
//     reproducible gradients, SURVEY.md section 7 "hard parts").
//   first layer (C->32, stride 2, fused x/255): same scheme with the input channels split in two halves
//     of ceil(C/2) (one zero plane pads an odd C).
#include "augcl_common.h"

#include <algorithm>

namespace augcl {

typedef float f32x16 __attribute__((ext_vector_type(16)));

// words per channel plane of k_conv32's LDS tiles: 2 buffers x 32 planes x 624 words = 156 KB of the CU's 160 KB
constexpr int kConvPlane = 624;

__device__ __forceinline__ float div255e(float x)
{
    const float r = 1.0f / 255.0f;
    const float q = x * r;
    return __builtin_fmaf(__builtin_fmaf(-q, 255.0f, x), r, q); // == x / 255.0f for every finite x (see aug_kernels)
}

__device__ __forceinline__ int acc_row(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

// =========================================================================================
// 32 -> 32 channel 3x3 conv, stride 1.   MODE 0: forward (valid conv, + bias, optional ReLU)
//                                         MODE 1: backward-data (full conv of dZ, * (a_prev > 0))
// in : (B,32,Hi,Wi)   out: (B,32,Ho,Wo) with Ho = Hi + 2P - 2, P = (MODE ? 2 : 0)
// One workgroup (4 waves) per item = (image, strip of RT output rows); persistent over items.
// =========================================================================================
struct ConvArgs {
    const float *in;     // forward: a_{l-1} (post-ReLU);  bwd-data: dZ_l
    const float *w;      // (32,32,3,3) forward weights
    const float *bias;   // (32) or null                       [MODE 0]
    const float *mask;   // a_{l-1} (B,32,Ho,Wo) or null       [MODE 1]
    float *out;
    int B, Hi, Wi, Ho, Wo, RT, S; // S strips per image
    int relu;                     // MODE 0: apply ReLU to the output
    int n_items;                  // B * S
    // two-weight-set forward (two encoders with the same shapes in one launch): images [B/2, B) use w2 / bias2;
    // even workgroups serve the first half, odd ones the second, so a workgroup still holds ONE weight set
    const float *w2, *bias2;
    int dual;
    uint32_t magic_wi; // floor(2^32 / Wi) + 1 (MODE 1: row of a run element)
};

// ---- global -> LDS staging of contiguous runs ------------------------------------------------------
// A wave copies `run` floats (4-byte aligned in global, 16-byte aligned in LDS) of TWO channels at once:
// up to 2*UN unaligned dwordx4 loads per lane are issued before the first LDS write, so one batch covers
// 2 x 256*UN floats with a single exposed memory latency (the naive `d[j] = g[j]` loop paid one latency
// per 64 floats and made the MFMA kernels staging-bound).
typedef float f32x4ua __attribute__((ext_vector_type(4), aligned(4)));

template <int UN, bool VECST = true, typename XF>
__device__ __forceinline__ void wave_copy_runs2(const float *__restrict__ g0, const float *__restrict__ g1,
                                                float *__restrict__ d0, float *__restrict__ d1, int run, int lane,
                                                XF xf)
{
    const int nfull = run >> 2;
    for (int c0 = 0; c0 < nfull; c0 += 64 * UN) {
        f32x4ua v0[UN], v1[UN];
#pragma unroll
        for (int u = 0; u < UN; ++u) { // branch-free: chunks past the end are clamped to the last one (idempotent)
            const int c = min(c0 + u * 64 + lane, nfull - 1);
            v0[u] = *reinterpret_cast<const f32x4ua *>(g0 + 4 * c);
            v1[u] = *reinterpret_cast<const f32x4ua *>(g1 + 4 * c);
        }
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const int c = min(c0 + u * 64 + lane, nfull - 1);
            {
                if (VECST) { // destination 16-byte aligned: one ds_write_b128
                    *reinterpret_cast<float4 *>(d0 + 4 * c) = make_float4(xf(v0[u].x), xf(v0[u].y), xf(v0[u].z), xf(v0[u].w));
                    *reinterpret_cast<float4 *>(d1 + 4 * c) = make_float4(xf(v1[u].x), xf(v1[u].y), xf(v1[u].z), xf(v1[u].w));
                } else { // odd plane strides (bank-conflict-free column reads): dword stores
                    d0[4 * c] = xf(v0[u].x); d0[4 * c + 1] = xf(v0[u].y); d0[4 * c + 2] = xf(v0[u].z); d0[4 * c + 3] = xf(v0[u].w);
                    d1[4 * c] = xf(v1[u].x); d1[4 * c + 1] = xf(v1[u].y); d1[4 * c + 2] = xf(v1[u].z); d1[4 * c + 3] = xf(v1[u].w);
                }
            }
        }
    }
    const int t = (nfull << 2) + lane; // < 4 leftover floats
    if (t < run) {
        d0[t] = xf(g0[t]);
        d1[t] = xf(g1[t]);
    }
}

// NCH channels (stride cs in global memory, stride plane in LDS) of one contiguous run each, ALL loads of the batch in
// flight before the first LDS write: one exposed memory latency for the whole set (8 channels x UN dwordx4 per lane).
// Destination planes have odd strides: dword LDS stores.
template <int UN, int NCH, bool VECST = false>
__device__ __forceinline__ void wave_copy_runs_n(const float *__restrict__ g, size_t cs, float *__restrict__ d, int plane,
                                                 int run, int lane)
{
    const int nfull = run >> 2;
    for (int c0 = 0; c0 < nfull; c0 += 64 * UN) {
        f32x4ua v[NCH][UN];
#pragma unroll
        for (int ch = 0; ch < NCH; ++ch)
#pragma unroll
            for (int u = 0; u < UN; ++u) // branch-free: chunks past the end are clamped to the last one (idempotent)
                v[ch][u] = *reinterpret_cast<const f32x4ua *>(g + ch * cs + 4 * min(c0 + u * 64 + lane, nfull - 1));
#pragma unroll
        for (int ch = 0; ch < NCH; ++ch)
#pragma unroll
            for (int u = 0; u < UN; ++u) {
                float *o = d + ch * plane + 4 * min(c0 + u * 64 + lane, nfull - 1);
                if (VECST) { // 16-byte aligned planes: one ds_write_b128
                    *reinterpret_cast<float4 *>(o) = make_float4(v[ch][u].x, v[ch][u].y, v[ch][u].z, v[ch][u].w);
                } else {
                    o[0] = v[ch][u].x; o[1] = v[ch][u].y; o[2] = v[ch][u].z; o[3] = v[ch][u].w;
                }
            }
    }
    const int t = (nfull << 2) + lane; // < 4 leftover floats per channel
    if (t < run) {
#pragma unroll
        for (int ch = 0; ch < NCH; ++ch) d[ch * plane + t] = g[ch * cs + t];
    }
}

struct XfScale { // x / scale with the exact 3-op path for 255
    float scale;
    __device__ __forceinline__ float operator()(float v) const
    {
        return scale == 1.0f ? v : (scale == 255.0f ? div255e(v) : v / scale);
    }
};

struct XfId {
    __device__ __forceinline__ float operator()(float v) const { return v; }
};

__device__ __forceinline__ int round4(int v) { return (v + 3) & ~3; }

// k_conv32: 512 threads = 4 COMPUTE waves (one per SIMD: they own the MFMA pipe and keep the 144 weights
// in registers) + 4 LOADER waves that stage the NEXT item's tile into the other half of a double-buffered
// LDS image while the compute waves run the current one.  One workgroup per CU, one barrier per item.
// (Two independent 256-thread workgroups per CU ran in lockstep -- same program, same start -- so their
// staging and MFMA phases never overlapped: ablation showed t = t_mfma + t_rest, not max.)
// LDS image per buffer: 32 planes of `plane` words.  Forward: a plane mirrors the channel's memory rows (row stride Wi).
// Backward-data: rows are 2 words further apart and those 2 words are zero (right pad of one row = left pad of the next),
// dZ rows outside the image are zero rows: the full correlation reads its padding from the tile, nothing is masked in
// front of the MFMAs (per-lane selects there cost 92.6 instead of 70.5 cycles per MFMA -- see the loop below).
template <int MODE, int ABL = 0> // ABL: developer ablation switch (tools/ebench), 0 = product
__global__ __launch_bounds__(512, 2) void k_conv32(ConvArgs a)
{
    extern __shared__ __attribute__((aligned(16))) float lds_raw[];
    constexpr int P = MODE ? 2 : 0;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6); // scalar: role / loop control on the SALU
    const bool loader = wave >= 4;
    const int cw = wave & 3; // index inside the role
    const int l31 = lane & 31, h = lane >> 5;
    const int Wi = a.Wi;
    // LDS row pitch.  MODE 1 keeps 2 zero words between consecutive dZ rows (the right pad of one row = the left pad of the
    // next) and zero rows outside the image: every tap of every output pixel reads a real word or a zero, nothing is masked.
    const int pitch = MODE ? Wi + 2 : Wi;
    // The channel-plane stride is a COMPILE-TIME constant (the largest plane the LDS budget admits; the host sizes RT so that
    // a strip's rows fit): channel offsets then fold into the 16-bit immediate of every ds_read and the MFMA chain carries
    // no address arithmetic for them -- a vector instruction between a wave's MFMAs costs MFMA time.
    constexpr int plane = kConvPlane;
    constexpr int bufw = 32 * plane; // words per buffer
    constexpr int WROW = 292;            // weight staging row pitch (16-byte aligned rows, conflict-free reads)
    float *lbias = lds_raw + max(2 * bufw, bufw + 32 * WROW);

    // stage item `it` into buffer `bi` (loader waves only): wave cw copies channels 8cw .. 8cw+7, four
    // channel pairs back to back => 24 dwordx4 per lane in flight, one exposed latency per item
    auto stage = [&](int item, int bi, bool first_use) {
        const int b = item / a.S, s = item - b * a.S;
        const int y0 = s * a.RT;
        const int rows = min(a.RT, a.Ho - y0);
        const int r0 = MODE ? max(0, y0 - P) : y0;
        const int r1 = MODE ? min(a.Hi, y0 + rows) : y0 + rows + 2;
        const int run = (r1 - r0) * Wi;
        const float *src = a.in + ((size_t)b * 32 * a.Hi + r0) * Wi;
        const size_t cs = (size_t)a.Hi * Wi;
        float *dst = lds_raw + bi * bufw;
        if (MODE == 0) { // the wave's 8 channels in flight at once: ONE exposed memory latency per strip
            wave_copy_runs_n<3, 8, true>(src + (size_t)(8 * cw) * cs, cs, dst + 8 * cw * plane, plane, run, lane);
            return;
        }
        // MODE 1: tile row j = dZ row (y0 - 2 + j) at words [j * pitch + 2, j * pitch + 2 + Wi) of its channel plane
        const int top = r0 - (y0 - P), nrow = r1 - r0, tile_rows = rows + 2; // rows above / inside / (below) the image
        float *d8 = dst + 8 * cw * plane;
        if (first_use) { // the 2 pad words in front of every row and behind the last: written once per buffer
            for (int i = lane; i < 8 * (a.RT + 3) * 2; i += 64) {
                const int c = i / ((a.RT + 3) * 2), r = i - c * (a.RT + 3) * 2;
                d8[c * plane + (r >> 1) * pitch + (r & 1)] = 0.0f;
            }
        }
        const int bot = tile_rows - top - nrow;
        if (top + bot > 0) { // first / last strip of an image: dZ rows outside it are zero
            for (int i = lane; i < 8 * (top + bot) * Wi; i += 64) {
                const int c = i / ((top + bot) * Wi), rem = i - c * (top + bot) * Wi;
                const int jr = rem / Wi, x = rem - jr * Wi;
                const int j = jr < top ? jr : top + nrow + (jr - top);
                d8[c * plane + j * pitch + 2 + x] = 0.0f;
            }
        }
        // row by row, lanes along x: 8 channels x 8 rows = 64 dwords per lane in flight, then 64 conflict-free ds_write_b32 at
        // row base + immediate -- no per-element address arithmetic (a chunked copy of the contiguous run needs a division
        // per element to find its row, and made the loaders the slower half of the pipeline; walking the (channel, row)
        // pairs as one list without the duplicate rows of a ragged last batch measured 20 % slower: scalar bookkeeping)
        for (int x0 = 0; x0 < Wi; x0 += 64) {
            const int x = min(x0 + lane, Wi - 1); // lanes past the row end repeat its last word
            {
                const float *s8 = src + (size_t)(8 * cw) * cs + x;
                float *dd = d8 + top * pitch + 2 + x;
                for (int rb = 0; rb < nrow; rb += 8) { // ceil(nrow / 8) exposed latencies per strip
                    float v[8][8];
#pragma unroll
                    for (int c = 0; c < 8; ++c)
#pragma unroll
                        for (int r = 0; r < 8; ++r) v[c][r] = s8[c * cs + (size_t)min(rb + r, nrow - 1) * Wi];
#pragma unroll
                    for (int c = 0; c < 8; ++c)
#pragma unroll
                        for (int r = 0; r < 8; ++r) dd[c * plane + min(rb + r, nrow - 1) * pitch] = v[c][r];
                }
            }
        }
        (void)run;
    };

    int it = 0;
    const int grp = a.dual ? (int)(blockIdx.x & 1) : 0, nhalf = a.n_items >> 1;
    const int first = a.dual ? grp * nhalf + (int)(blockIdx.x >> 1) : xcd_slot((int)blockIdx.x, (int)gridDim.x);
    const int NG = a.dual ? (int)(gridDim.x >> 1) : (int)gridDim.x;
    const int n_end = a.dual ? (grp + 1) * nhalf : a.n_items;
    // ABL == 4 (tools/ebench only): compute wave 0 of workgroup 0 logs (event id, shader clock) pairs through the pointer
    // smuggled in a.bias2: 0 tile start, 1 MFMA chain drained, 2 stores issued, 4 before / 5 after the strip's barrier
    unsigned long long *dbg = nullptr;
    int ndbg = 0;
    if (ABL == 4 && blockIdx.x == 0 && wave == 0) dbg = reinterpret_cast<unsigned long long *>(const_cast<float *>(a.bias2));
    auto stamp = [&](int id) {
        if (ABL == 4 && dbg != nullptr && ndbg < 2000) {
            const unsigned long long t = __builtin_amdgcn_s_memtime();
            if (lane == 0) dbg[ndbg] = (t << 4) | (unsigned)id;
            ++ndbg;
        }
    };
    // Prologue: the 4 compute waves copy the 36 KB of weights into LDS with coalesced 16-byte loads (into the region of tile
    // buffer 1, which the loaders first write after the second barrier) WHILE the 4 loader waves already stage the first
    // strip into buffer 0 -- the two memory latencies overlap; then each compute lane picks its 144 weights out of LDS.
    // (Gathered straight from global memory they cost ~4 us per launch: 2304 cache lines per wave.)
    if (!loader) {
        typedef float wf4 __attribute__((ext_vector_type(4)));
        float *wst = lds_raw + bufw;
        const wf4 *w4 = reinterpret_cast<const wf4 *>((a.dual && (blockIdx.x & 1)) ? a.w2 : a.w);
        for (int i = tid; i < 32 * 288 / 4; i += 256) {
            const int row = (4 * i) / 288, col = 4 * i - row * 288;
            *reinterpret_cast<wf4 *>(wst + row * WROW + col) = w4[i];
        }
        const float *bsrc = (a.dual && (blockIdx.x & 1)) ? a.bias2 : a.bias;
        if (tid < 32) lbias[tid] = (MODE == 0 && bsrc) ? bsrc[tid] : 0.0f;
    } else if (first < n_end && ABL != 1) {
        stage(first, 0, true);
    }
    __syncthreads();
    float wreg[144];
    if (!loader) {
        const float *wst = lds_raw + bufw;
        if (MODE == 0) {
            const float *wp = wst + l31 * WROW + h * 144; // contiguous 144 floats
#pragma unroll
            for (int k = 0; k < 144; k += 4) {
                const float4 v = *reinterpret_cast<const float4 *>(wp + k);
                wreg[k] = v.x; wreg[k + 1] = v.y; wreg[k + 2] = v.z; wreg[k + 3] = v.w;
            }
        } else {
#pragma unroll
            for (int cil = 0; cil < 16; ++cil)
#pragma unroll
                for (int t = 0; t < 9; ++t) // W'[ci=l31][co=cil+16h][ky][kx] = W[co][ci][2-ky][2-kx]
                    wreg[cil * 9 + t] = wst[(cil + 16 * h) * WROW + l31 * 9 + (8 - t)];
        }
    }
    f32x16 bias16; // accumulator seed: C operand of a tile's first MFMA
#pragma unroll
    for (int r = 0; r < 16; ++r) bias16[r] = lbias[acc_row(r, h)];
    const int relu_floor = __builtin_amdgcn_readfirstlane(a.relu ? 0 : (int)0x80000000); // opaque: stays ONE v_max_i32
    // raw buffer resources (base, stride 0, no bound, gfx9 dword-3 flags): offsets < 2^32 by the host's size check
    const __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc(a.out, 0, -1, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_mask =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(MODE == 1 && a.mask ? a.mask : a.in), 0, -1, 0x00020000);

    __syncthreads(); // the weights are in registers: buffer 1 may be overwritten
    // The two roles run SEPARATE loops with the same barrier sequence (one per strip): in a shared loop the 144 weight and 16
    // accumulator registers of the compute waves are live across the loaders' branch as well and leave them ~60 registers
    // for loads in flight; on their own the loaders have the whole file (fewer, larger batches = fewer exposed latencies).
    if (loader) {
        for (int item = first; item < n_end; item += NG, ++it) {
            const int next = item + NG;
            if (next < n_end && ABL != 1) stage(next, (it + 1) & 1, it == 0);
            __syncthreads();
        }
        return;
    }
    for (int item = first; item < n_end; item += NG, ++it) {
        {
            const int b = item / a.S, s = item - b * a.S;
            const int y0 = s * a.RT;
            const int rows = min(a.RT, a.Ho - y0); // output rows of this strip
            const int npix = rows * a.Wo;
            const float *lds = lds_raw + (it & 1) * bufw;
            // ---- M-tiles of 32 consecutive strip pixels, round-robin over the 4 compute waves
            const int ntiles = (npix + 31) >> 5;
            float gv[16]; // backward-data: this lane's 16 ReLU gates of the CURRENT tile
            bool gv_ready = false; // (wave-uniform) fetched ahead, in front of the previous tile's stores
            for (int tile = cw; tile < ntiles; tile += 4) {
                if (ABL == 4) stamp(0);
                int m = tile * 32 + l31;
                const bool valid = m < npix;
                if (!valid) m = npix - 1; // read a real pixel, discard the result
                const int yy = m / a.Wo, xx = m - yy * a.Wo;
                // top-left tap of this pixel in the tile: forward -- tile row yy, column xx; backward-data -- dZ row
                // y0 + yy - 2 = tile row yy, padded column xx (the tile's own zero pads stand in for everything outside dZ)
                const float *base = lds + yy * pitch + xx + h * 16 * plane;
                // backward-data: the ReLU gate of the layer below is (a_prev > 0) at this lane's 16 outputs -- the same
                // addresses it will store to.  Issued NOW, consumed after the 144 MFMAs: the latency is free.
                const int chw = a.Ho * a.Wo; // 32-bit offsets: the host checks B*32*Ho*Wo < 2^30
                const int o0 = (b * 32 * a.Ho + (y0 + yy)) * a.Wo + xx + 4 * h * chw;
                // one 32-bit lane offset for all 16 rows; the per-row part (rows of co: a multiple of chw) is wave-uniform and
                // goes into the scalar base of each access (global_load / global_store with an SGPR base: no 64-bit vector
                // address arithmetic per access -- the epilogue took ~1040 of a tile's ~11300 cycles with it)
                const unsigned boff = (unsigned)o0 * 4u; // bytes: the host checks B*32*Ho*Wo < 2^30
                // The accumulating waves reach global memory through BUFFER instructions: address = resource base + this
                // lane's 32-bit byte offset + a wave-uniform row offset in an SGPR (`buffer_load / buffer_store_dword v, voff,
                // s[rsrc], soff offen`), i.e. no vector address arithmetic per access.  hipcc's flat-pointer code carried a
                // 64-bit vector address from row to row and selected the ReLU per element -- 6 vector instructions per
                // row, ~1040 of a tile's ~11300 cycles (tools/ebench, per-tile s_memtime stamps).
                auto gates = [&](unsigned bo, float (&g)[16]) {
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        g[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                                                             rs_mask, bo, (unsigned)(((r & 3) + 8 * (r >> 2)) * chw) * 4u, 0));
                };
                if (MODE == 1 && a.mask != nullptr && !gv_ready) gates(boff, gv); // a strip's first tile
                f32x16 acc;
                // explicit software pipeline: the 9 taps of channel cil+1 are in flight while the 9 MFMAs of
                // channel cil issue (one wave per SIMD must never wait on a just-issued ds_read)
                // single ds_read_b32 per tap, kept apart by the volatile access: merged into ds_read2_b32, whose offsets
                // reach only 1 KB, a pair needs its own base register -- one vector add per pair inside the MFMA chain --
                // while a lone ds_read_b32 takes the whole (channel plane + tap) offset as its 16-bit immediate
#define LDSV(p) (*(const volatile __attribute__((address_space(3))) float *)(p))
                float xa[9], xb[9];
                const float *const rb[3] = {base, base + pitch, base + 2 * pitch}; // the three tap rows of this lane's pixel
                constexpr int NC = (ABL == 2 ? 2 : 16);
#pragma unroll
                for (int t = 0; t < 9; ++t) xa[t] = LDSV(rb[t / 3] + (t % 3));
#pragma unroll
                for (int cil = 0; cil < NC; cil += 2) {
#pragma unroll
                    for (int t = 0; t < 9; ++t) xb[t] = LDSV(rb[t / 3] + (cil + 1) * plane + (t % 3));
                    __builtin_amdgcn_sched_barrier(0); // keep the 9 reads AHEAD of the MFMA group (hipcc sinks them)
                    // (backward-data used to select out-of-image taps to zero per MFMA into one temporary: the select's write
                    // waited for the in-flight MFMA that still named that temporary as its operand -- 92.6 instead of 70.5
                    // cycles per MFMA, tools/ebench stamps; with the zero-padded tile there is nothing to select)
#pragma unroll
                    for (int t = 0; t < 9; ++t)
                        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wreg[cil * 9 + t], xa[t], (cil == 0 && t == 0) ? bias16 : acc, 0, 0, 0);
                    if (cil + 2 < NC) {
#pragma unroll
                        for (int t = 0; t < 9; ++t) xa[t] = LDSV(rb[t / 3] + (cil + 2) * plane + (t % 3));
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int t = 0; t < 9; ++t)
                        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wreg[(cil + 1) * 9 + t], xb[t], acc, 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
#undef LDSV
                if (ABL == 4) stamp(acc[0] == 12345.678f ? 3 : 1); // after the MFMA chain has drained (reads acc)
                // Backward-data epilogue in three steps: gate the 16 accumulator rows in place; request the NEXT tile's gates
                // (into the registers the current ones just vacated); store.  The vector-memory counter is in-order and a
                // store is only counted off when memory acknowledges it (longer than a whole tile, tools/ebench stamps):
                // gates requested at their own tile's start sat behind the previous tile's 16 stores, and the wait for
                // them waited for those stores -- 924-1500 cycles per epilogue against the forward's 556.
                if (MODE == 1 && a.mask != nullptr) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[r] = gv[r] > 0.0f ? acc[r] : 0.0f;
                    gv_ready = tile + 4 < ntiles;
                    if (gv_ready) {
                        int m2 = (tile + 4) * 32 + l31;
                        if (m2 >= npix) m2 = npix - 1;
                        const int yy2 = m2 / a.Wo, xx2 = m2 - yy2 * a.Wo;
                        gates((unsigned)((b * 32 * a.Ho + (y0 + yy2)) * a.Wo + xx2 + 4 * h * chw) * 4u, gv);
                    }
                }
                if (valid && (ABL != 3 || acc[0] == 12345.678f)) {
                    // rows in accumulator order: co = 0,1,2,3, 8,9,10,11, ... (+4 h in the lane offset).  Forward ReLU on
                    // the bit pattern is one v_max_i32 (negative floats are negative integers; floor INT_MIN = no ReLU).
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        float v = acc[r];
                        if (MODE == 0) v = __int_as_float(max(__float_as_int(v), relu_floor));
                        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rs_out, boff,
                                                              (unsigned)(((r & 3) + 8 * (r >> 2)) * chw) * 4u, 0);
                    }
                }
                if (ABL == 4) stamp(2);
            }
        }
        if (ABL == 4) stamp(4);
        __syncthreads(); // next tile landed AND current tile's readers are done
        if (ABL == 4) stamp(5);
    }
}

// =========================================================================================
// first layer: C -> 32, 3x3, stride 2, input divided by `scale` (255) at staging.
// channel halves: CH = ceil(C/2); lane half h takes channels h*CH .. h*CH+CH-1 (zero plane if >= C)
// =========================================================================================
struct Conv1Args {
    const float *in; // (B,C,Hi,Wi) raw 0..255
    const float *w;  // (32,C,3,3)
    const float *bias;
    float *out;      // (B,32,Ho,Wo)
    int B, C, Hi, Wi, Ho, Wo, RT, S, relu, n_items, stride;
    float scale;
};

typedef __bf16 bf16x4k __attribute__((ext_vector_type(4)));

template <int CH, bool OUT_BF16 = false> // CH*9 MFMA steps; OUT_BF16: write bf16 NHWC (perf mode) instead of f32 NCHW
__global__ __launch_bounds__(256, 3) void k_conv_first(Conv1Args a)
{
    extern __shared__ __attribute__((aligned(16))) float lds[]; // [2*CH][rows_in][Wi]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, h = lane >> 5;
    const int st = a.stride;
    const int rows_in_max = st * (a.RT - 1) + 3;
    const int plane = rows_in_max * a.Wi;
    float wreg[CH * 9];
#pragma unroll
    for (int cil = 0; cil < CH; ++cil) {
        const int c = cil + CH * h;
#pragma unroll
        for (int t = 0; t < 9; ++t) wreg[cil * 9 + t] = c < a.C ? a.w[((size_t)l31 * a.C + c) * 9 + t] : 0.0f;
    }
    float bias_r[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) bias_r[r] = a.bias ? a.bias[acc_row(r, h)] : 0.0f;

    for (int item = blockIdx.x; item < a.n_items; item += gridDim.x) {
        const int b = item / a.S, s = item - b * a.S;
        const int y0 = s * a.RT;
        const int rows = min(a.RT, a.Ho - y0);
        const int npix = rows * a.Wo;
        const int rows_in = st * (rows - 1) + 3;
        __syncthreads();
        {
            const int run = rows_in * a.Wi;
            const float *src = a.in + (size_t)b * a.C * a.Hi * a.Wi;
            const size_t cs = (size_t)a.Hi * a.Wi;
            const XfScale xf{a.scale};
            for (int c = wave * 2; c < 2 * CH; c += 8) { // channel pairs; planes >= C are zero planes
                const int ca = min(c, a.C - 1), cb = min(c + 1, a.C - 1);
                wave_copy_runs2<2, false>(src + ca * cs + (size_t)st * y0 * a.Wi, src + cb * cs + (size_t)st * y0 * a.Wi,
                                          lds + c * plane, lds + (c + 1) * plane, run, lane, xf);
            }
            __syncthreads();
            for (int c = a.C + wave; c < 2 * CH; c += 4) // pad planes: finite zeros (0 * NaN would poison the MFMA)
                for (int j = lane; j < run; j += 64) lds[c * plane + j] = 0.0f;
        }
        __syncthreads();
        const int ntiles = (npix + 31) >> 5;
        for (int tile = wave; tile < ntiles; tile += 4) {
            int m = tile * 32 + l31;
            const bool valid = m < npix;
            if (!valid) m = npix - 1;
            const int yy = m / a.Wo, xx = m - yy * a.Wo;
            const float *base = lds + (st * yy * a.Wi + st * xx) + h * CH * plane;
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = bias_r[r];
#pragma unroll
            for (int cil = 0; cil < CH; ++cil) {
                const float *pc = base + cil * plane;
#pragma unroll
                for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx)
                        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wreg[cil * 9 + ky * 3 + kx], pc[ky * a.Wi + kx], acc,
                                                                   0, 0, 0);
            }
            if (valid && OUT_BF16) {
                bf16x4k *o = reinterpret_cast<bf16x4k *>(a.out) + ((size_t)(b * a.Ho + y0 + yy) * a.Wo + xx) * 8;
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) { // lane holds channels 8*g4 + 4*h + e
                    bf16x4k pk;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float v = acc[g4 * 4 + e];
                        if (a.relu) v = fmaxf(v, 0.0f);
                        pk[e] = (__bf16)v;
                    }
                    o[2 * g4 + h] = pk;
                }
            } else if (valid) {
                const size_t o0 = ((size_t)b * 32 * a.Ho + (y0 + yy)) * a.Wo + xx;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    float v = acc[r];
                    if (a.relu) v = fmaxf(v, 0.0f);
                    a.out[o0 + (size_t)acc_row(r, h) * a.Ho * a.Wo] = v;
                }
            }
        }
    }
}

// =========================================================================================
// backward-weights, 32 -> 32 layer:  dW[co][ci][t] = sum_{b,y,x} dZ[b,co,y,x] * A[b,ci,y+ky,x+kx]
//                                     db[co]       = sum dZ[b,co,y,x]
// LDS: A tile [32][RT+2][Wi] (plane stride odd: lanes = channels read one pixel -> conflict-free)
//      Z tile [32][RT][Wo]    (plane stride odd)
// each wave walks pixel PAIRS (lane half h = pixel p+h); 9 MFMAs per pair into 9 accumulator tiles.
// =========================================================================================
struct WgradArgs {
    const float *act; // A = a_{l-1} (B,32,Hi,Wi)
    const float *dz;  // (B,32,Ho,Wo)
    float *partial;   // [gridDim.x][32*288 + 32]
    int B, Hi, Wi, Ho, Wo, RT, S, n_items;
};

constexpr int WG_SLAB = 32 * 288 + 32;

// One 256-thread workgroup per CU (the 144 accumulators + operands need the whole 512-register budget, so
// there is no room for loader waves as in k_conv32).  Staging is REGISTER double-buffered instead: the global
// loads of item i+1 (<= 24 dwordx4 + 16 tail dwords per lane) are issued before the MFMA loop of item i, land
// while it runs, and are written to the other LDS buffer afterwards -- one barrier per item.
// The host picks RT so that a channel's A run fits 2 and its dZ run 1 dwordx4 per lane (see the launcher).
struct WgPrefetch { // lane-contiguous dwords: conflict-free ds_write_b32 into the odd-stride planes
    float a[8][8], z[8][4];
};

__device__ __forceinline__ float bf16lo(uint32_t d) { return __uint_as_float(d << 16); }
__device__ __forceinline__ float bf16hi(uint32_t d) { return __uint_as_float(d & 0xffff0000u); }

// BF16IN: activations / dZ are bf16 NHWC (perf mode): wave w loads 16-byte chunk w (channels 8w..8w+7) of every
// pixel of the runs -- fully coalesced -- and scatters the 8 values (widened to f32) into its 8 channel planes.
// SPLIT (f32 product path): 512 threads = the 4 accumulating waves + 4 LOADER waves that copy the next strip straight into
// the other LDS buffer with dwordx4 loads while the MFMAs run -- without the 96 prefetch registers the accumulating waves
// fit the 256-register budget of two waves per SIMD, and they no longer spend ~400 load / store instructions per strip
// next to their MFMAs (ablation: staging cost 14 % of the launch).  One barrier per strip, as in k_conv32.
template <int ABL = 0, bool BF16IN = false, bool SPLIT = false>
__global__ __launch_bounds__(SPLIT ? 512 : 256, 1) void k_wgrad32(WgradArgs a)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave_all = __builtin_amdgcn_readfirstlane(tid >> 6); // scalar: loop bounds stay on the SALU
    const bool loader = SPLIT && wave_all >= 4;
    const int wave = SPLIT ? (loader ? 8 : wave_all) : wave_all; // accumulating-wave index 0..3 (loaders match none)
    const int lw = wave_all & 3;                                   // loader index
    const int l31 = lane & 31, h = lane >> 5;
    const int planeA = ((a.RT + 2) * a.Wi) | 1;
    const int planeZ = (a.RT * a.Wo + 1) | 1; // word RT*Wo of every dZ plane is never written by data: it stays ZERO
    const int bufw = 32 * planeA + 32 * planeZ;
    // Pair table behind the two tile buffers: for pixel pair pr and lane half hh, the byte offsets of pixel p = 2 pr + hh in a
    // dZ plane and of its top-left tap in an A plane -- for a full strip and for the image's last (shorter) strip.  A pixel
    // past the strip points at the zero word / at tap 0, so the MFMA loop needs no bounds logic and no select: every vector
    // instruction a wave issues between its MFMAs costs MFMA time (tools/ebench k_wg_pair: the ~12-instruction address walk
    // alone took the loop from 576 to 650 cycles per pair; the LDS reads themselves are free).
    const int ntab = 2 * (((a.RT * a.Wo + 1) >> 1) + 8); // entries per table (8 pairs of padding: the loop reads ahead)
    int2 *tab = reinterpret_cast<int2 *>(lds + 2 * bufw);
    {
        const int rows_last = a.Ho - (a.S - 1) * a.RT;
        for (int i = tid; i < 2 * ntab; i += (SPLIT ? 512 : 256)) {
            const int set = i >= ntab, e = i - set * ntab;
            const int npix_s = (set ? rows_last : a.RT) * a.Wo;
            const int p = e; // = 2 pr + hh
            const bool valid = p < npix_s;
            const int pc = valid ? p : 0;
            const int yy = pc / a.Wo, xx = pc - yy * a.Wo;
            tab[i] = make_int2((valid ? p : a.RT * a.Wo) * 4, (yy * a.Wi + xx) * 4);
        }
        if (tid < 64) lds[(tid >> 5) * bufw + 32 * planeA + (tid & 31) * planeZ + a.RT * a.Wo] = 0.0f; // the zero words
    }

    f32x16 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;
    float dbacc = 0.0f;

    // wave w owns channels 8w .. 8w+7 of both tiles
    auto issue = [&](int item, WgPrefetch &pf) {
        const int b = item / a.S, s = item - b * a.S;
        const int y0 = s * a.RT;
        const int rows = min(a.RT, a.Ho - y0);
        const int runA = (rows + 2) * a.Wi, runZ = rows * a.Wo;
        if (BF16IN) {
            const uint4 *ga = reinterpret_cast<const uint4 *>(a.act) + ((size_t)(b * a.Hi + y0) * a.Wi) * 4 + wave;
            const uint4 *gz = reinterpret_cast<const uint4 *>(a.dz) + ((size_t)(b * a.Ho + y0) * a.Wo) * 4 + wave;
            uint4 *pa = reinterpret_cast<uint4 *>(&pf.a[0][0]); // 8 chunks (A run <= 512 px) + 4 chunks (dZ run <= 256 px)
            uint4 *pz = reinterpret_cast<uint4 *>(&pf.a[4][0]);
#pragma unroll
            for (int u = 0; u < 8; ++u) pa[u] = ga[(size_t)min(lane + 64 * u, runA - 1) * 4];
#pragma unroll
            for (int u = 0; u < 4; ++u) pz[u] = gz[(size_t)min(lane + 64 * u, runZ - 1) * 4];
            return;
        }
        const float *src = a.act + ((size_t)b * 32 * a.Hi + y0) * a.Wi; // rows contiguous in NCHW
        const float *sz = a.dz + ((size_t)b * 32 * a.Ho + y0) * a.Wo;
        const size_t csA = (size_t)a.Hi * a.Wi, csZ = (size_t)a.Ho * a.Wo;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const float *ga = src + (8 * wave + k) * csA, *gz = sz + (8 * wave + k) * csZ;
            // branch-free: indices past the run are clamped to its last element (a duplicate load / store of
            // the same value to the same address) -- predicated loads compile to one exec-masked block each
#pragma unroll
            for (int u = 0; u < 8; ++u) pf.a[k][u] = ga[min(lane + 64 * u, runA - 1)];
#pragma unroll
            for (int u = 0; u < 4; ++u) pf.z[k][u] = gz[min(lane + 64 * u, runZ - 1)];
        }
    };
    auto commit = [&](int item, int bi, const WgPrefetch &pf) {
        const int s = item % a.S;
        const int rows = min(a.RT, a.Ho - s * a.RT);
        const int runA = (rows + 2) * a.Wi, runZ = rows * a.Wo;
        float *ldsA = lds + bi * bufw, *ldsZ = ldsA + 32 * planeA;
        if (BF16IN) {
            const uint4 *pa = reinterpret_cast<const uint4 *>(&pf.a[0][0]);
            const uint4 *pz = reinterpret_cast<const uint4 *>(&pf.a[4][0]);
            float *da = ldsA + 8 * wave * planeA, *dz = ldsZ + 8 * wave * planeZ;
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int p = min(lane + 64 * u, runA - 1);
                const uint32_t d[4] = {pa[u].x, pa[u].y, pa[u].z, pa[u].w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    da[(2 * e) * planeA + p] = bf16lo(d[e]);
                    da[(2 * e + 1) * planeA + p] = bf16hi(d[e]);
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int p = min(lane + 64 * u, runZ - 1);
                const uint32_t d[4] = {pz[u].x, pz[u].y, pz[u].z, pz[u].w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    dz[(2 * e) * planeZ + p] = bf16lo(d[e]);
                    dz[(2 * e + 1) * planeZ + p] = bf16hi(d[e]);
                }
            }
            return;
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            float *da = ldsA + (8 * wave + k) * planeA, *dz = ldsZ + (8 * wave + k) * planeZ;
#pragma unroll
            for (int u = 0; u < 8; ++u) da[min(lane + 64 * u, runA - 1)] = pf.a[k][u];
#pragma unroll
            for (int u = 0; u < 4; ++u) dz[min(lane + 64 * u, runZ - 1)] = pf.z[k][u];
        }
    };

    // SPLIT: loader wave lw copies channels 8 lw .. 8 lw + 7 of both tiles, channel pairs back to back
    auto stage = [&](int item, int bi) {
        const int b = item / a.S, s = item - b * a.S;
        const int y0 = s * a.RT;
        const int rows = min(a.RT, a.Ho - y0);
        const int runA = (rows + 2) * a.Wi, runZ = rows * a.Wo;
        const float *src = a.act + ((size_t)b * 32 * a.Hi + y0) * a.Wi;
        const float *sz = a.dz + ((size_t)b * 32 * a.Ho + y0) * a.Wo;
        const size_t csA = (size_t)a.Hi * a.Wi, csZ = (size_t)a.Ho * a.Wo;
        float *ldsA = lds + bi * bufw, *ldsZ = ldsA + 32 * planeA;
        // all 8 channels of BOTH tiles in flight at once (the host caps an A run at 512 and a dZ run at 256 floats: 2 + 1
        // dwordx4 per lane and channel): ONE exposed memory latency per strip instead of eight -- the loaders, walking
        // channel pair after channel pair, needed as long per strip as the MFMA waves
        {
            const float *gA = src + (size_t)(8 * lw) * csA, *gZ = sz + (size_t)(8 * lw) * csZ;
            float *dA = ldsA + 8 * lw * planeA, *dZ = ldsZ + 8 * lw * planeZ;
            const int nfA = runA >> 2, nfZ = runZ >> 2;
            f32x4ua va[8][2], vz[8];
#pragma unroll
            for (int ch = 0; ch < 8; ++ch) {
#pragma unroll
                for (int u = 0; u < 2; ++u) va[ch][u] = *reinterpret_cast<const f32x4ua *>(gA + ch * csA + 4 * min(u * 64 + lane, nfA - 1));
                vz[ch] = *reinterpret_cast<const f32x4ua *>(gZ + ch * csZ + 4 * min(lane, nfZ - 1));
            }
#pragma unroll
            for (int ch = 0; ch < 8; ++ch) {
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    float *o = dA + ch * planeA + 4 * min(u * 64 + lane, nfA - 1);
                    o[0] = va[ch][u].x; o[1] = va[ch][u].y; o[2] = va[ch][u].z; o[3] = va[ch][u].w;
                }
                float *o = dZ + ch * planeZ + 4 * min(lane, nfZ - 1);
                o[0] = vz[ch].x; o[1] = vz[ch].y; o[2] = vz[ch].z; o[3] = vz[ch].w;
            }
            const int tA = (nfA << 2) + lane, tZ = (nfZ << 2) + lane; // < 4 leftover floats per channel
#pragma unroll
            for (int ch = 0; ch < 8; ++ch) {
                if (tA < runA) dA[ch * planeA + tA] = gA[ch * csA + tA];
                if (tZ < runZ) dZ[ch * planeZ + tZ] = gZ[ch * csZ + tZ];
            }
        }
    };

    // ABL == 4 (tools/ebench only): accumulating wave 0 of workgroup 0 logs (event, shader clock) behind the slabs:
    // 0 strip start, 4 before / 5 after the strip's barrier, 6 loop done, 7 slab written
    unsigned long long *dbg = nullptr;
    int ndbg = 0;
    if ((ABL == 4 || ABL == 5) && blockIdx.x == 0 && wave == 0) dbg = reinterpret_cast<unsigned long long *>(a.partial + (size_t)gridDim.x * WG_SLAB + 8 * WG_SLAB);
    auto stamp = [&](int id) {
        if ((ABL == 4 || ABL == 5) && dbg != nullptr && ndbg < 500) {
            const unsigned long long t = __builtin_amdgcn_s_memtime();
            if (lane == 0) dbg[ndbg] = (t << 4) | (unsigned)id;
            ++ndbg;
        }
    };
    WgPrefetch pf;
    int it = 0;
    const int first = xcd_slot((int)blockIdx.x, (int)gridDim.x); // neighbouring strips on one XCD (halo rows hit in L2)
    if (first < a.n_items) {
        if (SPLIT) {
            if (loader) stage(first, 0);
        } else {
            issue(first, pf);
            commit(first, 0, pf);
        }
    }
    __syncthreads();
    if (SPLIT && loader) { // the loaders' own loop (same barriers as the accumulating waves' loop and reduction below): in a
        // shared loop the 144 accumulator registers stay live across the loaders' branch and cap their loads in flight
        for (int item = first; item < a.n_items; item += gridDim.x, ++it) {
            const int next = item + gridDim.x;
            if (next < a.n_items && ABL != 1) stage(next, (it + 1) & 1);
            __syncthreads();
        }
        for (int w = 1; w < 4; ++w) {
            __syncthreads();
            __syncthreads();
        }
        return;
    }
    for (int item = first; item < a.n_items; item += gridDim.x, ++it) {
        const int next = item + gridDim.x;
        if (!SPLIT && next < a.n_items && ABL != 1) issue(next, pf);
        __builtin_amdgcn_sched_barrier(0);
        {
            if (ABL == 4 || ABL == 5) stamp(0);
            const int s = item % a.S;
            const int y0 = s * a.RT;
            const int rows = min(a.RT, a.Ho - y0);
            const int npix = rows * a.Wo;
            const float *ldsA = lds + (it & 1) * bufw, *ldsZ = ldsA + 32 * planeA;
            const int npair = (npix + 1) >> 1;
            // operands of pixel pair pr: z = dZ[co=l31][p], taps A[ci=l31][p + tap]  (p = 2 pr + h); this wave takes pairs
            // wave, wave + 4, ...  Per tap row one single + one PAIR of consecutive floats (what ds_read2_b32 delivers: as a
            // 2-vector the pair keeps adjacent registers; as scalars hipcc shuffled one pair through v_mov after a wait).
            typedef float f32x2w __attribute__((ext_vector_type(2)));
            struct Ops {
                float z, s[3];
                f32x2w d[3];
            };
            const char *zc = reinterpret_cast<const char *>(ldsZ + l31 * planeZ);  // this lane's dZ plane
            const char *ac = reinterpret_cast<const char *>(ldsA + l31 * planeA);  // this lane's A plane
            const int2 *tp = tab + (rows == a.RT ? 0 : ntab) + 2 * wave + h;      // table entry of this lane's current pair
            auto rd1 = [&](const float *pa, const float *pz, Ops &o) { // 4 LDS instructions
                o.z = *pz;
                o.s[0] = pa[0];
                o.d[0] = f32x2w{pa[1], pa[2]};
                o.s[1] = pa[a.Wi];
            };
            auto rd2 = [&](const float *pa, Ops &o) { // 4 LDS instructions
                o.d[1] = f32x2w{pa[a.Wi + 1], pa[a.Wi + 2]};
                o.s[2] = pa[2 * a.Wi];
                o.d[2] = f32x2w{pa[2 * a.Wi + 1], pa[2 * a.Wi + 2]};
            };
            auto mf = [&](const Ops &o, int t) {
                const float xv = t % 3 == 0 ? o.s[t / 3] : o.d[t / 3][t % 3 - 1];
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(o.z, xv, acc[t], 0, 0, 0);
            };
#define AUGCL_SB() __builtin_amdgcn_sched_barrier(0)
            // One pixel pair = 9 MFMAs; the NEXT pair's fetch (2 address adds + the table read for the pair after it, then
            // 4 + 4 operand reads) is issued in three pieces after the 1st, 2nd and 3rd MFMA.  The s_waitcnt drains the
            // in-order LDS counter where everything outstanding is >= 6 MFMAs old, so that hipcc adds no waits of its own
            // inside the groups (it cannot tell how old loop-carried reads are and waited for the ones just issued).
            int2 tn; // table entry of the NEXT pair
            auto pairstep = [&](const Ops &cur, Ops &nxt) {
                __builtin_amdgcn_s_waitcnt(0xC07F); // lgkmcnt(0)
                dbacc += cur.z;
                mf(cur, 0);
                AUGCL_SB();
                const float *pz = reinterpret_cast<const float *>(zc + tn.x), *pa = reinterpret_cast<const float *>(ac + tn.y);
                if (ABL != 5) { // (ABL 5, tools/ebench: the MFMA loop without its operand fetch)
                    tp += 8;
                    tn = *tp;
                }
                AUGCL_SB();
                if (ABL != 2) mf(cur, 1);
                AUGCL_SB();
                if (ABL != 5) rd1(pa, pz, nxt);
                AUGCL_SB();
                if (ABL != 2) mf(cur, 2);
                AUGCL_SB();
                if (ABL != 5) rd2(pa, nxt);
                AUGCL_SB();
                if (ABL != 2) {
#pragma unroll
                    for (int t = 3; t < 9; ++t) mf(cur, t);
                }
                AUGCL_SB();
            };
            Ops o0, o1;
            {
                const int2 t0 = *tp;
                tp += 8;
                tn = *tp;
                rd1(reinterpret_cast<const float *>(ac + t0.y), reinterpret_cast<const float *>(zc + t0.x), o0);
                rd2(reinterpret_cast<const float *>(ac + t0.y), o0);
            }
            const int nw = (npair - wave + 3) >> 2; // pairs of this wave (reads past them land on the table's padding)
            int i = 0;
            for (; i + 1 < nw; i += 2) {
                pairstep(o0, o1);
                pairstep(o1, o0);
            }
            if (i < nw) pairstep(o0, o1);
#undef AUGCL_SB
        }
        if (!SPLIT && next < a.n_items && ABL != 1) commit(next, (it + 1) & 1, pf);
        if ((ABL == 4 || ABL == 5) && !loader) stamp(4);
        __syncthreads();
        if ((ABL == 4 || ABL == 5) && !loader) stamp(5);
    }
    if ((ABL == 4 || ABL == 5) && !loader) stamp(6);
    // ---- fixed-order reduction over the 4 waves through LDS, then one slab per workgroup
    // lane (ci = l31, h) holds dW[co = acc_row(r,h)][ci][t] in acc[t][r]
    float *slab = a.partial + (size_t)blockIdx.x * WG_SLAB;
    for (int w = 1; w < 4; ++w) {
        if (wave == w) {
#pragma unroll
            for (int t = 0; t < 9; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) lds[(t * 16 + r) * 64 + lane] = acc[t][r];
            lds[144 * 64 + lane] = dbacc;
        }
        __syncthreads();
        if (wave == 0) {
#pragma unroll
            for (int t = 0; t < 9; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[t][r] += lds[(t * 16 + r) * 64 + lane];
            dbacc += lds[144 * 64 + lane];
        }
        __syncthreads();
    }
    if (wave == 0) {
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) slab[(size_t)acc_row(r, h) * 288 + l31 * 9 + t] = acc[t][r];
        // db: lanes (co = l31, h = 0/1) hold the sums over even / odd pixels
        const float other = __shfl_xor(dbacc, 32, 64);
        if (h == 0) slab[32 * 288 + l31] = dbacc + other;
    }
    if (ABL == 4) stamp(7);
}

// first layer: dW1[co][c][t] = sum dZ[co][p] * X[c][st*y+ky][st*x+kx]/scale ; n = c*9+t over NT n-tiles of 32
struct Wgrad1Args {
    const float *x;  // (B,C,Hi,Wi) raw
    const float *dz; // (B,32,Ho,Wo)
    float *partial;  // [gridDim.x][32*C*9 + 32]
    int B, C, Hi, Wi, Ho, Wo, RT, S, n_items, stride;
    float scale;
};

template <int NT, bool BF16DZ = false>
__global__ __launch_bounds__(256, (NT <= 3 ? 4 : NT <= 6 ? 2 : 1)) void k_wgrad_first(Wgrad1Args a)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, h = lane >> 5;
    const int st = a.stride;
    const int rows_in_max = st * (a.RT - 1) + 3;
    const int planeX = rows_in_max * a.Wi;
    const int planeZ = (a.RT * a.Wo) | 1;
    float *ldsX = lds;
    float *ldsZ = lds + a.C * planeX;
    const int K = a.C * 9;
    int noff[NT]; // LDS offset of column n = l31 + 32 j: (c, ky, kx); invalid n -> offset 0 with weight masked
    bool nval[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const int n = l31 + 32 * j;
        nval[j] = n < K;
        const int nn = nval[j] ? n : 0;
        const int c = nn / 9, t = nn - c * 9;
        noff[j] = c * planeX + (t / 3) * a.Wi + (t % 3);
    }
    f32x16 acc[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.0f;
    float dbacc = 0.0f;

    for (int item = blockIdx.x; item < a.n_items; item += gridDim.x) {
        const int b = item / a.S, s = item - b * a.S;
        const int y0 = s * a.RT;
        const int rows = min(a.RT, a.Ho - y0);
        const int npix = rows * a.Wo;
        const int rows_in = st * (rows - 1) + 3;
        __syncthreads();
        {
            const int run = rows_in * a.Wi;
            const float *src = a.x + (size_t)b * a.C * a.Hi * a.Wi + (size_t)st * y0 * a.Wi;
            const size_t cs = (size_t)a.Hi * a.Wi;
            const XfScale xf{a.scale};
            for (int c = wave * 2; c < a.C; c += 8) {
                const int cb = min(c + 1, a.C - 1); // odd C: the last pair copies the last channel twice (same bytes)
                wave_copy_runs2<2, false>(src + c * cs, src + cb * cs, ldsX + c * planeX, ldsX + cb * planeX, run, lane, xf);
            }
            if (BF16DZ) { // dZ is bf16 NHWC: wave w takes chunk w (channels 8w..8w+7) of every strip pixel
                const uint4 *gz = reinterpret_cast<const uint4 *>(a.dz) + ((size_t)(b * a.Ho + y0) * a.Wo) * 4 + wave;
                float *dzl = ldsZ + 8 * wave * planeZ;
                for (int p = lane; p < npix; p += 64) {
                    const uint4 v = gz[(size_t)p * 4];
                    const uint32_t d[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        dzl[(2 * e) * planeZ + p] = bf16lo(d[e]);
                        dzl[(2 * e + 1) * planeZ + p] = bf16hi(d[e]);
                    }
                }
            } else {
                const float *sz = a.dz + ((size_t)b * 32 * a.Ho + y0) * a.Wo;
                const size_t csZ = (size_t)a.Ho * a.Wo;
                for (int c = wave * 2; c < 32; c += 8)
                    wave_copy_runs2<3, false>(sz + c * csZ, sz + (c + 1) * csZ, ldsZ + c * planeZ, ldsZ + (c + 1) * planeZ,
                                              npix, lane, XfId());
            }
        }
        __syncthreads();
        const int npair = (npix + 1) >> 1;
        for (int pr = wave; pr < npair; pr += 4) {
            int p = 2 * pr + h;
            const bool valid = p < npix;
            if (!valid) p = npix - 1;
            const int yy = p / a.Wo, xx = p - yy * a.Wo;
            float z = ldsZ[l31 * planeZ + p];
            if (!valid) z = 0.0f;
            dbacc += z;
            const float *px = ldsX + st * yy * a.Wi + st * xx;
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const float xv = nval[j] ? px[noff[j]] : 0.0f;
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(z, xv, acc[j], 0, 0, 0);
            }
        }
    }
    __syncthreads();
    const int slab_sz = 32 * K + 32;
    float *slab = a.partial + (size_t)blockIdx.x * slab_sz;
    for (int w = 1; w < 4; ++w) {
        if (wave == w) {
#pragma unroll
            for (int j = 0; j < NT; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) lds[(j * 16 + r) * 64 + lane] = acc[j][r];
            lds[NT * 16 * 64 + lane] = dbacc;
        }
        __syncthreads();
        if (wave == 0) {
#pragma unroll
            for (int j = 0; j < NT; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[j][r] += lds[(j * 16 + r) * 64 + lane];
            dbacc += lds[NT * 16 * 64 + lane];
        }
        __syncthreads();
    }
    if (wave == 0) {
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const int n = l31 + 32 * j;
            if (n < K)
#pragma unroll
                for (int r = 0; r < 16; ++r) slab[(size_t)acc_row(r, h) * K + n] = acc[j][r];
        }
        const float other = __shfl_xor(dbacc, 32, 64);
        if (h == 0) slab[32 * K + l31] = dbacc + other;
    }
}

// k_wgrad_first_pipe: the same sums for the shapes the training loop uses (Wi a multiple of 4, a strip of at most 128
// output pixels and 1024 16-byte chunks of x), with the staging restructured around ONE exposed memory latency per
// workgroup instead of ~5 per strip: every thread fetches its share of the NEXT strip (x: <= 4 dwordx4; dZ: 16 dwords or
// 2 dwordx4 of bf16) into registers before the MFMAs of the current strip, and writes them to LDS after.  (The loop
// above copies channel pair after channel pair, each waiting for its own loads: 137 us at B = 512 with the MFMA pipe
// busy 25 % of the time.)
constexpr int WG1_XU = 4;   // x chunks (16 B) per thread per strip (bf16 dZ: four workgroups per CU, 2-row strips)
constexpr int WG1_XU32 = 6; // ... with float32 dZ: three workgroups per CU and 3-row strips (7 input rows of 9 x 84: 1323 chunks)
constexpr int WG1_ZP = 128; // dZ pixel slots per channel in the f32 fetch pattern (>= pixels per strip)

// Workgroups per CU: 4 with bf16 dZ; 3 with f32 dZ, whose 16 prefetch registers more do not fit a 128-VGPR budget.
template <int NT, bool BF16DZ>
__global__ __launch_bounds__(256, (BF16DZ ? 4 : 3)) void k_wgrad_first_pipe(Wgrad1Args a, uint32_t magic_full, uint32_t magic_last,
                                                                              uint32_t magic_wo)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6); // scalar: loop control on the SALU
    const int l31 = lane & 31, h = lane >> 5;
    const int st = a.stride;
    const int rows_in_max = st * (a.RT - 1) + 3;
    const int planeX = rows_in_max * a.Wi; // a multiple of 4: 16-byte LDS stores
    const int planeZ = (a.RT * a.Wo + 1) | 1; // word RT*Wo of every dZ plane is never written by data: it stays ZERO
    float *ldsX = lds;
    float *ldsZ = lds + a.C * planeX;
    // pair table (as in k_wgrad32): byte offsets of pixel p = 2 pr + hh in a dZ plane and of its top-left tap in an X plane,
    // for a full strip and for the last strip of an image; pixels past the strip point at the zero word / tap 0
    const int ntab = 2 * (((a.RT * a.Wo + 1) >> 1) + 8);
    int2 *tab = reinterpret_cast<int2 *>(lds + ((a.C * planeX + 32 * planeZ + 1) & ~1));
    {
        const int rows_last = a.Ho - (a.S - 1) * a.RT;
        for (int i = tid; i < 2 * ntab; i += 256) {
            const int set = i >= ntab, e = i - set * ntab;
            const int npix_s = (set ? rows_last : a.RT) * a.Wo;
            const bool valid = e < npix_s;
            const int pc = valid ? e : 0;
            const int yy = pc / a.Wo, xx = pc - yy * a.Wo;
            tab[i] = make_int2((valid ? e : a.RT * a.Wo) * 4, st * (yy * a.Wi + xx) * 4);
        }
        if (tid < 32) ldsZ[tid * planeZ + a.RT * a.Wo] = 0.0f;
    }
    const int K = a.C * 9;
    const int W4 = a.Wi >> 2;
    // LDS offset of column n = l31 + 32 j: (c, ky, kx).  Columns n >= K are never stored: they read a valid address (0) and
    // accumulate garbage of their own (an MFMA output column depends on its own B column only) -- no masking in the loop.
    int noff[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const int n = l31 + 32 * j;
        const int nn = n < K ? n : 0;
        const int c = nn / 9, t = nn - c * 9;
        noff[j] = c * planeX + (t / 3) * a.Wi + (t % 3);
    }
    f32x16 acc[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.0f;
    float dbacc = 0.0f;

    typedef float vf4 __attribute__((ext_vector_type(4)));
    typedef uint32_t vu4 __attribute__((ext_vector_type(4)));
    constexpr int XU = BF16DZ ? WG1_XU : WG1_XU32;
    struct Regs {
        vf4 x[XU];
        float z[BF16DZ ? 1 : 16];
        vu4 zb[BF16DZ ? 2 : 1];
    };
    struct Geo {
        int b, y0, rows, npix, run4, nchunk;
        uint32_t magic;
    };
    auto geom = [&](int item) {
        Geo g;
        g.b = item / a.S;
        const int s = item - g.b * a.S;
        g.y0 = s * a.RT;
        g.rows = min(a.RT, a.Ho - g.y0);
        g.npix = g.rows * a.Wo;
        g.run4 = (st * (g.rows - 1) + 3) * W4; // chunks per channel run
        g.nchunk = a.C * g.run4;
        g.magic = g.rows == a.RT ? magic_full : magic_last;
        return g;
    };
    auto issue = [&](int item, Regs &R) {
        const Geo g = geom(item);
        const float *src = a.x + ((size_t)g.b * a.C * a.Hi + (size_t)st * g.y0) * a.Wi;
#pragma unroll
        for (int u = 0; u < XU; ++u) {
            const int q = min(u * 256 + tid, g.nchunk - 1); // idempotent tail
            const int c = (int)__umulhi((uint32_t)q, g.magic);
            R.x[u] = *reinterpret_cast<const vf4 *>(src + (size_t)c * a.Hi * a.Wi + (q - c * g.run4) * 4);
        }
        if (BF16DZ) { // dZ bf16 NHWC: one 16-byte chunk = 8 channels of one pixel; chunk id = pixel * 4 + channel group
            const vu4 *gz = reinterpret_cast<const vu4 *>(a.dz) + ((size_t)(g.b * a.Ho + g.y0) * a.Wo) * 4;
#pragma unroll
            for (int u = 0; u < 2; ++u) R.zb[u] = gz[min(u * 256 + tid, g.npix * 4 - 1)];
        } else { // dZ f32 NCHW: slot e = channel * 128 + pixel
            const float *sz = a.dz + ((size_t)g.b * 32 * a.Ho + g.y0) * a.Wo;
            const size_t csZ = (size_t)a.Ho * a.Wo;
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const int e = u * 256 + tid, c = e >> 7, p = min(e & (WG1_ZP - 1), g.npix - 1);
                R.z[u] = sz[c * csZ + p];
            }
        }
    };
    auto commit = [&](int item, const Regs &R) {
        const Geo g = geom(item);
        const XfScale xf{a.scale};
#pragma unroll
        for (int u = 0; u < XU; ++u) {
            const int q = min(u * 256 + tid, g.nchunk - 1);
            const int c = (int)__umulhi((uint32_t)q, g.magic);
            vf4 v = R.x[u];
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = xf(v[e]);
            *reinterpret_cast<vf4 *>(ldsX + c * planeX + (q - c * g.run4) * 4) = v;
        }
        if (BF16DZ) {
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int q = min(u * 256 + tid, g.npix * 4 - 1);
                const int p = q >> 2, c0 = (q & 3) * 8;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    ldsZ[(c0 + 2 * e) * planeZ + p] = bf16lo(R.zb[u][e]);
                    ldsZ[(c0 + 2 * e + 1) * planeZ + p] = bf16hi(R.zb[u][e]);
                }
            }
        } else {
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const int e = u * 256 + tid, c = e >> 7, p = e & (WG1_ZP - 1);
                if (p < g.npix) ldsZ[c * planeZ + p] = R.z[u];
            }
        }
    };

    Regs R;
    // neighbouring strips of an image share a halo row (1 of the 5 input rows of a 2-row strip): XCD-contiguous slots make them
    // run at the same time on ONE XCD, where the second read hits in L2 (PMC FETCH was 1.38x the algorithmic bytes with
    // consecutive strips dealt round-robin over the eight XCDs)
    const int first = xcd_slot((int)blockIdx.x, (int)gridDim.x);
    if (first < a.n_items) issue(first, R);
    for (int item = first; item < a.n_items; item += gridDim.x) {
        __syncthreads(); // the previous strip's readers are done
        commit(item, R);
        __syncthreads();
        if (item + (int)gridDim.x < a.n_items) issue(item + gridDim.x, R); // in flight under the MFMAs below
        const Geo g = geom(item);
        const int npair = (g.npix + 1) >> 1;
        // operands of pixel pair pr: z = dZ[co = l31][p], x[j] = X tap n = l31 + 32 j of pixel p = 2 pr + h.  Addresses come
        // from the pair table (2 adds + NT adds per pair instead of a division, a multiply and selects: every vector
        // instruction between a wave's MFMAs costs MFMA time); two pairs in flight per wave.
        const char *zc = reinterpret_cast<const char *>(ldsZ + l31 * planeZ);
        const char *xc[NT];
#pragma unroll
        for (int j = 0; j < NT; ++j) xc[j] = reinterpret_cast<const char *>(ldsX + noff[j]);
        const int2 *tp = tab + (g.rows == a.RT ? 0 : ntab) + 2 * wave + h;
        auto ld = [&](float &z, float (&xv)[NT]) {
            const int2 t = *tp;
            tp += 8;
            z = *reinterpret_cast<const float *>(zc + t.x);
#pragma unroll
            for (int j = 0; j < NT; ++j) xv[j] = *reinterpret_cast<const float *>(xc[j] + t.y);
        };
        float z0, z1, x0[NT], x1[NT];
        ld(z0, x0); // (pairs past the strip read the zero word: harmless)
        for (int pr = wave; pr < npair; pr += 8) {
            ld(z1, x1);
            __builtin_amdgcn_sched_barrier(0);
            dbacc += z0;
#pragma unroll
            for (int j = 0; j < NT; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(z0, x0[j], acc[j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            ld(z0, x0);
            __builtin_amdgcn_sched_barrier(0);
            if (pr + 4 < npair) {
                dbacc += z1;
#pragma unroll
                for (int j = 0; j < NT; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(z1, x1[j], acc[j], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    __syncthreads();
    const int slab_sz = 32 * K + 32;
    float *slab = a.partial + (size_t)blockIdx.x * slab_sz;
    for (int w = 1; w < 4; ++w) {
        if (wave == w) {
#pragma unroll
            for (int j = 0; j < NT; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) lds[(j * 16 + r) * 64 + lane] = acc[j][r];
            lds[NT * 16 * 64 + lane] = dbacc;
        }
        __syncthreads();
        if (wave == 0) {
#pragma unroll
            for (int j = 0; j < NT; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[j][r] += lds[(j * 16 + r) * 64 + lane];
            dbacc += lds[NT * 16 * 64 + lane];
        }
        __syncthreads();
    }
    if (wave == 0) {
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const int n = l31 + 32 * j;
            if (n < K)
#pragma unroll
                for (int r = 0; r < 16; ++r) slab[(size_t)acc_row(r, h) * K + n] = acc[j][r];
        }
        const float other = __shfl_xor(dbacc, 32, 64);
        if (h == 0) slab[32 * K + l31] = dbacc + other;
    }
}

// the per-workgroup slabs are summed in a FIXED order by encoder_reduce.hip (reduce_slabs)
constexpr int RG = 8; // rows of its level-1 scratch

} // namespace augcl

// =========================================================================================
// C ABI
// =========================================================================================
using namespace augcl;

namespace {

constexpr int kMaxLds = 76 * 1024; // two workgroups per CU within the 160 KiB LDS
constexpr int kFirstLds = 50 * 1024; // first-layer forward: three workgroups per CU
constexpr int kWg1Lds = 37 * 1024;   // first-layer backward-weights: four workgroups per CU
constexpr int kWg1PerCu = 4;
constexpr int kBufLds = 78 * 1024; // one of the two tile buffers of a double-buffered, one-per-CU workgroup

int pick_rt(int Ho, int words_per_row_pair /* LDS words per output row (all planes) */, int extra_rows,
            int max_lds_bytes, int *S, int extra = 0)
{
    // largest RT with (RT + extra_rows) * words <= budget, then balance the strips
    int rt = max_lds_bytes / 4 / words_per_row_pair - extra_rows;
    if (rt < 1) return 0;
    if (rt > Ho) rt = Ho;
    int s = (Ho + rt - 1) / rt;
    s = std::min(Ho, s + extra); // measured per layer width: augcl::extra_strips
    rt = (Ho + s - 1) / s;
    *S = (Ho + rt - 1) / rt;
    return rt;
}

template <typename K>
int set_lds(K kern, size_t bytes, const char *name)
{
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)bytes);
    if (e != hipSuccess) return augcl::fail(AUGCL_E_LAUNCH, "%s: cannot set dynamic LDS %zu: %s", name, bytes,
                                            hipGetErrorString(e));
    return AUGCL_OK;
}

int grid_for(int n_items, int per_cu)
{
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) {
        int v = 0;
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) cus = v;
    }
    const int g = cus * per_cu;
    return n_items < g ? n_items : g;
}

} // namespace

extern "C" {

static int conv32_forward_impl(const float *in, const float *w, const float *bias, const float *w2, const float *bias2,
                               float *out, int B, int Hi, int Wi, int relu, augcl_stream_t stream, const char *name)
{
    AUGCL_REQUIRE(B >= 0 && Hi >= 3 && Wi >= 3, "%s: bad shape", name);
    if (B == 0) return AUGCL_OK;
    AUGCL_REQUIRE(in && w && out, "%s: null tensor", name);
    AUGCL_REQUIRE(aligned16(w) && aligned16(w2), "%s: weights must be 16-byte aligned", name);
    AUGCL_REQUIRE(w2 == nullptr || (B & 1) == 0, "%s: a two-weight-set batch must be even", name);
    AUGCL_REQUIRE((size_t)B * 32 * Hi * Wi < ((size_t)1 << 30), "%s: tensor too large for 32-bit byte offsets (2^30 elements)", name);
    ConvArgs a{};
    a.w2 = w2; a.bias2 = bias2; a.dual = w2 != nullptr;
    a.in = in; a.w = w; a.bias = bias; a.mask = nullptr; a.out = out;
    a.B = B; a.Hi = Hi; a.Wi = Wi; a.Ho = Hi - 2; a.Wo = Wi - 2; a.relu = relu;
    int extra = augcl::extra_strips(0, Wi);
    {   // tiny batches (acting: B = 1): the few big strips of the throughput setting leave most CUs idle and make one
        // layer ~20 us; cut the image into as many strips as there are CUs to go around (a strip re-reads two halo rows
        // -- irrelevant at this size).  Any strip count gives the same values.
        int s0 = 0;
        const int rt0 = pick_rt(a.Ho, 32 * Wi, 2, kBufLds, &s0, extra);
        const int cus = grid_for(1 << 30, 1);
        if (rt0 > 0 && B * s0 * 2 <= cus) extra += std::min(a.Ho, cus / (2 * B)) - s0;
    }
    a.RT = pick_rt(a.Ho, 32 * Wi, 2, kBufLds, &a.S, extra);
    AUGCL_REQUIRE(a.RT > 0 && (a.RT + 2) * Wi <= kConvPlane, "%s: image row too wide for LDS (W=%d)", name, Wi);
    a.n_items = B * a.S;
    const size_t bufw = (size_t)32 * kConvPlane; // fixed plane stride (compile-time ds_read offsets)
    const size_t lds = (std::max(2 * bufw, bufw + 32 * 292) + 32) * 4; // two tile buffers, or one + the weight staging
    if (int rc = set_lds(k_conv32<0>, lds, name)) return rc;
    int grid = grid_for(a.n_items, 1);
    if (a.dual) grid &= ~1; // even / odd workgroups serve the two halves
    hipLaunchKernelGGL(k_conv32<0>, dim3(grid), dim3(512), lds, as_stream(stream), a);
    return augcl::check_launch(name);
}

int augcl_enc_conv32_forward(const float *in, const float *w, const float *bias, float *out, int B, int Hi, int Wi,
                             int relu, augcl_stream_t stream)
{
    return conv32_forward_impl(in, w, bias, nullptr, nullptr, out, B, Hi, Wi, relu, stream, "augcl_enc_conv32_forward");
}

// internal (encoder_stack.cpp): B2 images, the first half with (w, bias), the second with (w2, bias2)
int augcl_internal_conv32_forward_dual(const float *in, const float *w, const float *bias, const float *w2,
                                       const float *bias2, float *out, int B2, int Hi, int Wi, int relu,
                                       augcl_stream_t stream)
{
    return conv32_forward_impl(in, w, bias, w2, bias2, out, B2, Hi, Wi, relu, stream, "augcl_enc_stack_forward_dual");
}

int augcl_enc_conv32_backward_data(const float *dz, const float *w, const float *act_prev, float *dact, int B,
                                   int Ho, int Wo, augcl_stream_t stream)
{
    const char *name = "augcl_enc_conv32_backward_data";
    AUGCL_REQUIRE(B >= 0 && Ho >= 1 && Wo >= 1, "%s: bad shape", name);
    if (B == 0) return AUGCL_OK;
    AUGCL_REQUIRE(dz && w && dact, "%s: null tensor", name);
    AUGCL_REQUIRE((size_t)B * 32 * (Ho + 2) * (Wo + 2) < ((size_t)1 << 30), "%s: tensor too large for 32-bit byte offsets (2^30 elements)", name);
    ConvArgs a{};
    a.in = dz; a.w = w; a.bias = nullptr; a.mask = act_prev; a.out = dact;
    a.B = B; a.Hi = Ho; a.Wi = Wo; a.Ho = Ho + 2; a.Wo = Wo + 2; a.relu = 0;
    AUGCL_REQUIRE(aligned16(w), "%s: weights must be 16-byte aligned", name);
    // (strip heights: the even split of pick_rt measured better over the 10 layer sizes than minimising the number of
    // 4-wave tile rounds per image -- fewer, larger strips starve the loader/compute pipeline)
    // the dZ tile is kept zero-padded in LDS: row pitch Wo + 2, + 2 words behind the last row
    a.RT = pick_rt(a.Ho, 32 * (Wo + 2), 2, kBufLds - 32 * 8, &a.S, augcl::extra_strips(1, Wo));
    AUGCL_REQUIRE(a.RT > 0 && (a.RT + 2) * (Wo + 2) + 2 <= kConvPlane, "%s: image row too wide for LDS (W=%d)", name, Wo);
    a.n_items = B * a.S;
    a.magic_wi = (uint32_t)(0x100000000ull / (unsigned)Wo) + 1u;
    const size_t bufw = (size_t)32 * kConvPlane;
    const size_t lds = (std::max(2 * bufw, bufw + 32 * 292) + 32) * 4;
    if (int rc = set_lds(k_conv32<1>, lds, name)) return rc;
    hipLaunchKernelGGL(k_conv32<1>, dim3(grid_for(a.n_items, 1)), dim3(512), lds, as_stream(stream), a);
    return augcl::check_launch(name);
}

static int conv_first_forward_impl(const float *in, const float *w, const float *bias, float *out, int B, int C, int Hi,
                                   int Wi, int stride, float scale, int relu, bool out_bf16, augcl_stream_t stream,
                                   const char *name)
{
    AUGCL_REQUIRE(B >= 0 && C >= 1 && Hi >= 3 && Wi >= 3 && (stride == 1 || stride == 2) && scale > 0.0f,
                  "%s: bad shape", name);
    if (B == 0) return AUGCL_OK;
    AUGCL_REQUIRE(in && w && out, "%s: null tensor", name);
    {   // the training loop's shape (9 x 84 x 84, stride 2) has its own pipelined kernel
        const int rc = augcl::conv_first9_try(in, w, bias, out, B, C, Hi, Wi, stride, scale, relu, out_bf16, as_stream(stream), name);
        if (rc <= 0) return rc; // launched (or failed); +1 = shape not covered
    }
    Conv1Args a{};
    a.in = in; a.w = w; a.bias = bias; a.out = out;
    a.B = B; a.C = C; a.Hi = Hi; a.Wi = Wi; a.stride = stride; a.scale = scale; a.relu = relu;
    a.Ho = (Hi - 3) / stride + 1; a.Wo = (Wi - 3) / stride + 1;
    const int CH = (C + 1) / 2;
    // LDS rows per strip: stride*(RT-1)+3 for 2*CH planes
    // three workgroups per CU (measured best: staging of one overlaps the MFMA loops of the others)
    int rt = (kFirstLds / 4 / (2 * CH * Wi) - 3) / stride + 1;
    if (rt < 1) rt = (kMaxLds / 4 / (2 * CH * Wi) - 3) / stride + 1;
    AUGCL_REQUIRE(rt >= 1, "%s: image row too wide for LDS", name);
    if (rt > a.Ho) rt = a.Ho;
    a.S = (a.Ho + rt - 1) / rt;
    a.RT = (a.Ho + a.S - 1) / a.S;
    a.n_items = B * a.S;
    const size_t lds = (size_t)2 * CH * (stride * (a.RT - 1) + 3) * Wi * 4;
    hipStream_t s = as_stream(stream);
#define LAUNCH_FIRST(CHV)                                                                                  \
    do {                                                                                                   \
        if (out_bf16) {                                                                                    \
            if (int rc = set_lds(k_conv_first<CHV, true>, lds, name)) return rc;                           \
            hipLaunchKernelGGL((k_conv_first<CHV, true>), dim3(grid_for(a.n_items, 3)), dim3(256), lds, s, a); \
        } else {                                                                                           \
            if (int rc = set_lds(k_conv_first<CHV, false>, lds, name)) return rc;                          \
            hipLaunchKernelGGL((k_conv_first<CHV, false>), dim3(grid_for(a.n_items, 3)), dim3(256), lds, s, a); \
        }                                                                                                  \
    } while (0)
    switch (CH) {
    case 1: LAUNCH_FIRST(1); break;
    case 2: LAUNCH_FIRST(2); break;
    case 3: LAUNCH_FIRST(3); break;
    case 5: LAUNCH_FIRST(5); break;
    case 6: LAUNCH_FIRST(6); break;
    case 8: LAUNCH_FIRST(8); break;
    default:
        return augcl::fail(AUGCL_E_UNSUPPORTED, "%s: %d input channels not instantiated (have 1-6, 9-12, 15-16)", name, C);
    }
#undef LAUNCH_FIRST
    return augcl::check_launch(name);
}

extern "C" int augcl_enc_conv_first_forward(const float *in, const float *w, const float *bias, float *out, int B, int C,
                                            int Hi, int Wi, int stride, float scale, int relu, augcl_stream_t stream)
{
    return conv_first_forward_impl(in, w, bias, out, B, C, Hi, Wi, stride, scale, relu, false, stream,
                                   "augcl_enc_conv_first_forward");
}

extern "C" int augcl_enc_bf16_conv_first_forward(const float *in, const float *w, const float *bias, void *out_bf16_nhwc,
                                                 int B, int C, int Hi, int Wi, int stride, float scale, int relu,
                                                 augcl_stream_t stream)
{
    return conv_first_forward_impl(in, w, bias, reinterpret_cast<float *>(out_bf16_nhwc), B, C, Hi, Wi, stride, scale,
                                   relu, true, stream, "augcl_enc_bf16_conv_first_forward");
}

size_t augcl_enc_wgrad_workspace_bytes(int C)
{
    // slabs for up to 256 CUs x workgroups per CU (1 for the 32-channel layers, kWg1PerCu for a first layer) + RG
    // rows of the two-level reduction; C = input channels of the layer
    const size_t per_cu = C == 32 ? 1 : kWg1PerCu;
    return (size_t)(256 * per_cu + 8) * ((size_t)32 * C * 9 + 32) * sizeof(float);
}

static int wgrad32_impl(const float *act_prev, const float *dz, float *dw, float *db, float *workspace,
                        size_t workspace_bytes, int B, int Hi, int Wi, bool bf16in, augcl_stream_t stream,
                        const char *name)
{
    AUGCL_REQUIRE(B >= 1 && Hi >= 3 && Wi >= 3, "%s: bad shape", name);
    AUGCL_REQUIRE(act_prev && dz && dw && db && workspace, "%s: null tensor", name);
    WgradArgs a{};
    a.act = act_prev; a.dz = dz; a.partial = workspace;
    a.B = B; a.Hi = Hi; a.Wi = Wi; a.Ho = Hi - 2; a.Wo = Wi - 2;
    // LDS words per output row: A rows (32*Wi) + Z rows (32*Wo); +2 halo rows of A; + slack for the odd strides
    const int budget = kBufLds; // per buffer of the double-buffered {A, dZ} tile pair (one workgroup per CU)
    int rt = (budget / 4 - 64 * Wi - 64 - 32 - 600) / (32 * Wi + 32 * a.Wo); // - zero words - half the pair tables
    // register prefetch geometry: an A run (rt+2 rows) fits 8, a dZ run (rt rows) 4 dwords per lane
    rt = std::min(rt, std::min(512 / Wi - 2, 256 / a.Wo));
    AUGCL_REQUIRE(rt >= 1 && a.Wo >= 8, "%s: unsupported width %d (need 10 <= W <= 170)", name, Wi);
    if (rt > a.Ho) rt = a.Ho;
    a.S = (a.Ho + rt - 1) / rt;
    a.RT = (a.Ho + a.S - 1) / a.S;
    a.n_items = B * a.S;
    const int planeA = ((a.RT + 2) * Wi) | 1, planeZ = (a.RT * a.Wo + 1) | 1;
    const int ntab = 2 * (((a.RT * a.Wo + 1) >> 1) + 8); // pair table: two sets of ntab (dZ offset, A offset) entries
    size_t lds = 2 * (size_t)(32 * planeA + 32 * planeZ) * 4 + (size_t)2 * ntab * 8;
    const size_t red = (size_t)(145 * 64) * 4; // wave-reduction scratch reuses the tiles
    if (lds < red) lds = red;
    const int grid = grid_for(a.n_items, 1);
    AUGCL_REQUIRE(workspace_bytes >= (size_t)(grid + RG) * WG_SLAB * 4, "%s: workspace too small (%zu < %zu)", name,
                  workspace_bytes, (size_t)(grid + RG) * WG_SLAB * 4);
    hipStream_t s = as_stream(stream);
    if (bf16in) {
        if (int rc = set_lds(k_wgrad32<0, true>, lds, name)) return rc;
        hipLaunchKernelGGL((k_wgrad32<0, true>), dim3(grid), dim3(256), lds, s, a);
    } else { // f32: 4 accumulating + 4 loader waves
        if (int rc = set_lds(k_wgrad32<0, false, true>, lds, name)) return rc;
        hipLaunchKernelGGL((k_wgrad32<0, false, true>), dim3(grid), dim3(512), lds, s, a);
    }
    if (int rc = augcl::check_launch(name)) return rc;
    return augcl::reduce_slabs(ReduceDesc{workspace, workspace + (size_t)grid * WG_SLAB, dw, db, grid, WG_SLAB, 32 * 288}, s, name);
}

int augcl_enc_conv32_backward_weights(const float *act_prev, const float *dz, float *dw, float *db, float *workspace,
                                      size_t workspace_bytes, int B, int Hi, int Wi, augcl_stream_t stream)
{
    return wgrad32_impl(act_prev, dz, dw, db, workspace, workspace_bytes, B, Hi, Wi, false, stream,
                        "augcl_enc_conv32_backward_weights");
}

int augcl_enc_bf16_conv32_backward_weights(const void *act_prev_bf16_nhwc, const void *dz_bf16_nhwc, float *dw, float *db,
                                           float *workspace, size_t workspace_bytes, int B, int Hi, int Wi,
                                           augcl_stream_t stream)
{
    return wgrad32_impl(reinterpret_cast<const float *>(act_prev_bf16_nhwc), reinterpret_cast<const float *>(dz_bf16_nhwc),
                        dw, db, workspace, workspace_bytes, B, Hi, Wi, true, stream,
                        "augcl_enc_bf16_conv32_backward_weights");
}

static int g_wg1_rows3 = 1; // (A/B switch for tools: augcl_internal_wgrad_first_rows3)
constexpr int kWg1Lds3 = 50 * 1024; // float32 dZ: three workgroups per CU (the kernel's register budget), 3-row strips

static int wgrad_first_impl(const float *x, const float *dz, float *dw, float *db, float *workspace,
                            size_t workspace_bytes, int B, int C, int Hi, int Wi, int stride, float scale,
                            bool bf16dz, augcl_stream_t stream, const char *name)
{
    AUGCL_REQUIRE(B >= 1 && C >= 1 && Hi >= 3 && Wi >= 3 && (stride == 1 || stride == 2) && scale > 0.0f,
                  "%s: bad shape", name);
    AUGCL_REQUIRE(x && dz && dw && db && workspace, "%s: null tensor", name);
    Wgrad1Args a{};
    a.x = x; a.dz = dz; a.partial = workspace;
    a.B = B; a.C = C; a.Hi = Hi; a.Wi = Wi; a.stride = stride; a.scale = scale;
    a.Ho = (Hi - 3) / stride + 1; a.Wo = (Wi - 3) / stride + 1;
    // words: X planes C*(stride*(RT-1)+3)*Wi + Z planes 32*(RT*Wo|1).  Four small workgroups per CU: the kernel
    // stages and computes in turn, so co-resident workgroups are what overlaps the two (221 -> 132 us at B=512)
    const int nt = (C * 9 + 31) / 32;
    int per_cu = nt <= 3 ? kWg1PerCu : nt <= 6 ? 2 : 1; // matches the kernel's launch bounds (accumulator tiles)
    int rt = (kWg1Lds * kWg1PerCu / per_cu / 4 - 3 * C * Wi - 64) / (C * stride * Wi + 32 * a.Wo);
    if (rt < 1) {
        per_cu = 1;
        rt = (150 * 1024 / 4 - 3 * C * Wi - 64) / (C * stride * Wi + 32 * a.Wo);
    }
    AUGCL_REQUIRE(rt >= 1, "%s: image row too wide for LDS", name);
    if (rt > a.Ho) rt = a.Ho;
    // float32 dZ runs three workgroups per CU whatever the strip (153 registers): their LDS share holds a taller strip -- more
    // bytes in flight per workgroup (the launch is bound by how many it keeps in flight) and fewer halo rows per output row.
    // Taken only where the pipelined kernel covers the taller strip.
    size_t lds_cap = (size_t)kWg1Lds * kWg1PerCu / per_cu;
    if (g_wg1_rows3 && !bf16dz && nt <= 3 && per_cu == kWg1PerCu && Wi % 4 == 0 && aligned16(x)) {
        int rt3 = (kWg1Lds3 / 4 - 3 * C * Wi - 64) / (C * stride * Wi + 32 * a.Wo);
        if (rt3 > a.Ho) rt3 = a.Ho;
        if (rt3 > rt) {
            const int s3 = (a.Ho + rt3 - 1) / rt3, r3 = (a.Ho + s3 - 1) / s3;
            if (r3 * a.Wo <= WG1_ZP && C * (stride * (r3 - 1) + 3) * (Wi / 4) <= 256 * WG1_XU32) {
                rt = rt3;
                lds_cap = kWg1Lds3;
            }
        }
    }
    a.S = (a.Ho + rt - 1) / rt;
    a.RT = (a.Ho + a.S - 1) / a.S;
    a.n_items = B * a.S;
    const int K = C * 9;
    const int NT = (K + 31) / 32;
    const int planeX = (stride * (a.RT - 1) + 3) * Wi, planeZ = (a.RT * a.Wo) | 1;
    size_t lds = (size_t)(C * planeX + 32 * planeZ) * 4;
    const size_t red = (size_t)((NT * 16 + 1) * 64) * 4;
    if (lds < red) lds = red;
    const int grid = grid_for(a.n_items, per_cu);
    const int slab = 32 * K + 32;
    AUGCL_REQUIRE(workspace_bytes >= (size_t)(grid + RG) * slab * 4, "%s: workspace too small", name);
    hipStream_t s = as_stream(stream);
    // pipelined staging (k_wgrad_first_pipe) for the shapes it covers; the generic kernel otherwise
    const int run4_full = (stride * (a.RT - 1) + 3) * (Wi / 4);
    const int rows_last = a.Ho - (a.S - 1) * a.RT;
    const int run4_last = (stride * (rows_last - 1) + 3) * (Wi / 4);
    // its LDS: X planes, dZ planes with a spare zero word each, the pair tables
    const int planeZp = (a.RT * a.Wo + 1) | 1, ntab = 2 * (((a.RT * a.Wo + 1) >> 1) + 8);
    const size_t lds_pipe = std::max((size_t)(((C * planeX + 32 * planeZp + 1) & ~1) + 2 * ntab * 2) * 4, red);
    const bool pipe = Wi % 4 == 0 && aligned16(x) && a.RT * a.Wo <= WG1_ZP && C * run4_full <= 256 * (bf16dz ? WG1_XU : WG1_XU32) &&
                      (!bf16dz || a.RT * a.Wo * 4 <= 512) && NT <= 3 && lds_pipe <= lds_cap;
    if (pipe) {
        const size_t lds = lds_pipe;
        const int grid = grid_for(a.n_items, bf16dz ? 4 : 3); // = the kernel's launch bounds: every workgroup resident
        const uint32_t mf = (uint32_t)(0x100000000ull / (unsigned)run4_full) + 1u;
        const uint32_t ml = (uint32_t)(0x100000000ull / (unsigned)run4_last) + 1u;
        const uint32_t mwo = (uint32_t)(0x100000000ull / (unsigned)a.Wo) + 1u;
#define LAUNCH_WG1P(NTV)                                                                                  \
    do {                                                                                                  \
        if (bf16dz) {                                                                                     \
            if (int rc = set_lds(k_wgrad_first_pipe<NTV, true>, lds, name)) return rc;                    \
            hipLaunchKernelGGL((k_wgrad_first_pipe<NTV, true>), dim3(grid), dim3(256), lds, s, a, mf, ml, mwo);  \
        } else {                                                                                          \
            if (int rc = set_lds(k_wgrad_first_pipe<NTV, false>, lds, name)) return rc;                   \
            hipLaunchKernelGGL((k_wgrad_first_pipe<NTV, false>), dim3(grid), dim3(256), lds, s, a, mf, ml, mwo); \
        }                                                                                                 \
    } while (0)
        switch (NT) {
        case 1: LAUNCH_WG1P(1); break;
        case 2: LAUNCH_WG1P(2); break;
        default: LAUNCH_WG1P(3); break;
        }
#undef LAUNCH_WG1P
        float *part2p = workspace + (size_t)grid * slab;
        if (int rc = augcl::check_launch(name)) return rc;
        if (int rc = augcl::reduce_slabs(ReduceDesc{workspace, part2p, dw, db, grid, slab, 32 * K}, s, name)) return rc;
        return augcl::check_launch(name);
    }
#define LAUNCH_WG1(NTV)                                                                      \
    do {                                                                                     \
        if (bf16dz) {                                                                        \
            if (int rc = set_lds(k_wgrad_first<NTV, true>, lds, name)) return rc;            \
            hipLaunchKernelGGL((k_wgrad_first<NTV, true>), dim3(grid), dim3(256), lds, s, a);  \
        } else {                                                                             \
            if (int rc = set_lds(k_wgrad_first<NTV, false>, lds, name)) return rc;           \
            hipLaunchKernelGGL((k_wgrad_first<NTV, false>), dim3(grid), dim3(256), lds, s, a); \
        }                                                                                    \
    } while (0)
    switch (NT) {
    case 1: LAUNCH_WG1(1); break;
    case 2: LAUNCH_WG1(2); break;
    case 3: LAUNCH_WG1(3); break;
    case 4: LAUNCH_WG1(4); break;
    case 5: LAUNCH_WG1(5); break;
    default:
        return augcl::fail(AUGCL_E_UNSUPPORTED, "%s: %d input channels not supported (max 17)", name, C);
    }
#undef LAUNCH_WG1
    float *part2 = workspace + (size_t)grid * slab;
    if (int rc = augcl::check_launch(name)) return rc;
    if (int rc = augcl::reduce_slabs(ReduceDesc{workspace, part2, dw, db, grid, slab, 32 * K}, s, name)) return rc;
    return augcl::check_launch(name);
}

int augcl_internal_wgrad_first_rows3(int on) // (tools: A/B of the taller strips; returns the previous setting)
{
    const int was = g_wg1_rows3;
    g_wg1_rows3 = on ? 1 : 0;
    return was;
}

int augcl_enc_conv_first_backward_weights(const float *x, const float *dz, float *dw, float *db, float *workspace,
                                          size_t workspace_bytes, int B, int C, int Hi, int Wi, int stride, float scale,
                                          augcl_stream_t stream)
{
    return wgrad_first_impl(x, dz, dw, db, workspace, workspace_bytes, B, C, Hi, Wi, stride, scale, false, stream,
                            "augcl_enc_conv_first_backward_weights");
}

int augcl_enc_bf16_conv_first_backward_weights(const float *x, const void *dz_bf16_nhwc, float *dw, float *db,
                                               float *workspace, size_t workspace_bytes, int B, int C, int Hi, int Wi,
                                               int stride, float scale, augcl_stream_t stream)
{
    return wgrad_first_impl(x, reinterpret_cast<const float *>(dz_bf16_nhwc), dw, db, workspace, workspace_bytes, B, C, Hi,
                            Wi, stride, scale, true, stream, "augcl_enc_bf16_conv_first_backward_weights");
}

} // extern "C"
