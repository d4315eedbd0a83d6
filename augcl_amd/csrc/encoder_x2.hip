// encoder_x2.hip -- the 32 -> 32 layers of the shared conv encoder (src/algorithms/modules.py:145-165) with
// FLOAT32-ACCURATE products on the f16 matrix pipe, THREE matrix instructions per product group instead of the split-bf16
// kernels' six (encoder_x3.hip) or the native f32 pipe's eight at half the rate: the default parity path since round 3.
//
// An f16 carries 11 significand bits.  With a per-tensor power of two s (exact) that brings max|x| into [2^14, 2^15),
//     x s = hi + lo' 2^-11,   hi = f16(x s),   lo' = f16((x s - hi) 2^11)        (both round-to-nearest-even)
// holds to 2^-23 relative for every element down to 2^-26 of the tensor's maximum and to 2^-21 down to 2^-29 (lo', then hi
// go subnormal); below that the absolute error stays under 2^-36 of the maximum (tests/test_f16_pair_numerics.py).  f16 x f16 products are exact in
// f32, the MFMA accumulates in f32, and of the four cross products hh + (h l' + l' h) 2^-11 + l' l' 2^-22 the last one is
// below float32's resolution.  Measured on this chip (tools/x3probe.hip, K = 288, against float64): 1.36e-7 relative L2 for
// the three products (all four: 1.29e-7; split-bf16 six products: 2.6e-7; native f32 MFMA chain: 3.0e-7).  The main
// product accumulates in one register set, the two cross products in a second one that is scaled by 2^-11 at the end.
//
// What f16 lacks is float32's exponent range, hence the scales: every producer of an activation / gradient tensor on this
// path leaves max|x| next to it (one v_max per output value, one atomic per workgroup), every consumer derives s from it.  The
// weights' maximum is taken by the fragment-preparation launch.  Tensors whose elements span more than 2^26 lose RELATIVE
// accuracy in their smallest elements (not in any norm); the split-bf16 kernels (precision "f32_x3") have no such limit.
//
// The f16 MFMA's accumulation is biased toward -infinity like the bf16 one's, 16x less (-9e-11 sum|w x| per output): each
// strip / workgroup alternates the SIGN of its pixel operand (the loaders scale by +s or -s, the epilogue undoes it), so
// the bias cancels in every sum over pixels (db, dW) instead of adding up coherently.
//
// Structure as in encoder_x3.hip: 4 loader + 4 compute waves, double-buffered pixel-major LDS tile (slot = hi | lo' x 32
// channels + pad = 144 B), loaders two strips ahead, weights' fragments (hi and lo', 144 VGPRs) resident in the compute
// waves' registers, operand reads two groups ahead of their MFMAs, pipelined epilogue (stores of tile i inside tile i + 1).
#include "encoder_x2_common.h"

#include <algorithm>
#include <type_traits>

namespace augcl {

struct ConvX2Args {
    const float *in;     // forward: a_{l-1} (post-ReLU) (B,32,Hi,Wi);  bwd-data: dZ_l
    const float *bias;   // (32) or null                       [MODE 0]
    const float *mask;   // a_{l-1} (B,32,Ho,Wo) or null       [MODE 1]
    float *out;          // (B,32,Ho,Wo)
    int B, Hi, Wi, Ho, Wo, RT, S, relu, n_items;
    uint32_t magic_wo;   // floor(2^32 / Wo) + 1
    uint32_t magic_s;    // floor(2^32 / S) + 1 (S == 1 is tested for)
    uint32_t magic_qw;   // MODE 1: floor(2^32 / QW) + 1, QW = quads per dZ row
    uint32_t magic_wi;   // MODE 0: floor(2^32 / Wi) + 1
    const float *bias2;  // two-weight-set forward: images [B/2, B) use wfrag2 / bias2 (odd workgroups)
    int dual;
    const vu4 *wfrag, *wfrag2; // prepared by k_x2_prep: [hi | lo'][18 operands][64 lanes] x 16 B, then the exponent
    const uint32_t *amax_in;   // max |in| (float bits); dual: [0] first half, [1] second half
    uint32_t *amax_out;        // max |out| is atomically max-ed into it (zeroed by the caller), or null; dual: [0], [1]
    uint32_t *bits;            // ReLU gate bits, one word per pixel (B,Ho,Wo), bit c = (channel c of that pixel > 0): MODE 0 writes
                               // them for its output (or null), MODE 1 with GATE = 1 reads them instead of the float mask
    int abl;                   // tuning builds only: ablation bits (tools/x2_bench.py)
};

#ifdef AUGCL_TUNING
// tuning builds: compute wave 0 of workgroup 0 records (shader cycles, 100 MHz ticks, MFMA tiles it ran) of the last launch
__device__ unsigned long long g_x2_clk[4];
#endif

// MODE 0: forward (valid 3x3).  MODE 1: backward-data = full correlation of dZ with the flipped, transposed weights; its
// LDS tile keeps every dZ row with 2 zero pixels in front (which double as the 2 behind the previous row) and zero rows
// outside the image, so no tap needs masking.
// GATE: MODE 0: 1 = also write the output's gate bits (a.bits);  MODE 1: 0 = the ReLU gate is the sign of the float mask (or
// there is none), 16 dword loads per tile and lane; 1 = the gate bits the forward pass left (ONE dword per tile and lane,
// 1/32 of the mask's bytes).
template <int MODE, int GATE>
__global__ __launch_bounds__(512, 2) void k_conv32_x2(ConvX2Args a)
{
    extern __shared__ __attribute__((aligned(16))) char ldsx[];
    constexpr int P = MODE ? 2 : 0;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool loader = wave >= 4;
    const int cw = wave & 3;
    const int l31 = lane & 31, h = lane >> 5;
    const int Wi = a.Wi;
    const int Wl = MODE ? Wi + 2 : Wi;                     // LDS row pitch in pixels
    const int tilepix = (a.RT + 2) * Wl + (MODE ? 2 : 0);
    // LDS bytes between tile rows.  A 32-pixel tile crosses one or two output-row ends; backward-data's rows are as long as its
    // output's (W + 2), so consecutive pixels are consecutive slots and a 16-lane read group never meets a bank twice.  The
    // forward's tile row is TWO slots longer than an output row: 14 chunks of padding per row make the chunk index
    // 9 (y Wl + x) + 14 y = 9 m + 32 y linear in the pixel number m again (mod 16) -- without it a third of the forward
    // kernels' LDS cycles were bank-conflict cycles (profiles/r03_x2_pmc.txt), against 6 % in backward-data.
    const int rowb = Wl * X2_SLOT + (MODE ? 0 : X2_ROWPAD);
    const int bufbytes = tilepix * X2_SLOT + (MODE ? 0 : (a.RT + 2) * X2_ROWPAD);
    float *lbias = reinterpret_cast<float *>(ldsx);        // [0..32) +bias s_in s_w, [32..64) the same negated
    char *tiles = ldsx + X2_HDR;
    // two weight sets: the halves' items are dealt so that NEIGHBOURING strips of an image run on one XCD here too (workgroup i
    // sits on XCD i % 8; with even / odd workgroups taking the halves and slot = i / 2, consecutive strips landed on different
    // XCDs and their two shared halo rows were fetched twice: PMC FETCH 1.157x the algorithmic bytes where the one-set launch,
    // which uses xcd_slot, reads 1.000x).  Grids that are not a multiple of 16 keep the even / odd map.
    const bool dual16 = a.dual && (gridDim.x & 15) == 0;
    const int grp = a.dual ? (dual16 ? (int)((blockIdx.x >> 3) & 1) : (int)(blockIdx.x & 1)) : 0, nhalf = a.n_items >> 1;
    const int dslot = dual16 ? (int)(blockIdx.x & 7) * (int)(gridDim.x >> 4) + (int)(blockIdx.x >> 4) : (int)(blockIdx.x >> 1);
    const int first = a.dual ? grp * nhalf + dslot : xcd_slot((int)blockIdx.x, (int)gridDim.x);
    const int NG = a.dual ? (int)(gridDim.x >> 1) : (int)gridDim.x;
    const int n = a.dual ? (grp + 1) * nhalf : a.n_items;
    const vu4 *const wfrag = grp ? a.wfrag2 : a.wfrag;
#ifdef AUGCL_TUNING
    const unsigned long long t_start = __builtin_amdgcn_s_memtime(), r_start = __builtin_amdgcn_s_memrealtime();
    unsigned long long n_tiles = 0;
#endif
    // scales (wave-uniform): input from its producer's maximum, weights from the preparation launch
    const int k_in = scale_exp(__builtin_amdgcn_readfirstlane((int)a.amax_in[grp]));
    const int k_w = __builtin_amdgcn_readfirstlane((int)wfrag[X2_FRAG_VU4][0]);
    const float s_in = pow2f(k_in);

    uint32_t *lred = reinterpret_cast<uint32_t *>(ldsx) + 64; // [0] the workgroup's max |out| (float bits), [1] waves that have added theirs
    if (!loader) {
        const float *bsrc = grp ? a.bias2 : a.bias;
        if (tid < 32) {
            const float bv = (MODE == 0 && bsrc) ? bsrc[tid] * s_in * pow2f(k_w) : 0.0f;
            lbias[tid] = bv;
            lbias[32 + tid] = -bv;
        }
        if (tid < 2) lred[tid] = 0u;
    }

    // The two roles run DIFFERENT code with the same barrier sequence (1 + one per item): the branch is wave-uniform.
    if (loader) {
        // unit = (4 consecutive pixels of the strip's input rows) x (8 channels): 8 dwordx4 BUFFER loads (address = resource base +
        // this lane's 32-bit byte offset + a scalar offset carrying image, first row and channel; item geometry in SGPRs), then
        // per pixel one scale + split of its 8 channels and two 16-byte LDS stores.  The loader runs TWO strips ahead.
        const int lt = cw * 64 + lane; // 0..255 among the loader threads
        struct Regs {
            vu4 d[X2_NU][8];
        };
        struct Geo {
            int rows, r0, nrow, top;
            unsigned base; // byte offset of (image b, channel 0, row r0)
        };
        const int QW = (Wi + 3) >> 2; // MODE 1: quads per dZ row (the last one overlaps its neighbour)
        const unsigned cs4 = (unsigned)(a.Hi * Wi) * 4u; // bytes per channel plane
        const int cg = lt & 3, q0 = lt >> 2;             // this lane's channel group (8 channels) and first quad
        const unsigned lane_c = (unsigned)cg * 8u * cs4;
        const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a.in), 0, -1, 0x00020000);
        auto geom = [&](int item) {
            Geo g;
            const int b = a.S == 1 ? item : (int)__umulhi((uint32_t)item, a.magic_s);
            const int y0 = (item - b * a.S) * a.RT;
            g.rows = min(a.RT, a.Ho - y0);
            const int rbase = y0 - P;                             // image row of tile row 0
            g.r0 = MODE ? max(0, rbase) : y0;                     // image rows [r0, r0 + nrow) exist
            g.nrow = (MODE ? min(a.Hi, y0 + g.rows) : y0 + g.rows + 2) - g.r0;
            g.top = g.r0 - rbase;
            g.base = (unsigned)((b * 32 * a.Hi + g.r0) * Wi) * 4u;
            return g;
        };
        auto locate = [&](const Geo &g, int u, unsigned &voff, int &lb) {
            if (MODE == 0) { // the strip's input rows are one contiguous run: quad qd = pixels 4 qd .. 4 qd + 3 of it
                const int p0 = min(4 * (q0 + 64 * u), g.nrow * Wi - 4); // units past the run repeat its last quad (idempotent)
                voff = lane_c + 4u * (unsigned)p0;
                lb = p0 * X2_SLOT + cg * 16;
            } else {
                const int qd = min(q0 + 64 * u, g.nrow * QW - 1);
                const int row = (int)__umulhi((uint32_t)qd, a.magic_qw), x0 = min(4 * (qd - row * QW), Wi - 4);
                voff = lane_c + 4u * (unsigned)(row * Wi + x0);
                lb = ((row + g.top) * Wl + 2 + x0) * X2_SLOT + cg * 16;
            }
        };
        auto issue = [&](int item, Regs &R) {
            const Geo g = geom(item);
#ifdef AUGCL_TUNING
            if (a.abl & 16) return; // ablation: no global loads
#endif
#pragma unroll
            for (int u = 0; u < X2_NU; ++u) {
                unsigned voff;
                int lb;
                locate(g, u, voff, lb);
#pragma unroll
                for (int c = 0; c < 8; ++c)
                    R.d[u][c] = __builtin_bit_cast(vu4, __builtin_amdgcn_raw_buffer_load_b128(rs_in, voff, g.base + (unsigned)c * cs4, 0));
            }
        };
        auto commit = [&](int item, int bi, const Regs &R, bool first_use) {
            const Geo g = geom(item);
            char *dst = tiles + bi * bufbytes;
            const float s = bi ? -s_in : s_in; // odd strips of a workgroup carry the opposite sign (see the header)
#ifdef AUGCL_TUNING
            if (a.abl & 2) { // ablation: no split, no LDS stores (the loads stay live)
                uint32_t x = 0;
#pragma unroll
                for (int u = 0; u < X2_NU; ++u)
#pragma unroll
                    for (int c = 0; c < 8; ++c) x ^= R.d[u][c][0] ^ R.d[u][c][1] ^ R.d[u][c][2] ^ R.d[u][c][3];
                if (x == 0x12345678u) *reinterpret_cast<uint32_t *>(dst) = x;
                return;
            }
#endif
#pragma unroll
            for (int u = 0; u < X2_NU; ++u) {
                unsigned voff;
                int lb;
                locate(g, u, voff, lb);
                char *o = dst + lb;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    vu4 hi, lo;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        uint32_t th, tl;
                        split2h(__uint_as_float(R.d[u][2 * e][j]) * s, __uint_as_float(R.d[u][2 * e + 1][j]) * s, th, tl);
                        hi[e] = th; lo[e] = tl;
                    }
                    // forward: the quad may straddle a row end of the contiguous run: every pixel takes its own row's padding
                    const int pad = MODE ? 0 : (int)__umulhi((uint32_t)(min(4 * (q0 + 64 * u), g.nrow * Wi - 4) + j), a.magic_wi) * X2_ROWPAD;
                    *reinterpret_cast<vu4 *>(o + j * X2_SLOT + pad) = hi;
                    *reinterpret_cast<vu4 *>(o + j * X2_SLOT + pad + 64) = lo;
                }
            }
            if (MODE == 1) {
                const vu4 zero = {0u, 0u, 0u, 0u};
                // the 2 pad pixels in front of every tile row (and behind the last one): data never lands on them, so once
                // per buffer is enough (8 chunks of 16 B per pixel slot)
                if (first_use)
                    for (int id = lt; id < (a.RT + 3) * 16; id += 256) {
                        const int pr = id >> 4, rem = id & 15, px = rem >> 3, ch = rem & 7;
                        *reinterpret_cast<vu4 *>(dst + (pr * Wl + px) * X2_SLOT + ch * 16) = zero;
                    }
                // tile rows above / below the image (first / last strip)
                const int top = g.top, bot = g.rows + 2 - g.top - g.nrow, rowch = Wi * 8;
                for (int id = lt; id < (top + bot) * rowch; id += 256) {
                    const int jr = id / rowch, rem = id - jr * rowch, x = rem >> 3, ch = rem & 7;
                    const int j = jr < top ? jr : g.nrow + jr; // below the image: tile rows top + nrow + (jr - top)
                    *reinterpret_cast<vu4 *>(dst + (j * Wl + x + 2) * X2_SLOT + ch * 16) = zero;
                }
            }
        };
        Regs ra, rb;
        if (first < n) {
            issue(first, ra);
            commit(first, 0, ra, true);
        }
        if (first + NG < n) issue(first + NG, ra);
        if (first + 2 * NG < n) issue(first + 2 * NG, rb);
        __syncthreads();
        bool fresh1 = true;
        for (int item = first; item < n; item += 2 * NG) {
            if (item + NG < n) commit(item + NG, 1, ra, fresh1); // even half: the compute waves read buffer 0
            fresh1 = false;
            if (item + 3 * NG < n) issue(item + 3 * NG, ra);
            __syncthreads();
            if (item + NG >= n) break;
            if (item + 2 * NG < n) commit(item + 2 * NG, 0, rb, false); // odd half: they read buffer 1
            if (item + 4 * NG < n) issue(item + 4 * NG, rb);
            __syncthreads();
        }
        return;
    }

    // ---- compute waves.  Operand i = (tap t = i / 2, channel half s = i & 1); A fragment: W_eff[row l31][k = 16 s + 8 h + j][t],
    // hi and lo' parts in registers for the whole launch (36 coalesced 16-byte loads per lane).
    f16x8 wh[18], wl[18];
#pragma unroll
    for (int i = 0; i < 18; ++i) {
        wh[i] = __builtin_bit_cast(f16x8, wfrag[i * 64 + lane]);
        wl[i] = __builtin_bit_cast(f16x8, wfrag[(18 + i) * 64 + lane]);
    }
    const int relu_floor = __builtin_amdgcn_readfirstlane(a.relu ? 0 : (int)0x80000000); // opaque: stays ONE v_max_i32
    // raw buffer resources: the output's is bounded by the tensor's size, so a lane with nothing to store just names an offset
    // beyond it (kNoStore): the hardware drops the store and the epilogue needs no branch
    const __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc(a.out, 0, (int)((unsigned)(a.B * 32 * a.Ho * a.Wo) * 4u), 0x00020000);
    constexpr unsigned kNoStore = 0xfffffff0u;
    // backward-data with a float mask (GATE 0): the layer input at the addresses this lane stores to, passed when its bits are
    // >= gate_floor = 1 (a positive float).  Without a mask the floor is INT_MIN -- everything passes -- and the loads read the
    // output tensor itself (valid memory, value irrelevant): one code path, no branch on the pointer in the hot loop.
    const __amdgpu_buffer_rsrc_t rs_mask = __builtin_amdgcn_make_buffer_rsrc(
        MODE == 1 && a.mask ? const_cast<float *>(a.mask) : a.out, 0, (int)((unsigned)(a.B * 32 * a.Ho * a.Wo) * 4u), 0x00020000);
    const int gate_floor = __builtin_amdgcn_readfirstlane(a.mask ? 1 : (int)0x80000000);
    // the gate bits: one word per pixel (B,Ho,Wo)
    const __amdgpu_buffer_rsrc_t rs_bits = __builtin_amdgcn_make_buffer_rsrc(
        GATE == 1 ? a.bits : reinterpret_cast<uint32_t *>(a.out), 0, (int)((unsigned)(a.B * a.Ho * a.Wo) * 4u), 0x00020000);
    const int chw = a.Ho * a.Wo; // 32-bit offsets: the host checks B*32*Ho*Wo < 2^30
    const unsigned hoff = (unsigned)(16 * h) * (unsigned)chw; // boff - hoff - (the image's 31 other planes) = the pixel's byte in the bit map
    const float un_in = pow2f(-k_in), un_w = pow2f(-k_w); // two exact factors (their product may leave float32's range)

    // A tile's 16 output rows are stored WHILE the next tile's MFMAs run (one store after the first MFMA of each of its first
    // 16 operand groups), also across the strip's barrier.  (Finishing them there as well -- un-scale, ReLU, maximum from a second
    // accumulator set -- measured 8 % SLOWER in the stack: the vector instructions compete with the loader wave of the same SIMD
    // for the issue port in exactly the cycles the MFMAs need it; tools/rejected/x2_two_accumulator_sets.patch.)
    float res[16];             // the previous tile's results (bias, ReLU / gate applied)
    unsigned pboff = kNoStore; // ... and this lane's byte offset for them
    float vmax = 0.0f;         // max |result| of this lane (a lane past the strip's end repeats a real pixel: harmless)
    auto store_row = [&](int r) {
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, res[r]), rs_out, pboff,
                                              (unsigned)(((r & 3) + 8 * (r >> 2)) * chw) * 4u, 0);
    };
    auto compute = [&](int item, int bi) {
        const int b = a.S == 1 ? item : (int)__umulhi((uint32_t)item, a.magic_s), s = item - b * a.S;
        const int y0 = s * a.RT;
        const int rows = min(a.RT, a.Ho - y0);
        const int npix = rows * a.Wo;
        const char *tile = tiles + bi * bufbytes;
        const int ntiles = (npix + 31) >> 5;
        if (cw >= ntiles) return; // wave-uniform
#ifdef AUGCL_TUNING
        if (a.abl & 1) return; // ablation: the loaders alone
#endif
        const float un0 = bi ? -un_in : un_in; // this strip's pixels were scaled by -s
        const unsigned img31 = (unsigned)(b * 31) * (unsigned)chw * 4u; // the 31 other planes of the images in front of this one
        auto locate = [&](int tl, bool &valid_, int &off_, unsigned &boff_) {
            int m_ = tl * 32 + l31;
            valid_ = m_ < npix;
            if (!valid_) m_ = npix - 1; // read a real pixel, discard the result
            const int yy_ = (int)__umulhi((uint32_t)m_, a.magic_wo), xx_ = m_ - yy_ * a.Wo;
            off_ = yy_ * rowb + xx_ * X2_SLOT + h * 16;
            boff_ = (unsigned)((b * 32 * a.Ho + (y0 + yy_)) * a.Wo + xx_ + 4 * h * chw) * 4u;
        };
        struct Ops {
            vu4 xh, xl;
        };
        auto rd1 = [&](int off_, int i, Ops &o, int part) { // one of the operand's two reads: row base + immediate
            const int t = i >> 1;
#ifdef AUGCL_TUNING
            if ((a.abl & 32) && (t % 3) != 0) return; // ablation: only the kx = 0 taps' operand reads (a third of the LDS read bytes)
#endif
            const char *p = tile + off_ + (t / 3) * rowb + ((t % 3) * X2_SLOT + (i & 1) * 32);
            if (part == 0) o.xh = *reinterpret_cast<const vu4 *>(p);
            else o.xl = *reinterpret_cast<const vu4 *>(p + 64);
        };
        bool valid;
        int off;
        unsigned boff;
        locate(cw, valid, off, boff);
        // backward-data, float mask: this lane's 16 ReLU gates.  Row r of the previous tile is stored and gate r of the CURRENT
        // tile requested in the same slot of the MFMA stream (the result register the store frees is the one the gate lands
        // in), two rows per operand group over the first eight groups.  With gate bits: ONE word per tile, requested up front.
        float gv[16];
        uint32_t gword = 0u;
        auto side_row = [&](int r) {
            store_row(r);
            if (GATE == 0)
                gv[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                                                      rs_mask, boff, (unsigned)(((r & 3) + 8 * (r >> 2)) * chw) * 4u, 0));
        };
        Ops o[3]; // operands i, i + 1, i + 2: reads run two groups (6 MFMAs, ~190 cycles) ahead of their use
        rd1(off, 0, o[0], 0); rd1(off, 0, o[0], 1);
        rd1(off, 1, o[1], 0); rd1(off, 1, o[1], 1);
        for (int tl = cw; tl < ntiles; tl += 4) {
            bool valid2;
            int off2;
            unsigned boff2;
            locate(tl + 4 < ntiles ? tl + 4 : tl, valid2, off2, boff2); // no next tile: re-read this one
#ifdef AUGCL_TUNING
            ++n_tiles;
#endif
            if (MODE == 1 && GATE == 1) gword = __builtin_amdgcn_raw_buffer_load_b32(rs_bits, boff - hoff - img31, 0, 0);
            f32x16 acc, accx; // main product (seeded with the bias) / the two cross products (x 2^-11 at the end)
            {
                const float *bsrc = lbias + (bi ? 32 : 0);
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) { // accumulator seed = bias of rows 8 g4 + 4 h + (0..3)
                    const vf4 b4 = *reinterpret_cast<const vf4 *>(bsrc + 8 * g4 + 4 * h);
                    acc[4 * g4] = b4[0]; acc[4 * g4 + 1] = b4[1]; acc[4 * g4 + 2] = b4[2]; acc[4 * g4 + 3] = b4[3];
                }
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) accx[r] = 0.0f;
#define X2_MM(C_, A_, B_) C_ = __builtin_amdgcn_mfma_f32_32x32x16_f16(A_, __builtin_bit_cast(f16x8, B_), C_, 0, 0, 0)
#pragma unroll
            for (int i = 0; i < 18; ++i) {
                Ops &cur = o[i % 3];
                Ops &nn = o[(i + 2) % 3];
                const int noff = i < 16 ? off : off2, ni = i < 16 ? i + 2 : i - 16; // the operand two groups on (across the tile boundary)
                // smallest products first: wl' xh, wh xl' (cross), wh xh (main); one LDS read / store / gate load between MFMAs
                X2_MM(accx, wl[i], cur.xh);
                __builtin_amdgcn_sched_barrier(0);
                rd1(noff, ni, nn, 0);
                if (MODE == 0 && i < 16) store_row(i);
                if (MODE == 1 && i < 8) side_row(2 * i);
                __builtin_amdgcn_sched_barrier(0);
                X2_MM(accx, wh[i], cur.xl);
                __builtin_amdgcn_sched_barrier(0);
                rd1(noff, ni, nn, 1);
                if (MODE == 1 && i < 8) side_row(2 * i + 1);
                __builtin_amdgcn_sched_barrier(0);
                X2_MM(acc, wh[i], cur.xh);
                __builtin_amdgcn_sched_barrier(0);
            }
#undef X2_MM
            // rows in accumulator order: co = 0,1,2,3, 8,9,10,11, ... (+4 h in the lane offset)
            uint32_t gbits = 0u;
            const uint32_t gmine = gword >> (4 * h); // MODE 1 / GATE 1: bit co of it is this lane's channel co + 4 h
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = (r & 3) + 8 * (r >> 2);
                float v = __builtin_fmaf(accx[r], 1.0f / 2048.0f, acc[r]) * un0 * un_w;
                if (MODE == 0) {
                    v = __int_as_float(max(__float_as_int(v), relu_floor));
                    // gate bit of the NEXT layer's backward-data: v >= +0 here (ReLU layers), positive <=> bit pattern >= 1
                    if (GATE == 1) gbits = (min(__float_as_uint(v), 1u) << co) | gbits;
                } else if (GATE == 1) {
                    v = __uint_as_float(__float_as_uint(v) & (uint32_t)__builtin_amdgcn_sbfe((int)gmine, co, 1));
                } else {
                    v = __float_as_int(gv[r]) >= gate_floor ? v : 0.0f;
                }
                vmax = fmaxf(vmax, fabsf(v));
                res[r] = v;
            }
            pboff = valid ? boff : kNoStore;
            if (MODE == 0 && GATE == 1) {
                // the two half-waves hold channels {0-3, 8-11, ...} and {4-7, 12-15, ...} of the same 32 pixels: v_permlane32_swap
                // hands lanes 0-31 the upper half's bits (no LDS round trip), and they store the pixel's word
                const uint32_t mine = gbits << (4 * h);
                const auto sw = __builtin_amdgcn_permlane32_swap(mine, mine, false, false);
                __builtin_amdgcn_raw_buffer_store_b32(mine | (uint32_t)sw[1], rs_bits, (h == 0 && valid) ? boff - img31 : kNoStore, 0, 0);
            }
            // (18 = 6 x 3: the next tile's operands 0 and 1 sit in o[0], o[1] again)
            valid = valid2; off = off2; boff = boff2;
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
#pragma unroll
    for (int r = 0; r < 16; ++r) store_row(r); // the last tile's results
#ifdef AUGCL_TUNING
    if (blockIdx.x == 0 && wave == 0 && lane == 0) {
        g_x2_clk[0] = __builtin_amdgcn_s_memtime() - t_start;
        g_x2_clk[1] = __builtin_amdgcn_s_memrealtime() - r_start;
        g_x2_clk[2] = n_tiles;
    }
#endif
    if (a.amax_out != nullptr) {
        // ONE global atomic per workgroup (a launch's 1024 compute waves on one address cost ~12 us of serialised atomics): every
        // compute wave adds its maximum in LDS and takes a ticket; the last one (LDS atomics execute in arrival order) sends it on
        vmax = wave_max(vmax);
        if (lane == 0) {
            atomicMax(lred, __float_as_uint(vmax));
            if (atomicAdd(lred + 1, 1u) == 3u) atomicMax(a.amax_out + grp, lred[0]);
        }
    }
}

// =========================================================================================
// backward-weights: D_t[co][ci] = sum_p dZ[p][co] * A[p + tap_t][ci], K = 16 pixels per MFMA; ds_read_b64_tr_b16 delivers the
// pixel-major tiles transposed.  Tiles per buffer: {A hi, A lo, dZ hi, dZ lo}; residuals UNSCALED (split2u), so the three
// products zh xh, zh xl, zl xh share one accumulator per tap (nine taps = 144 accumulator registers: no room for a second set).
// =========================================================================================
struct WgX2Args {
    const float *act; // a_{l-1} f32 NCHW (B,32,Hi,Wi)
    const float *dz;  // f32 NCHW (B,32,Ho,Wo)
    float *partial;   // [gridDim.x][32*288 + 32]
    int B, Hi, Wi, Ho, Wo, RT, S, n_items;
    uint32_t magic_wo, magic_s;
    const uint32_t *amax_act, *amax_dz; // max |act|, max |dz| (float bits)
    int abl;                            // tuning builds only: ablation bits (tools/x2_bench.py)
};

constexpr int WGH_SLAB = 32 * 288 + 32;
constexpr int WGH_NU = 2; // loader rounds per item: round 0 = the activation rows, round 1 = the dZ rows

__device__ __forceinline__ s16x4 lds_tr_h(const char *p)
{
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3))) *)(p));
}

__global__ __launch_bounds__(512, 2) void k_wgrad32_x2(WgX2Args a)
{
    extern __shared__ __attribute__((aligned(16))) char ldsw[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool loader = wave >= 4;
    const int cw = wave & 3;
    const int l31 = lane & 31, h = lane >> 5;
    const int Wi = a.Wi, Wo = a.Wo;
    const int apix = (a.RT + 2) * Wi;                // A tile pixels
    const int zpix = (a.RT * Wo + 15) & ~15;         // dZ tile pixels, zero-padded to whole 16-pixel steps
    const int abytes = apix * 64, zbytes = zpix * 64; // one plane
    const int bufbytes = 2 * (abytes + zbytes);
    const int q = (lane & 15) >> 2, pp = lane & 3, cblk = (lane >> 4) & 1;
    const int colbyte = (16 * cblk + 4 * pp) * 2;
    const int G = gridDim.x, n = a.n_items, first = xcd_slot((int)blockIdx.x, G);
    // odd workgroups accumulate -dW (their loaders scale dZ by -s, exact) and negate the slab when they write it: the MFMA's
    // accumulation bias cancels between even and odd workgroups in the slab sum
    const bool wg_neg = (blockIdx.x & 1) != 0;
    const int k_a = scale_exp(__builtin_amdgcn_readfirstlane((int)a.amax_act[0]));
    const int k_z = scale_exp(__builtin_amdgcn_readfirstlane((int)a.amax_dz[0]));

    if (loader) {
        const int lt = cw * 64 + lane;
        struct Regs {
            vu4 d[WGH_NU][8];
        };
        struct Geo {
            int na, nz;
            unsigned base_a, base_z;
        };
        const int cg = lt & 3, q0 = lt >> 2;
        const unsigned csA4 = (unsigned)(a.Hi * Wi) * 4u, csZ4 = (unsigned)(a.Ho * Wo) * 4u;
        const unsigned lane_a = (unsigned)cg * 8u * csA4, lane_z = (unsigned)cg * 8u * csZ4;
        const __amdgpu_buffer_rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a.act), 0, -1, 0x00020000);
        const __amdgpu_buffer_rsrc_t rs_z = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a.dz), 0, -1, 0x00020000);
        const float s_a = pow2f(k_a), s_z = wg_neg ? -pow2f(k_z) : pow2f(k_z);
        auto geom = [&](int item) {
            Geo g;
            const int b = a.S == 1 ? item : (int)__umulhi((uint32_t)item, a.magic_s);
            const int y0 = (item - b * a.S) * a.RT;
            const int rows = min(a.RT, a.Ho - y0);
            g.na = (rows + 2) * Wi;
            g.nz = rows * Wo;
            g.base_a = (unsigned)((b * 32 * a.Hi + y0) * Wi) * 4u;
            g.base_z = (unsigned)((b * 32 * a.Ho + y0) * Wo) * 4u;
            return g;
        };
        auto issue = [&](int item, Regs &R) {
            const Geo g = geom(item);
#ifdef AUGCL_TUNING
            if (a.abl & 16) return; // ablation: no global loads
#endif
            // lanes past a run repeat its last quad (idempotent)
            const int pa0 = min(4 * q0, g.na - 4), pz0 = min(4 * q0, g.nz - 4);
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                R.d[0][c] = __builtin_bit_cast(vu4, __builtin_amdgcn_raw_buffer_load_b128(rs_a, lane_a + 4u * (unsigned)pa0, g.base_a + (unsigned)c * csA4, 0));
                R.d[1][c] = __builtin_bit_cast(vu4, __builtin_amdgcn_raw_buffer_load_b128(rs_z, lane_z + 4u * (unsigned)pz0, g.base_z + (unsigned)c * csZ4, 0));
            }
        };
        auto commit = [&](int item, int bi, const Regs &R) {
            const Geo g = geom(item);
            char *dst = ldsw + bi * bufbytes;
#ifdef AUGCL_TUNING
            if (a.abl & 2) { // ablation: no split, no LDS stores (the loads stay live)
                uint32_t x = 0;
#pragma unroll
                for (int u = 0; u < WGH_NU; ++u)
#pragma unroll
                    for (int c = 0; c < 8; ++c) x ^= R.d[u][c][0] ^ R.d[u][c][1] ^ R.d[u][c][2] ^ R.d[u][c][3];
                if (x == 0x12345678u) *reinterpret_cast<uint32_t *>(dst) = x;
                return;
            }
#endif
            const int pa0 = min(4 * q0, g.na - 4), pz0 = min(4 * q0, g.nz - 4);
#pragma unroll
            for (int u = 0; u < WGH_NU; ++u) {
                const bool isz = u == 1;
                const int pstride = isz ? zbytes : abytes; // bytes between the hi and lo planes
                const float s = isz ? s_z : s_a;
                char *o = dst + (isz ? 2 * abytes + pz0 * 64 : pa0 * 64) + cg * 16;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    vu4 hi, lw;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        uint32_t th, tl;
                        split2u(__uint_as_float(R.d[u][2 * e][j]) * s, __uint_as_float(R.d[u][2 * e + 1][j]) * s, th, tl);
                        hi[e] = th; lw[e] = tl;
                    }
                    *reinterpret_cast<vu4 *>(o + j * 64) = hi;
                    *reinterpret_cast<vu4 *>(o + j * 64 + pstride) = lw;
                }
            }
            // zero the padded dZ pixels of both planes (16-byte chunks: 4 per pixel)
            char *dzl = dst + 2 * abytes;
            const int npad = (zpix - g.nz) * 4;
            for (int c = lt; c < 2 * npad; c += 256) {
                const int pl = c / npad, r = c - pl * npad;
                *reinterpret_cast<vu4 *>(dzl + pl * zbytes + g.nz * 64 + r * 16) = vu4{0u, 0u, 0u, 0u};
            }
        };
        Regs ra, rb;
        if (first < n) {
            issue(first, ra);
            commit(first, 0, ra);
        }
        if (first + G < n) issue(first + G, ra);
        if (first + 2 * G < n) issue(first + 2 * G, rb);
        __syncthreads();
        for (int item = first; item < n; item += 2 * G) {
            if (item + G < n) commit(item + G, 1, ra);
            if (item + 3 * G < n) issue(item + 3 * G, ra);
            __syncthreads();
            if (item + G >= n) break;
            if (item + 2 * G < n) commit(item + 2 * G, 0, rb);
            if (item + 4 * G < n) issue(item + 4 * G, rb);
            __syncthreads();
        }
        for (int w = 0; w < 6; ++w) __syncthreads(); // the compute waves' reduction barriers
        return;
    }

    f32x16 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;
    float dbacc = 0.0f;
    auto compute = [&](int item, int bi) {
        const int s = item - (a.S == 1 ? item : (int)__umulhi((uint32_t)item, a.magic_s)) * a.S;
        const int rows = min(a.RT, a.Ho - s * a.RT);
        const int npix = rows * Wo;
        const char *At = ldsw + bi * bufbytes;
        const char *Zt = At + 2 * abytes;
        const int nsteps = (npix + 15) >> 4;
        if (cw >= nsteps) return; // wave-uniform
#ifdef AUGCL_TUNING
        if (a.abl & 1) return; // ablation: the loaders alone
#endif
        // Software pipeline at tap granularity, two taps deep: the four transposed reads of tap t + 2 (two planes x two row groups)
        // are issued before the three MFMAs of tap t, the NEXT step's dZ fragments during tap 6.
        typedef s16x4 Frag[2][2]; // [plane][row group]
        auto locate = [&](int k, const char *(&pz)[2], const char *(&pa)[2]) {
#pragma unroll
            for (int rd = 0; rd < 2; ++rd) {
                const int m = 16 * k + 8 * h + 4 * rd + q; // the pixel whose row this lane supplies
                pz[rd] = Zt + m * 64 + colbyte;
                const int mc = min(m, npix - 1);            // padded pixels: dZ is zero, read any real activation
                const int yy = (int)__umulhi((uint32_t)mc, a.magic_wo), xx = mc - yy * Wo;
                pa[rd] = At + (yy * Wi + xx) * 64 + colbyte;
            }
        };
        auto rdz = [&](const char *const (&pz)[2], Frag &f) {
#pragma unroll
            for (int pl = 0; pl < 2; ++pl)
#pragma unroll
                for (int rd = 0; rd < 2; ++rd) f[pl][rd] = lds_tr_h(pz[rd] + pl * zbytes);
        };
        auto rdx = [&](const char *const (&pa)[2], int t, Frag &f) {
#pragma unroll
            for (int pl = 0; pl < 2; ++pl)
#pragma unroll
                for (int rd = 0; rd < 2; ++rd) f[pl][rd] = lds_tr_h(pa[rd] + pl * abytes + ((t / 3) * Wi + (t % 3)) * 64);
        };
        auto cat = [&](const Frag &f, int pl) {
            return __builtin_bit_cast(f16x8, (s16x8)__builtin_shufflevector(f[pl][0], f[pl][1], 0, 1, 2, 3, 4, 5, 6, 7));
        };
        const char *pz[2], *pa[2];
        Frag zr, x[3];
        locate(cw, pz, pa);
        rdz(pz, zr);
        rdx(pa, 0, x[0]);
        rdx(pa, 1, x[1]);
        for (int k = cw; k < nsteps; k += 4) { // 16 strip pixels per step; all 64 lanes stay active (tr reads)
            const f16x8 zf[2] = {cat(zr, 0), cat(zr, 1)};
#pragma unroll
            for (int j = 0; j < 8; ++j) dbacc += (float)zf[0][j] + (float)zf[1][j];
            const char *pzn[2], *pan[2];
            locate(k + 4 < nsteps ? k + 4 : k, pzn, pan); // no next step: re-read this one
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                Frag &cur = x[t % 3];
                Frag &nn = x[(t + 2) % 3];
                if (t < 7) rdx(pa, t + 2, nn);
                else rdx(pan, t - 7, nn);   // (9 = 3 x 3: the next step's taps 0, 1 land in x[0], x[1] again)
                if (t == 6) rdz(pzn, zr);   // zf holds this step's fragments by value
                __builtin_amdgcn_sched_barrier(0);
                const f16x8 xf[2] = {cat(cur, 0), cat(cur, 1)};
                f32x16 &c = acc[t];
                // smallest products first: zl xh, zh xl, zh xh
                c = __builtin_amdgcn_mfma_f32_32x32x16_f16(zf[1], xf[0], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_f16(zf[0], xf[1], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_f16(zf[0], xf[0], c, 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
            pa[0] = pan[0]; pa[1] = pan[1];
        }
    };
    __syncthreads();
    for (int item = first; item < n; item += 2 * G) {
        compute(item, 0);
        __syncthreads();
        if (item + G >= n) break;
        compute(item + G, 1);
        __syncthreads();
    }
    // ---- fixed-order reduction over the 4 compute waves through LDS, then one slab per workgroup
    float *lds = reinterpret_cast<float *>(ldsw);
    float *slab = a.partial + (size_t)blockIdx.x * WGH_SLAB;
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
        const float un_z = wg_neg ? -pow2f(-k_z) : pow2f(-k_z), un_a = pow2f(-k_a);
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) slab[(size_t)acc_row_h(r, h) * 288 + l31 * 9 + t] = acc[t][r] * un_z * un_a;
        const float other = __shfl_xor(dbacc, 32, 64); // lanes (co, h=0/1) hold pixels 8h..8h+7 of every step
        if (h == 0) slab[32 * 288 + l31] = (dbacc + other) * un_z;
    }
}

// Weight fragments of a whole stack in one small launch: block (4 operands, layer) first takes the layer's max |w| (every block
// the same 36 KB from L2; a maximum does not depend on the order), then lane l of wave j computes the fragment
// W_eff[row l & 31][k = 16 s + 8 (l >> 5) + j][tap t] of operand i = 2 t + s, scaled, as f16 (hi, lo') parts.
struct PrepX2Args {
    const float *w[32]; // (32,32,3,3) weights of the layers to prepare
    char *out;          // [n] blocks of X2_BLOCK_BYTES
    int mode;           // bit 0: 0 forward fragments, 1 backward-data (transposed, flipped taps); bit 1: residuals unscaled
    uint32_t *zero;     // the pass's per-tensor maxima: zeroed here (one launch fewer than a memset), or null
    int n_zero;
};

__global__ __launch_bounds__(256) void k_x2_prep(PrepX2Args a)
{
    __shared__ uint32_t red, red_l1[2];
    const int lane = threadIdx.x & 63, l31 = lane & 31, h = lane >> 5;
    const int i = 4 * blockIdx.x + (threadIdx.x >> 6), t = i >> 1, s = i & 1; // one operand per wave (18 = 4 x 4 + 2)
    const float *w = a.w[blockIdx.y];
    if (threadIdx.x == 0) red = red_l1[0] = red_l1[1] = 0u;
    __syncthreads();
    float m = 0.0f;
    const vf4 *w4 = reinterpret_cast<const vf4 *>(w);
#pragma unroll
    for (int c = 0; c < 9; ++c) { // 2304 chunks of 16 bytes, all in flight
        const vf4 v = w4[threadIdx.x + 256 * c];
        m = fmaxf(fmaxf(m, fmaxf(fabsf(v[0]), fabsf(v[1]))), fmaxf(fabsf(v[2]), fabsf(v[3])));
    }
    m = wave_max(m);
    if (lane == 0) atomicMax(&red, __float_as_uint(m));
    if (blockIdx.x == 4) {
        // (the block with two idle waves) the weights' largest row / column sums of absolute values: with max |input| they bound
        // max |output| of the layer BEFORE it is computed (forward: |y[co]| <= sum_{ci,t} |w[co][ci][t]| max|x| + |b[co]|;
        // backward-data: |dx[ci]| <= sum_{co,t} |w[co][ci][t]| max|dz|) -- the scale under which encoder_chain.hip stores an
        // intermediate that never leaves LDS.  8 threads per row / column, 36 weights each, fixed order.
        const int rc = threadIdx.x >> 3, part = threadIdx.x & 7;
        float sr = 0.0f, sc = 0.0f;
#pragma unroll 4
        for (int e = 0; e < 36; ++e) {
            sr += fabsf(w[rc * 288 + part * 36 + e]);
            sc += fabsf(w[(4 * part + e / 9) * 288 + rc * 9 + e % 9]);
        }
#pragma unroll
        for (int o = 1; o < 8; o <<= 1) {
            sr += __shfl_xor(sr, o, 64);
            sc += __shfl_xor(sc, o, 64);
        }
        sr = wave_max(sr);
        sc = wave_max(sc);
        if (lane == 0) {
            atomicMax(&red_l1[0], __float_as_uint(sr));
            atomicMax(&red_l1[1], __float_as_uint(sc));
        }
    }
    __syncthreads();
    if (blockIdx.x == 4 && threadIdx.x == 0) {
        uint32_t *tail = reinterpret_cast<uint32_t *>(a.out + (size_t)blockIdx.y * X2_BLOCK_BYTES + (size_t)X2_FRAG_VU4 * 16);
        tail[1] = red_l1[0]; // (block 0 writes word 0, the exponent)
        tail[2] = red_l1[1];
    }
    if (i >= 18) return;
    const int k_w = scale_exp(red);
    const float sw = pow2f(k_w);
    vu4 fh, fl;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        float v0, v1;
        if ((a.mode & 1) == 0) { // W[co = l31][ci = k][t]
            v0 = w[l31 * 288 + (16 * s + 8 * h + 2 * e) * 9 + t];
            v1 = w[l31 * 288 + (16 * s + 8 * h + 2 * e + 1) * 9 + t];
        } else {           // W'[ci = l31][co = k][t] = W[co][ci][8 - t]
            v0 = w[(16 * s + 8 * h + 2 * e) * 288 + l31 * 9 + (8 - t)];
            v1 = w[(16 * s + 8 * h + 2 * e + 1) * 288 + l31 * 9 + (8 - t)];
        }
        uint32_t th, tl;
        if (a.mode & 2) split2u(v0 * sw, v1 * sw, th, tl); // residual unscaled: every product of a group into ONE accumulator (encoder_chain.hip)
        else split2h(v0 * sw, v1 * sw, th, tl);
        fh[e] = th; fl[e] = tl;
    }
    vu4 *o = reinterpret_cast<vu4 *>(a.out + (size_t)blockIdx.y * X2_BLOCK_BYTES);
    o[i * 64 + lane] = fh;
    o[(18 + i) * 64 + lane] = fl;
    if (i == 0 && lane == 0) reinterpret_cast<uint32_t *>(o + X2_FRAG_VU4)[0] = (uint32_t)k_w; // words 1, 2: block 4 (the L1 sums)
    if (i == 1 && blockIdx.y == 0 && a.zero != nullptr)
        for (int j = lane; j < a.n_zero; j += 64) a.zero[j] = 0u;
}

// max |x| of a tensor, atomically max-ed into *out (float bits; zeroed by the caller): the gradient that enters the stack
// from the projection head has no producer on this path that could leave its maximum
__global__ __launch_bounds__(1024) void k_amax_f32(const float *x, size_t n4, size_t n, uint32_t *out)
{
    __shared__ uint32_t red;
    if (threadIdx.x == 0) red = 0u;
    __syncthreads();
    float m = 0.0f;
    const vf4 *x4 = reinterpret_cast<const vf4 *>(x);
    for (size_t i = (size_t)blockIdx.x * 1024 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 1024) {
        const vf4 v = x4[i];
        m = fmaxf(fmaxf(m, fmaxf(fabsf(v[0]), fabsf(v[1]))), fmaxf(fabsf(v[2]), fabsf(v[3])));
    }
    if (blockIdx.x == 0)
        for (size_t i = 4 * n4 + threadIdx.x; i < n; i += 1024) m = fmaxf(m, fabsf(x[i]));
    m = wave_max(m);
    if ((threadIdx.x & 63) == 0) atomicMax(&red, __float_as_uint(m));
    __syncthreads();
    if (threadIdx.x == 0) atomicMax(out, red); // one global atomic per block (same-address atomics serialise at ~12 ns each)
}

// gate bits of a tensor no kernel of this path produced (a layer that fell back to another kernel): word (b, y, x) bit c =
// x[b][c][y][x] > 0; one thread per pixel, 32 coalesced plane reads
__global__ __launch_bounds__(256) void k_gate_bits(const float *x, uint32_t *bits, int B, int HW)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (size_t)B * HW) return;
    const size_t b = i / HW, p = i - b * HW;
    const float *src = x + b * 32 * HW + p;
    uint32_t w = 0u;
#pragma unroll
    for (int c = 0; c < 32; ++c) w |= src[(size_t)c * HW] > 0.0f ? (1u << c) : 0u;
    bits[i] = w;
}

} // namespace augcl

using namespace augcl;

namespace {

int cu_count_x2()
{
    int dev = 0, v = 256;
    if (hipGetDevice(&dev) == hipSuccess) {
        int n = 0;
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0) v = n;
    }
    return v;
}

// +1: the shape is outside what these kernels cover (the caller runs the split-bf16 kernel, which has its own fallback)
template <int MODE, int GATE>
int launch_conv_x2(ConvX2Args &a, const char *name, hipStream_t s)
{
    const int Wi = a.Wi;
    const int Wl = MODE ? Wi + 2 : Wi;
    if (Wi < (MODE ? 5 : 4)) return 1; // a quad of 4 pixels per unit; MODE 1 divides by QW >= 2
    // rows per strip: the tile fits its LDS buffer and one strip is X2_NU rounds of loader units
    int rt = MODE ? (X2_MAXTILE / X2_SLOT - 2) / Wl - 2 : X2_MAXTILE / (Wl * X2_SLOT + X2_ROWPAD) - 2;
    const int QW = (Wi + 3) >> 2;
    rt = std::min(rt, (MODE ? 64 * X2_NU / QW : 256 * X2_NU / Wi) - 2); // quads of the tile's RT + 2 input rows <= 64 * X2_NU
    if (rt < 1) return 1;
    if (rt > a.Ho) rt = a.Ho;
    const int cus = cu_count_x2();
    {   // Strip height: a strip's 32-pixel tiles are dealt to 4 compute waves, so it costs ceil(tiles / 4) rounds of 54 MFMAs
        // whatever its size.  Among the heights that fit, take the one with the fewest rounds per image (+ a little per strip
        // for its barrier and halo rows); small batches additionally need enough strips to go around the CUs.
        const int rt_max = rt;
        double best_cost = 1e30;
        int best = rt_max;
        for (int r = 1; r <= rt_max; ++r) {
            const int S = (a.Ho + r - 1) / r, last = a.Ho - (S - 1) * r;
            if ((long)a.B * S < 2L * cus && r > 1) continue; // too few strips for this batch
            const double cost = (S - 1) * (double)(((r * a.Wo + 31) / 32 + 3) / 4) + (double)(((last * a.Wo + 31) / 32 + 3) / 4) + 0.2 * S;
            if (cost <= best_cost) { best_cost = cost; best = r; }
        }
#ifdef AUGCL_TUNING
        if (const char *e = getenv("AUGCL_X2_RT")) best = std::max(1, std::min(rt_max, atoi(e)));
#endif
        a.RT = best;
    }
    a.S = (a.Ho + a.RT - 1) / a.RT;
    a.n_items = a.B * a.S;
    a.magic_wo = (uint32_t)(0x100000000ull / (unsigned)a.Wo) + 1u;
    a.magic_qw = (uint32_t)(0x100000000ull / (unsigned)QW) + 1u;
    a.magic_wi = (uint32_t)(0x100000000ull / (unsigned)Wi) + 1u;
    a.magic_s = (uint32_t)(0x100000000ull / (unsigned)a.S) + 1u; // S == 1 wraps to 1: handled in the kernel
    const int tilepix = (a.RT + 2) * Wl + (MODE ? 2 : 0);
    const size_t lds = (size_t)X2_HDR + 2 * ((size_t)tilepix * X2_SLOT + (MODE ? 0 : (size_t)(a.RT + 2) * X2_ROWPAD));
    if (lds > 160 * 1024) return 1; // (cannot happen: rt is bounded by X2_MAXTILE)
#ifdef AUGCL_TUNING
    a.abl = getenv("AUGCL_X2_ABL") ? atoi(getenv("AUGCL_X2_ABL")) : 0;
#endif
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_conv32_x2<MODE, GATE>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return augcl::fail(AUGCL_E_LAUNCH, "%s: cannot set dynamic LDS %zu: %s", name, lds, hipGetErrorString(e));
    int grid = a.n_items < cus ? a.n_items : cus;
    if (a.dual) grid &= ~1; // even / odd workgroups serve the two halves
    hipLaunchKernelGGL((k_conv32_x2<MODE, GATE>), dim3(grid), dim3(512), lds, s, a);
    return augcl::check_launch(name);
}

} // namespace

extern "C" {

#ifdef AUGCL_TUNING
int augcl_tuning_x2_read_clk(unsigned long long *out4)
{
    return hipMemcpyFromSymbol(out4, HIP_SYMBOL(g_x2_clk), 32) == hipSuccess ? AUGCL_OK : AUGCL_E_LAUNCH;
}
#endif

// ---- internal entry points (encoder_stack.cpp, precision 3).  All return +1 when the shape is not covered. ----

// fragments of n (<= 32) 32 -> 32 layers into out (blocks of augcl_enc_frag_workspace_bytes(1, 1)): ONE launch, which also
// zeroes n_zero floats at `zero` (the pass's maxima; null: nothing)
int augcl_internal_x2_prep_weights(const float *const *w, int n, int mode, void *out, float *zero, int n_zero,
                                   augcl_stream_t stream)
{
    const char *name = "augcl_internal_x2_prep_weights";
    AUGCL_REQUIRE(w && out && n >= 1 && n <= 32 && mode >= 0 && mode <= 3 && aligned16(out), "%s: bad argument", name);
    static_assert((X2_FRAG_VU4 + 1) * 16 <= X2_BLOCK_BYTES, "fragment block");
    PrepX2Args a{};
    for (int i = 0; i < n; ++i) {
        AUGCL_REQUIRE(w[i] != nullptr && aligned16(w[i]), "%s: null or unaligned weight tensor", name);
        a.w[i] = w[i];
    }
    a.out = reinterpret_cast<char *>(out);
    a.mode = mode;
    a.zero = reinterpret_cast<uint32_t *>(zero);
    a.n_zero = zero ? n_zero : 0;
    hipLaunchKernelGGL(k_x2_prep, dim3(5, n), dim3(256), 0, as_stream(stream), a);
    return augcl::check_launch(name);
}

int augcl_internal_gate_bits(const float *x, void *bits, int B, int H, int W, augcl_stream_t stream)
{
    const char *name = "augcl_internal_gate_bits";
    AUGCL_REQUIRE(x && bits && B >= 1 && H >= 1 && W >= 1, "%s: bad argument", name);
    const size_t n = (size_t)B * H * W;
    hipLaunchKernelGGL(k_gate_bits, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, as_stream(stream), x,
                       reinterpret_cast<uint32_t *>(bits), B, H * W);
    return augcl::check_launch(name);
}

int augcl_internal_amax_f32(const float *x, size_t n, float *out, augcl_stream_t stream)
{
    const char *name = "augcl_internal_amax_f32";
    AUGCL_REQUIRE(x && out && aligned16(x), "%s: bad argument", name);
    const size_t n4 = n / 4;
    const int grid = (int)std::min<size_t>(256, std::max<size_t>(1, (n4 + 1023) / 1024));
    hipLaunchKernelGGL(k_amax_f32, dim3(grid), dim3(1024), 0, as_stream(stream), x, n4, n, reinterpret_cast<uint32_t *>(out));
    return augcl::check_launch(name);
}

// one layer; bias2 / wfrag2 != null: B images, the first half with (wfrag, bias), the second with (wfrag2, bias2);
// amax_in / amax_out: one float (two when dual)
int augcl_internal_x2_conv32_forward(const float *in, const float *bias, const float *bias2, float *out, int B, int Hi, int Wi,
                                     int relu, const void *wfrag, const void *wfrag2, const float *amax_in, float *amax_out,
                                     void *gate_bits_out, augcl_stream_t stream)
{
    const char *name = wfrag2 ? "augcl_enc_stack_forward_dual" : "augcl_enc_stack_forward";
    AUGCL_REQUIRE(B >= 0 && Hi >= 3 && Wi >= 3, "%s: bad shape", name);
    if (B == 0) return AUGCL_OK;
    AUGCL_REQUIRE(in && out && wfrag && amax_in, "%s: null tensor", name);
    AUGCL_REQUIRE(aligned4(in) && aligned4(out), "%s: tensors must be 4-byte aligned", name);
    AUGCL_REQUIRE(wfrag2 == nullptr || (B & 1) == 0, "%s: a two-weight-set batch must be even", name);
    if ((size_t)B * 32 * Hi * Wi >= ((size_t)1 << 30) - 8) return 1;
    ConvX2Args a{};
    a.in = in; a.bias = bias; a.mask = nullptr; a.out = out;
    a.bias2 = bias2; a.dual = wfrag2 != nullptr;
    a.wfrag = reinterpret_cast<const vu4 *>(wfrag);
    a.wfrag2 = reinterpret_cast<const vu4 *>(wfrag2);
    a.amax_in = reinterpret_cast<const uint32_t *>(amax_in);
    a.amax_out = reinterpret_cast<uint32_t *>(amax_out);
    a.bits = reinterpret_cast<uint32_t *>(gate_bits_out);
    a.B = B; a.Hi = Hi; a.Wi = Wi; a.Ho = Hi - 2; a.Wo = Wi - 2; a.relu = relu;
    AUGCL_REQUIRE(gate_bits_out == nullptr || relu, "%s: gate bits are those of a ReLU layer", name);
    return gate_bits_out ? launch_conv_x2<0, 1>(a, name, as_stream(stream)) : launch_conv_x2<0, 0>(a, name, as_stream(stream));
}

// gate_bits != null: the ReLU gate comes from the bit map the forward pass left for act_prev (act_prev itself is not read)
int augcl_internal_x2_conv32_backward_data(const float *dz, const float *act_prev, float *dact, int B, int Ho, int Wo,
                                           const void *wfrag, const float *amax_in, float *amax_out, const void *gate_bits,
                                           augcl_stream_t stream)
{
    const char *name = "augcl_enc_stack_backward";
    AUGCL_REQUIRE(B >= 0 && Ho >= 1 && Wo >= 1, "%s: bad shape", name);
    if (B == 0) return AUGCL_OK;
    AUGCL_REQUIRE(dz && dact && wfrag && amax_in, "%s: null tensor", name);
    AUGCL_REQUIRE(aligned4(dz) && aligned4(dact) && aligned4(act_prev), "%s: tensors must be 4-byte aligned", name);
    if ((size_t)B * 32 * (Ho + 2) * (Wo + 2) >= ((size_t)1 << 30) - 8) return 1;
    ConvX2Args a{};
    a.in = dz; a.bias = nullptr; a.mask = act_prev; a.out = dact;
    a.wfrag = reinterpret_cast<const vu4 *>(wfrag);
    a.amax_in = reinterpret_cast<const uint32_t *>(amax_in);
    a.amax_out = reinterpret_cast<uint32_t *>(amax_out);
    a.bits = reinterpret_cast<uint32_t *>(const_cast<void *>(gate_bits));
    a.B = B; a.Hi = Ho; a.Wi = Wo; a.Ho = Ho + 2; a.Wo = Wo + 2; a.relu = 0;
    return gate_bits ? launch_conv_x2<1, 1>(a, name, as_stream(stream)) : launch_conv_x2<1, 0>(a, name, as_stream(stream));
}

int augcl_internal_x2_conv32_backward_weights(const float *act_prev, const float *dz, float *dw, float *db, float *workspace,
                                              size_t workspace_bytes, int B, int Hi, int Wi, const float *amax_act,
                                              const float *amax_dz, augcl_stream_t stream)
{
    const char *name = "augcl_enc_stack_backward";
    AUGCL_REQUIRE(B >= 1 && Hi >= 3 && Wi >= 3, "%s: bad shape", name);
    AUGCL_REQUIRE(act_prev && dz && dw && db && workspace && amax_act && amax_dz, "%s: null tensor", name);
    AUGCL_REQUIRE(aligned4(act_prev) && aligned4(dz), "%s: tensors must be 4-byte aligned", name);
    if ((size_t)B * 32 * Hi * Wi >= ((size_t)1 << 30)) return 1;
    WgX2Args a{};
    a.act = act_prev; a.dz = dz; a.partial = workspace;
    a.amax_act = reinterpret_cast<const uint32_t *>(amax_act);
    a.amax_dz = reinterpret_cast<const uint32_t *>(amax_dz);
    a.B = B; a.Hi = Hi; a.Wi = Wi; a.Ho = Hi - 2; a.Wo = Wi - 2;
    // per buffer: 2 planes x ((RT+2)*Wi + roundup16(RT*Wo)) pixels of 64 bytes; one item = WGH_NU rounds of loader units
    const int budget_pix = 80 * 1024 / 128;
    int rt = (budget_pix - 2 * Wi - 16) / (Wi + a.Wo);
    // one loader round (64 quads) per operand: (RT+2)*Wi, RT*Wo <= 256.  (Three rounds -- two for the activations, strips half
    // again as high -- measured the same 0.50 ms over the ten layers at B = 512: the loaders' ~3.5 TB/s bound this kernel.)
    rt = std::min(rt, std::min(256 / Wi - 2, 256 / a.Wo));
    if (rt < 1 || a.Wo < 4) return 1;
    if (rt > a.Ho) rt = a.Ho;
    const int cus = cu_count_x2();
    a.S = (a.Ho + rt - 1) / rt; // the fewest strips that fit, split evenly (the kernel is bound by its loaders: see encoder_x3.hip)
    while ((long)B * a.S < 2L * cus && a.S < a.Ho) ++a.S;
    a.RT = (a.Ho + a.S - 1) / a.S;
    a.S = (a.Ho + a.RT - 1) / a.RT;
    a.n_items = B * a.S;
    a.magic_wo = (uint32_t)(0x100000000ull / (unsigned)a.Wo) + 1u;
    a.magic_s = (uint32_t)(0x100000000ull / (unsigned)a.S) + 1u; // (S == 1 is tested for in the kernel)
    const int apix = (a.RT + 2) * Wi, zpix = (a.RT * a.Wo + 15) & ~15;
    size_t lds = (size_t)2 * 2 * (apix + zpix) * 64;
    const size_t red = (size_t)(145 * 64) * 4;
    if (lds < red) lds = red;
    const int grid = a.n_items < cus ? a.n_items : cus;
#ifdef AUGCL_TUNING
    a.abl = getenv("AUGCL_X2_ABL") ? atoi(getenv("AUGCL_X2_ABL")) : 0;
#endif
    AUGCL_REQUIRE(workspace_bytes >= (size_t)(grid + 8) * WGH_SLAB * 4, "%s: workspace too small", name);
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_wgrad32_x2),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return augcl::fail(AUGCL_E_LAUNCH, "%s: cannot set dynamic LDS %zu: %s", name, lds, hipGetErrorString(e));
    hipStream_t s = as_stream(stream);
    hipLaunchKernelGGL(k_wgrad32_x2, dim3(grid), dim3(512), lds, s, a);
    if (int rc = augcl::check_launch(name)) return rc;
    return augcl::reduce_slabs(ReduceDesc{workspace, workspace + (size_t)grid * WGH_SLAB, dw, db, grid, WGH_SLAB, 32 * 288}, s, name);
}

} // extern "C"
