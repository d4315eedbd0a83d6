// encoder_x3.hip -- the 32 -> 32 layers of the shared conv encoder (src/algorithms/modules.py:145-165) with
// FLOAT32-ACCURATE products on the bf16 matrix pipe: the parity path past the f32-MFMA wall.
//
// v_mfma_f32_32x32x2_f32 runs at the f32 vector rate (64 FLOP/clk/SIMD), v_mfma_f32_32x32x16_bf16 at 16x that.  Every
// float32 is EXACTLY the sum of three bf16 numbers (24 significand bits = 3 x 8: x = hi + mid + lo with hi = bf16(x),
// mid = bf16(x - hi), lo = bf16(x - hi - mid), each residual exact in f32), a bf16 x bf16 product is exact in f32, and
// the MFMA accumulates in f32.  Of the nine cross products of w = wh + wm + wl and x = xh + xm + xl the six with
// weight >= 2^-16 -- hh, hm, mh, hl, lh, mm -- carry everything float32 can resolve: measured on this chip
// (tools/x3probe.hip, K = 288, against float64) 2.6e-7 relative L2 for the six terms, the SAME for all nine, 3.0e-7 for
// the native f32 MFMA chain and 2.8e-6 with five terms.  Six bf16 MFMAs replace eight f32 MFMAs of twice the duration:
// 2.67x the matrix rate at float32 accuracy, activations / gradients stay f32 NCHW in HBM, so every other kernel
// (first layer, projection, heads) is untouched.
//
//   forward / backward-data (k_conv32_x3): 4 LOADER waves read the strip's f32 NCHW rows (dwordx4 along x, 8 channels per
//     lane), split every value on the VALU and write the three bf16 planes PIXEL-MAJOR into a double-buffered LDS tile
//     (pixel slot = hi | mid | lo x 32 channels + pad = 208 B: a tap's operand is one conflict-free ds_read_b128 at
//     row base + immediate); 4 COMPUTE waves keep the weights' hi and mid fragments in registers (144 VGPRs), read the lo
//     fragment and the three pixel fragments from LDS, issue 6 MFMAs per (tap, channel half) = 108 per 32-pixel tile
//     (3456 cycles instead of 9216) and store f32 NCHW with the f32 kernels' epilogue (pixels on lanes: coalesced
//     rows, bias / ReLU / ReLU gate fused).
//   backward-weights (k_wgrad32_x3): D_t[co][ci] = sum_p dZ[p][co] A[p + tap_t][ci], 16 pixels per MFMA; both operands are
//     split by the loaders, ds_read_b64_tr_b16 delivers the pixel-major tiles transposed; per-workgroup slabs + the
//     fixed-order reduction of the f32 path (bitwise reproducible).
#include "augcl_common.h"

#include <algorithm>

namespace augcl {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t vu4 __attribute__((ext_vector_type(4))); // native vectors: register arrays of them stay in registers
typedef float vf4 __attribute__((ext_vector_type(4)));
typedef float vf4ua __attribute__((ext_vector_type(4), aligned(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

constexpr int X3_SLOT = 208;        // LDS bytes per pixel slot: 3 planes x 64 B + 16 B pad (52 p mod 64: 16 consecutive
                                    // pixels x one 16-byte chunk fall into 16 distinct bank quads)
constexpr int X3_NU = 2;            // loader rounds per strip: 256 lanes x X3_NU units of (4 pixels x 8 channels)
constexpr int X3_WROW = 292;        // f32 weight staging row pitch (16-byte aligned rows, conflict-free reads)
constexpr int X3_WST = 32 * X3_WROW * 4;
constexpr int X3_WLF = 18 * 64 * 16; // the weights' lo fragments: [18 operands][64 lanes] x 16 B
constexpr int X3_MAXTILE = 70912;   // bytes per tile buffer: 2 x this + lo fragments + bias <= 160 KiB

__device__ __forceinline__ int acc_row_x(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

__device__ __forceinline__ uint32_t pk_bf16(float a, float b) // v_cvt_pk_bf16_f32: (lo half = a, hi half = b), RNE
{
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2{a, b}, bf16x2));
}
__device__ __forceinline__ float bf_lo(uint32_t p) { return __uint_as_float(p << 16); }
__device__ __forceinline__ float bf_hi(uint32_t p) { return __uint_as_float(p & 0xffff0000u); }

// two floats -> their (hi, mid, lo) bf16 pairs; hi + mid + lo == the float, exactly (both residuals are exact in f32)
__device__ __forceinline__ void split2(float a, float b, uint32_t &h, uint32_t &m, uint32_t &l)
{
    h = pk_bf16(a, b);
    const float ra = a - bf_lo(h), rb = b - bf_hi(h);
    m = pk_bf16(ra, rb);
    l = pk_bf16(ra - bf_lo(m), rb - bf_hi(m));
}

struct ConvX3Args {
    const float *in;     // forward: a_{l-1} (post-ReLU) (B,32,Hi,Wi);  bwd-data: dZ_l
    const float *w;      // (32,32,3,3) forward weights
    const float *bias;   // (32) or null                       [MODE 0]
    const float *mask;   // a_{l-1} (B,32,Ho,Wo) or null       [MODE 1]
    float *out;          // (B,32,Ho,Wo)
    int B, Hi, Wi, Ho, Wo, RT, S, relu, n_items;
    uint32_t magic_wo;   // floor(2^32 / Wo) + 1
    uint32_t magic_s;    // floor(2^32 / S) + 1: item / S == umulhi(item, magic_s) for S >= 2 (S == 1 is tested for)
    uint32_t magic_qw;   // MODE 1: floor(2^32 / QW) + 1, QW = quads per dZ row
    const float *w2, *bias2; // two-weight-set forward: images [B/2, B) use w2 / bias2 (odd workgroups)
    int dual;
    // weight fragments prepared by k_x3_prep (whole-stack entry points: one small launch per pass instead of a 36 KB staging +
    // 144-value split in every layer launch's prologue), or null: [3 components][18 operands][64 lanes] x 16 B
    const vu4 *wfrag, *wfrag2;
    int abl;                 // tuning builds only: ablation bits (1: no MFMA loop, 2: loaders skip the LDS stores, 4: ... the split)
    unsigned long long *dbg; // tuning builds only (-DAUGCL_TUNING): (shader clock << 8 | event id) records of workgroup 0
};

// MODE 0: forward (valid 3x3).  MODE 1: backward-data = full correlation of dZ with the flipped, transposed weights; its
// LDS tile keeps every dZ row with 2 zero pixels in front (which double as the 2 behind the previous row) and zero rows
// outside the image, so no tap needs masking.
template <int MODE>
__global__ __launch_bounds__(512, 2) void k_conv32_x3(ConvX3Args a)
{
    extern __shared__ __attribute__((aligned(16))) char ldsx[];
    // LDS: [lo fragments of the weights][bias][tile buffer 0][tile buffer 1]; the f32 weights are staged over buffer 1
    constexpr int P = MODE ? 2 : 0;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool loader = wave >= 4;
    const int cw = wave & 3;
    const int l31 = lane & 31, h = lane >> 5;
    const int Wi = a.Wi;
    const int Wl = MODE ? Wi + 2 : Wi;                     // LDS row pitch in pixels
    const int tilepix = (a.RT + 2) * Wl + (MODE ? 2 : 0);
    const int bufbytes = max(tilepix * X3_SLOT, X3_WST);
    char *wlf = ldsx;
    float *lbias = reinterpret_cast<float *>(ldsx + X3_WLF);
    char *tiles = ldsx + X3_WLF + 128;
    const int grp = a.dual ? (int)(blockIdx.x & 1) : 0, nhalf = a.n_items >> 1;
    const int first = a.dual ? grp * nhalf + (int)(blockIdx.x >> 1) : xcd_slot((int)blockIdx.x, (int)gridDim.x);
    const int NG = a.dual ? (int)(gridDim.x >> 1) : (int)gridDim.x;
    const int n = a.dual ? (grp + 1) * nhalf : a.n_items;
#ifdef AUGCL_TUNING
    // per-phase time stamps of compute wave 0 (records 0..) and loader wave 4 (records 4096..) of workgroup 0; record 8190/8191 =
    // s_memrealtime (100 MHz) at start / end of compute wave 0: shader clock = d(memtime) / d(memrealtime) x 100 MHz
    int ndbg = 0;
    unsigned long long *dbgp = (a.dbg && blockIdx.x == 0 && (wave == 0 || wave == 4)) ? a.dbg + (wave ? 4096 : 0) : nullptr;
    auto stamp = [&](int id) {
        if (dbgp != nullptr && ndbg < 4000) {
            const unsigned long long t = __builtin_amdgcn_s_memtime();
            if (lane == 0) dbgp[ndbg] = (t << 8) | (unsigned)id;
            ++ndbg;
        }
    };
    if (dbgp && wave == 0 && lane == 0) a.dbg[8190] = __builtin_amdgcn_s_memrealtime();
#define X3_STAMP(id) stamp(id)
#else
#define X3_STAMP(id)
#endif
    X3_STAMP(0);

    // ---- weights: the 4 compute waves copy the 36 KB of f32 masters into LDS with coalesced 16-byte loads (into the region
    // of tile buffer 1, which the loaders first write after the second barrier) WHILE the loaders already have the first
    // strip's loads in flight: the two memory latencies of the prologue overlap
    const vu4 *const wfrag = grp ? a.wfrag2 : a.wfrag; // prepared fragments (wave-uniform choice) or null
    if (!loader && wfrag != nullptr) {
        const float *bsrc = grp ? a.bias2 : a.bias;
        if (tid < 32) lbias[tid] = (MODE == 0 && bsrc) ? bsrc[tid] : 0.0f;
        __syncthreads();
        X3_STAMP(1);
    } else if (!loader) {
        float *wst = reinterpret_cast<float *>(tiles + bufbytes);
        const vf4 *w4 = reinterpret_cast<const vf4 *>(grp ? a.w2 : a.w);
        const float *bsrc = grp ? a.bias2 : a.bias;
        vf4 wv[9];
#pragma unroll
        for (int u = 0; u < 9; ++u) wv[u] = w4[tid + 256 * u]; // 2304 chunks = 9 per thread, all in flight at once
#pragma unroll
        for (int u = 0; u < 9; ++u) {
            const int i = tid + 256 * u, row = (4 * i) / 288, col = 4 * i - row * 288;
            *reinterpret_cast<vf4 *>(wst + row * X3_WROW + col) = wv[u];
        }
        if (tid < 32) lbias[tid] = (MODE == 0 && bsrc) ? bsrc[tid] : 0.0f;
        __syncthreads();
        X3_STAMP(1);
    }

    // The two roles run DIFFERENT code with the same barrier sequence (2 + one per item): the branch is wave-uniform, and
    // the separate branches keep the loaders' prefetch registers and the compute waves' weight fragments apart.
    if (loader) {
        // unit = (4 consecutive pixels of the strip's input rows) x (8 channels): 8 dwordx4 loads (4-byte aligned: rows
        // are W floats wide), then per pixel one split of its 8 channels and three 16-byte LDS stores.  A strip is
        // X3_NU rounds; the loader runs TWO strips ahead: strip i is read from LDS buffer i&1, strip i+1 is written into
        // the other buffer from registers whose loads were issued two iterations ago, strips i+2 / i+3 are in flight.
        const int lt = cw * 64 + lane; // 0..255 among the loader threads
        struct Regs {
            vu4 d[X3_NU][8];
        };
        // Geometry of an item lives in SGPRs (item is wave-uniform; divisions are multiplications by host-made reciprocals) and
        // the loads are BUFFER loads: address = resource base + this lane's 32-bit byte offset + a scalar offset that carries
        // image, first row and channel -- two vector instructions per unit instead of a 64-bit vector address per load.  (The
        // first version spent ~3500 of a strip's ~11000 cycles in the loaders' integer arithmetic: tools/x3_stamps.py.)
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
        // unit u of this lane -> (byte offset of its 4 pixels inside the lane's first channel plane, LDS byte of the first pixel)
        auto locate = [&](const Geo &g, int u, unsigned &voff, int &lb) {
            if (MODE == 0) { // the strip's input rows are one contiguous run: quad qd = pixels 4 qd .. 4 qd + 3 of it
                const int p0 = min(4 * (q0 + 64 * u), g.nrow * Wi - 4); // units past the run repeat its last quad (idempotent)
                voff = lane_c + 4u * (unsigned)p0;
                lb = p0 * X3_SLOT + cg * 16;
            } else {
                const int qd = min(q0 + 64 * u, g.nrow * QW - 1);
                const int row = (int)__umulhi((uint32_t)qd, a.magic_qw), x0 = min(4 * (qd - row * QW), Wi - 4);
                voff = lane_c + 4u * (unsigned)(row * Wi + x0);
                lb = ((row + g.top) * Wl + 2 + x0) * X3_SLOT + cg * 16;
            }
        };
        auto issue = [&](int item, Regs &R) {
            const Geo g = geom(item);
#pragma unroll
            for (int u = 0; u < X3_NU; ++u) {
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
#pragma unroll
            for (int u = 0; u < X3_NU; ++u) {
                unsigned voff;
                int lb;
                locate(g, u, voff, lb);
                char *o = dst + lb;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    vu4 hi, mid, lo;
#ifdef AUGCL_TUNING
                    if (a.abl & 4) { // ablation: no split
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            hi[e] = R.d[u][2 * e][j]; mid[e] = R.d[u][2 * e + 1][j]; lo[e] = hi[e] ^ mid[e];
                        }
                    } else
#endif
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        uint32_t th, tm, tl;
                        split2(__uint_as_float(R.d[u][2 * e][j]), __uint_as_float(R.d[u][2 * e + 1][j]), th, tm, tl);
                        hi[e] = th; mid[e] = tm; lo[e] = tl;
                    }
#ifdef AUGCL_TUNING
                    if ((a.abl & 2) && hi[0] != 0x12345678u) continue; // ablation: no LDS stores (the values stay live)
#endif
                    *reinterpret_cast<vu4 *>(o + j * X3_SLOT) = hi;
                    *reinterpret_cast<vu4 *>(o + j * X3_SLOT + 64) = mid;
                    *reinterpret_cast<vu4 *>(o + j * X3_SLOT + 128) = lo;
                }
            }
            if (MODE == 1) {
                const vu4 zero = {0u, 0u, 0u, 0u};
                // the 2 pad pixels in front of every tile row (and behind the last one): data never lands on them, so once
                // per buffer is enough (12 chunks of 16 B per pixel slot)
                if (first_use)
                    for (int id = lt; id < (a.RT + 3) * 24; id += 256) {
                        const int pr = id / 24, rem = id - pr * 24, px = rem / 12, ch = rem - px * 12;
                        *reinterpret_cast<vu4 *>(dst + (pr * Wl + px) * X3_SLOT + ch * 16) = zero;
                    }
                // tile rows above / below the image (first / last strip)
                const int top = g.top, bot = g.rows + 2 - g.top - g.nrow, rowch = Wi * 12;
                for (int id = lt; id < (top + bot) * rowch; id += 256) {
                    const int jr = id / rowch, rem = id - jr * rowch, x = rem / 12, ch = rem - x * 12;
                    const int j = jr < top ? jr : g.nrow + jr; // below the image: tile rows top + nrow + (jr - top)
                    *reinterpret_cast<vu4 *>(dst + (j * Wl + x + 2) * X3_SLOT + ch * 16) = zero;
                }
            }
        };
#ifdef AUGCL_TUNING
        switch ((a.abl >> 3) & 3) { // tuning: loader priority 0 (product) / 1 / 2 / 3
        case 1: __builtin_amdgcn_s_setprio(1); break;
        case 2: __builtin_amdgcn_s_setprio(2); break;
        case 3: __builtin_amdgcn_s_setprio(3); break;
        default: break;
        }
#else
        // (product: default priority.  Raised, the loaders' commit took 3700 instead of 3900 cycles but the compute waves'
        // epilogue 1450 instead of 950: 3.5 % slower over the ten layer sizes -- tools/x3_sweep.py)
#endif
        Regs ra, rb;
        if (first < n) issue(first, ra);
        X3_STAMP(20);
        __syncthreads(); // (the compute waves' weight staging barrier)
        if (first < n) {
            commit(first, 0, ra, true);
            X3_STAMP(21);
        }
        if (first + NG < n) issue(first + NG, ra);
        if (first + 2 * NG < n) issue(first + 2 * NG, rb);
        X3_STAMP(22);
        __syncthreads();
        X3_STAMP(23);
        bool fresh1 = true;
        for (int item = first; item < n; item += 2 * NG) {
            X3_STAMP(20);
#ifdef AUGCL_TUNING
            asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); // (stamps only) ra's loads have landed; rb's 16 may be in flight
            X3_STAMP(24);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            X3_STAMP(25);
#endif
            if (item + NG < n) commit(item + NG, 1, ra, fresh1); // even half: the compute waves read buffer 0
            fresh1 = false;
            X3_STAMP(21);
            if (item + 3 * NG < n) issue(item + 3 * NG, ra);
            X3_STAMP(22);
            __syncthreads();
            X3_STAMP(23);
            if (item + NG >= n) break;
            X3_STAMP(20);
#ifdef AUGCL_TUNING
            asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
            X3_STAMP(24);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            X3_STAMP(25);
#endif
            if (item + 2 * NG < n) commit(item + 2 * NG, 0, rb, false); // odd half: they read buffer 1
            X3_STAMP(21);
            if (item + 4 * NG < n) issue(item + 4 * NG, rb);
            X3_STAMP(22);
            __syncthreads();
            X3_STAMP(23);
        }
        return;
    }

    // ---- compute waves.  Operand i = (tap t = i / 2, channel half s = i & 1); A fragment: W_eff[row l31][k = 16 s + 8 h + j][t].
    // hi and mid parts live in registers for the whole launch, the lo part (one of the six products) is read from LDS.
    // The fragments of the ODD operands (channels 16..31) are NEGATED and accumulate into a second accumulator; the result is
    // P - N.  Reason: the bf16 MFMA's internal accumulation truncates toward -infinity somewhere below the f32 ulp -- a bias
    // of ~ -1.5e-9 sum|w x| per output (tools/x3probe.hip), harmless per element but COHERENT over a gradient map, and the
    // backward-weights / bias sums over 10^4..10^6 nearly cancelling pixels amplify a coherent error by sqrt(#pixels)
    // (measured: db 1.2e-5 off float64 ten layers down at B = 8).  Two accumulators of opposite sign carry the same bias, so
    // it cancels in the difference (mean error -1.4e-9 -> -2e-12, rel-L2 2.6e-7 -> 1.8e-7: two shorter chains).
    bf16x8 wh[18], wm[18];
    if (wfrag != nullptr) { // 36 coalesced 16-byte loads per lane; the lo fragments go to LDS (each wave a quarter)
#pragma unroll
        for (int i = 0; i < 18; ++i) {
            wh[i] = __builtin_bit_cast(bf16x8, wfrag[i * 64 + lane]);
            wm[i] = __builtin_bit_cast(bf16x8, wfrag[(18 + i) * 64 + lane]);
            if ((i & 3) == cw) *reinterpret_cast<vu4 *>(wlf + (i * 64 + lane) * 16) = wfrag[(36 + i) * 64 + lane];
        }
    } else {
        const float *wst = reinterpret_cast<const float *>(tiles + bufbytes);
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            vf4 tmp[18]; // MODE 0: the 72 contiguous floats W[co = l31][ci = 16 s + 8 h .. + 7][0..8] (conflict-free ds_read_b128)
            if (MODE == 0) {
                const vf4 *src = reinterpret_cast<const vf4 *>(wst + l31 * X3_WROW + (16 * s + 8 * h) * 9);
#pragma unroll
                for (int v = 0; v < 18; ++v) tmp[v] = src[v];
            }
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const int i = 2 * t + s;
                vu4 fh, fm, fl;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float v0, v1;
                    if (MODE == 0) { // W[co = l31][ci = k][t]
                        v0 = tmp[((2 * e) * 9 + t) >> 2][((2 * e) * 9 + t) & 3];
                        v1 = tmp[((2 * e + 1) * 9 + t) >> 2][((2 * e + 1) * 9 + t) & 3];
                    } else {         // W'[ci = l31][co = k][t] = W[co][ci][8 - t]  (lanes 9 words apart: conflict-free)
                        v0 = wst[(16 * s + 8 * h + 2 * e) * X3_WROW + l31 * 9 + (8 - t)];
                        v1 = wst[(16 * s + 8 * h + 2 * e + 1) * X3_WROW + l31 * 9 + (8 - t)];
                    }
                    if (s) { v0 = -v0; v1 = -v1; } // the split of -v is minus the split of v (round-to-nearest-even is symmetric)
                    uint32_t th, tm, tl;
                    split2(v0, v1, th, tm, tl);
                    fh[e] = th; fm[e] = tm; fl[e] = tl;
                }
                wh[i] = __builtin_bit_cast(bf16x8, fh);
                wm[i] = __builtin_bit_cast(bf16x8, fm);
                if ((i & 3) == cw) *reinterpret_cast<vu4 *>(wlf + (i * 64 + lane) * 16) = fl; // every wave computes the same values
            }
        }
    }
    X3_STAMP(2);
#ifdef AUGCL_TUNING
    switch ((a.abl >> 5) & 3) { // tuning: compute-wave priority 0 (product) / 1 / 2 / 3
    case 1: __builtin_amdgcn_s_setprio(1); break;
    case 2: __builtin_amdgcn_s_setprio(2); break;
    case 3: __builtin_amdgcn_s_setprio(3); break;
    default: break;
    }
#endif
    const int rowb = Wl * X3_SLOT; // LDS bytes between tile rows
    const int relu_floor = __builtin_amdgcn_readfirstlane(a.relu ? 0 : (int)0x80000000); // opaque: stays ONE v_max_i32
    // raw buffer resources (base, stride 0, gfx9 dword-3 flags): offsets < 2^32 by the host's size check.  The output's
    // resource is bounded by the tensor's size, so a lane with nothing to store just names an offset beyond it (kNoStore):
    // the hardware drops the store and the epilogue needs no branch.
    const __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc(a.out, 0, (int)((unsigned)(a.B * 32 * a.Ho * a.Wo) * 4u), 0x00020000);
    constexpr unsigned kNoStore = 0xfffffff0u;
    const __amdgpu_buffer_rsrc_t rs_mask =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(MODE == 1 && a.mask ? a.mask : a.in), 0, -1, 0x00020000);
    const int chw = a.Ho * a.Wo; // 32-bit offsets: the host checks B*32*Ho*Wo < 2^30

    // A tile's 16 output rows are stored WHILE the next tile's MFMAs run (one store after the first MFMA of each of its first
    // 16 operand groups), also across the strip's barrier: bunched behind the tile they kept the matrix pipe idle for
    // 950-1450 of a tile's ~5300 cycles (tools/x3_stamps.py; without any stores the ten layers took 14 % less time).
    float res[16];            // the previous tile's results (bias, ReLU / gate applied)
    unsigned pboff = kNoStore; // ... and this lane's byte offset for them
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
        // this lane's pixel of tile tl: index m in the strip, (yy, xx), the LDS byte of its top-left tap (both modes: tile
        // row yy + ky, tile pixel xx + kx -- backward-data's +2 shift cancels its -2 taps) and its output byte offset
        auto locate = [&](int tl, bool &valid_, int &off_, unsigned &boff_) {
            int m_ = tl * 32 + l31;
            valid_ = m_ < npix;
            if (!valid_) m_ = npix - 1; // read a real pixel, discard the result
            const int yy_ = (int)__umulhi((uint32_t)m_, a.magic_wo), xx_ = m_ - yy_ * a.Wo;
            off_ = (yy_ * Wl + xx_) * X3_SLOT + h * 16;
            boff_ = (unsigned)((b * 32 * a.Ho + (y0 + yy_)) * a.Wo + xx_ + 4 * h * chw) * 4u;
        };
        struct Ops {
            vu4 xh, xm, xl, wl;
        };
        // operand i of the pixel at `off_`: three ds_read_b128 at row base + immediate (+ the lo weight fragment)
        auto rd1 = [&](int off_, int i, Ops &o, int part) { // one of the operand's four reads
            const int t = i >> 1;
            const char *p = tile + off_ + (t / 3) * rowb + ((t % 3) * X3_SLOT + (i & 1) * 32);
            if (part == 0) o.xh = *reinterpret_cast<const vu4 *>(p);
            else if (part == 1) o.xm = *reinterpret_cast<const vu4 *>(p + 64);
            else if (part == 2) o.xl = *reinterpret_cast<const vu4 *>(p + 128);
            else o.wl = *reinterpret_cast<const vu4 *>(wlf + (i * 64 + lane) * 16);
        };
        auto rd = [&](int off_, int i, Ops &o) {
#pragma unroll
            for (int part = 0; part < 4; ++part) rd1(off_, i, o, part);
        };
        bool valid;
        int off;
        unsigned boff;
        locate(cw, valid, off, boff);
        // backward-data: this lane's 16 ReLU gates (the layer input at the addresses it will store to).  Row r of the previous
        // tile is stored and gate r of the CURRENT tile requested in the same slot of the MFMA stream -- the result register
        // the store frees is the one the gate lands in -- two rows per operand group over the first eight groups, so the last
        // gate has ten groups (~2000 cycles) to arrive.
        float gv[16];
        auto side_row = [&](int r) {
            store_row(r);
            if (a.mask != nullptr)
                gv[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                                                      rs_mask, boff, (unsigned)(((r & 3) + 8 * (r >> 2)) * chw) * 4u, 0));
        };
        Ops o0, o1;
        rd(off, 0, o0);
        for (int tl = cw; tl < ntiles; tl += 4) {
            bool valid2;
            int off2;
            unsigned boff2;
            locate(tl + 4 < ntiles ? tl + 4 : tl, valid2, off2, boff2); // no next tile: re-read this one
            X3_STAMP(11);
            f32x16 acc, accn; // even operands (+w) / odd operands (-w): the result is acc - accn
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) { // accumulator seed = bias of rows 8 g4 + 4 h + (0..3)
                const vf4 b4 = *reinterpret_cast<const vf4 *>(lbias + 8 * g4 + 4 * h);
                acc[4 * g4] = b4[0]; acc[4 * g4 + 1] = b4[1]; acc[4 * g4 + 2] = b4[2]; acc[4 * g4 + 3] = b4[3];
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) accn[r] = 0.0f;
#define X3_MM(C_, A_, B_) C_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A_, __builtin_bit_cast(bf16x8, B_), C_, 0, 0, 0)
#pragma unroll
            for (int i = 0; i < 18; ++i) {
                Ops &cur = (i & 1) ? o1 : o0;
                Ops &nxt = (i & 1) ? o0 : o1;
                const int noff = i < 17 ? off : off2, ni = i < 17 ? i + 1 : 0; // the next operand (across the tile boundary)
                // One LDS read of the NEXT operand after each of the first four MFMAs (four reads bunched in front of a group
                // held the issue port when the next MFMA was due); every read is >= 4 MFMAs (128 cycles) ahead of its use.
                // Smallest products first: wm xm, wl xh, wh xl, wm xh, wh xm, wh xh.
#define X3_STEP(A_, B_, PART_)                       \
    do {                                             \
        if (i & 1) X3_MM(accn, A_, B_);              \
        else X3_MM(acc, A_, B_);                     \
        __builtin_amdgcn_sched_barrier(0);           \
        if (PART_ >= 0) rd1(noff, ni, nxt, PART_);   \
        if (MODE == 0 && PART_ == 1 && i < 16) store_row(i);                                  \
        if (MODE == 1 && (PART_ == 1 || PART_ == 3) && i < 8) side_row(2 * i + (PART_ == 3)); \
        __builtin_amdgcn_sched_barrier(0);           \
    } while (0)
                X3_STEP(wm[i], cur.xm, 1); // (xm is the first operand the next group uses)
                X3_STEP(__builtin_bit_cast(bf16x8, cur.wl), cur.xh, 3);
                X3_STEP(wh[i], cur.xl, 0);
                X3_STEP(wm[i], cur.xh, 2);
                X3_STEP(wh[i], cur.xm, -1);
                X3_STEP(wh[i], cur.xh, -1);
#undef X3_STEP
            }
#undef X3_MM
            // rows in accumulator order: co = 0,1,2,3, 8,9,10,11, ... (+4 h in the lane offset).  Forward ReLU on the bit pattern
            // is one v_max_i32 (negative floats are negative integers; floor INT_MIN = no ReLU); backward-data gates with the
            // layer input's sign.
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float v = acc[r] - accn[r];
                if (MODE == 0) v = __int_as_float(max(__float_as_int(v), relu_floor));
                else if (a.mask != nullptr) v = gv[r] > 0.0f ? v : 0.0f;
                res[r] = v;
            }
            X3_STAMP(res[0] == 12345.678f ? 99 : 12); // (reads the results: after the MFMA chain has drained)
            pboff = valid ? boff : kNoStore;
#ifdef AUGCL_TUNING
            if ((a.abl & 128) && res[0] != 12345.678f) pboff = kNoStore; // ablation: no output stores
#endif
            // (18 is even: the next tile's operand 0 sits in o0 again)
            X3_STAMP(13);
            valid = valid2; off = off2; boff = boff2;
        }
    };
    __syncthreads();
    X3_STAMP(3);
    for (int item = first; item < n; item += 2 * NG) {
        compute(item, 0);
        X3_STAMP(14);
        __syncthreads();
        X3_STAMP(15);
        if (item + NG >= n) break;
        compute(item + NG, 1);
        X3_STAMP(14);
        __syncthreads();
        X3_STAMP(15);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) store_row(r); // the last tile's results
#ifdef AUGCL_TUNING
    if (dbgp && wave == 0 && lane == 0) {
        a.dbg[8191] = __builtin_amdgcn_s_memrealtime();
        a.dbg[8189] = (unsigned long long)ndbg;
    }
#endif
}
#undef X3_STAMP

// =========================================================================================
// backward-weights: D_t[co][ci] = sum_p dZ[p][co] * A[p + tap_t][ci], K = 16 pixels per MFMA.  Both operands need 8
// consecutive PIXELS of one channel per lane while the tiles hold 32 channels of one pixel per 64-byte row:
// ds_read_b64_tr_b16 delivers exactly that transpose.  Tiles per buffer: {A hi, A mid, A lo, dZ hi, dZ mid, dZ lo}.
// =========================================================================================
struct WgX3Args {
    const float *act; // a_{l-1} f32 NCHW (B,32,Hi,Wi)
    const float *dz;  // f32 NCHW (B,32,Ho,Wo)
    float *partial;   // [gridDim.x][32*288 + 32]
    int B, Hi, Wi, Ho, Wo, RT, S, n_items;
    uint32_t magic_wo, magic_s; // floor(2^32 / Wo) + 1, floor(2^32 / S) + 1
};

constexpr int WGX_SLAB = 32 * 288 + 32;
constexpr int WGX_NU = 2; // loader rounds per item: 256 lanes x 2 units of (4 pixels x 8 channels) >= A + dZ quads x 4

__device__ __forceinline__ s16x4 lds_tr_x(const char *p)
{
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3))) *)(p));
}

__global__ __launch_bounds__(512, 2) void k_wgrad32_x3(WgX3Args a)
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
    const int bufbytes = 3 * (abytes + zbytes);
    // this lane's part in the transposed reads: it SUPPLIES row q, columns 4pp..4pp+3 of its group's 16-channel block
    const int q = (lane & 15) >> 2, pp = lane & 3, cblk = (lane >> 4) & 1;
    const int colbyte = (16 * cblk + 4 * pp) * 2;
    const int G = gridDim.x, n = a.n_items, first = xcd_slot((int)blockIdx.x, G);
    // The bf16 MFMA's accumulation is biased toward -infinity by a fraction of an ulp per instruction (see k_conv32_x3); a
    // workgroup's accumulators take hundreds of them.  Odd workgroups therefore accumulate -dW (their loaders negate dZ --
    // exact) and negate the slab when they write it: the bias of the slab sum cancels between even and odd workgroups.
    const bool wg_neg = (blockIdx.x & 1) != 0;

    if (loader) {
        // Round 0 of a lane's two units covers the activation rows, round 1 the dZ rows (the host keeps either within 256
        // units): which tensor a load reads is then known at compile time and the loads are BUFFER loads -- resource base +
        // this lane's 32-bit byte offset + a scalar offset carrying image, first row and channel -- with the item's geometry
        // in SGPRs (as in k_conv32_x3; 64-bit vector addresses and a runtime division cost this loader-bound kernel ~8 %).
        const int lt = cw * 64 + lane;
        struct Regs {
            vu4 d[WGX_NU][8];
        };
        struct Geo {
            int na, nz;
            unsigned base_a, base_z; // byte offsets of (image b, channel 0, first row)
        };
        const int cg = lt & 3, q0 = lt >> 2;
        const unsigned csA4 = (unsigned)(a.Hi * Wi) * 4u, csZ4 = (unsigned)(a.Ho * Wo) * 4u;
        const unsigned lane_a = (unsigned)cg * 8u * csA4, lane_z = (unsigned)cg * 8u * csZ4;
        const __amdgpu_buffer_rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a.act), 0, -1, 0x00020000);
        const __amdgpu_buffer_rsrc_t rs_z = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a.dz), 0, -1, 0x00020000);
        auto geom = [&](int item) { // item is wave-uniform: scalar arithmetic, the division a multiplication
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
            const int pa0 = min(4 * q0, g.na - 4), pz0 = min(4 * q0, g.nz - 4); // lanes past a run repeat its last quad
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                R.d[0][c] = __builtin_bit_cast(vu4, __builtin_amdgcn_raw_buffer_load_b128(rs_a, lane_a + 4u * (unsigned)pa0, g.base_a + (unsigned)c * csA4, 0));
                R.d[1][c] = __builtin_bit_cast(vu4, __builtin_amdgcn_raw_buffer_load_b128(rs_z, lane_z + 4u * (unsigned)pz0, g.base_z + (unsigned)c * csZ4, 0));
            }
        };
        auto commit = [&](int item, int bi, const Regs &R) {
            const Geo g = geom(item);
            char *dst = ldsw + bi * bufbytes;
            const int pa0 = min(4 * q0, g.na - 4), pz0 = min(4 * q0, g.nz - 4);
#pragma unroll
            for (int u = 0; u < WGX_NU; ++u) {
                const bool isz = u == 1;
                const int pstride = isz ? zbytes : abytes; // bytes between the hi / mid / lo planes
                char *o = dst + (isz ? 3 * abytes + pz0 * 64 : pa0 * 64) + cg * 16;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    vu4 hi, mid, lw;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float v0 = __uint_as_float(R.d[u][2 * e][j]), v1 = __uint_as_float(R.d[u][2 * e + 1][j]);
                        if (isz && wg_neg) { v0 = -v0; v1 = -v1; } // odd workgroups accumulate -dW (see the slab write)
                        uint32_t th, tm, tl;
                        split2(v0, v1, th, tm, tl);
                        hi[e] = th; mid[e] = tm; lw[e] = tl;
                    }
                    *reinterpret_cast<vu4 *>(o + j * 64) = hi;
                    *reinterpret_cast<vu4 *>(o + j * 64 + pstride) = mid;
                    *reinterpret_cast<vu4 *>(o + j * 64 + 2 * pstride) = lw;
                }
            }
            // zero the padded dZ pixels of all three planes (16-byte chunks: 4 per pixel)
            char *dzl = dst + 3 * abytes;
            const int npad = (zpix - g.nz) * 4;
            for (int c = lt; c < 3 * npad; c += 256) {
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
        const char *Zt = At + 3 * abytes;
        const int nsteps = (npix + 15) >> 4;
        if (cw >= nsteps) return; // wave-uniform
        // Software pipeline at tap granularity: the six transposed reads of tap t + 1 (three planes x two row groups) are issued
        // before the six MFMAs of tap t, the NEXT step's dZ fragments during tap 7 and its tap 0 during tap 8 -- no MFMA waits
        // for a read issued less than six MFMAs (192 cycles) earlier.  (Reading a whole kernel row's 18 fragments in front of
        // its 18 MFMAs stalled every row on the LDS latency: ~130 of 576 cycles, tools/kisa.sh.)
        typedef s16x4 Frag[3][2]; // [plane][row group]
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
            for (int pl = 0; pl < 3; ++pl)
#pragma unroll
                for (int rd = 0; rd < 2; ++rd) f[pl][rd] = lds_tr_x(pz[rd] + pl * zbytes);
        };
        auto rdx = [&](const char *const (&pa)[2], int t, Frag &f) {
#pragma unroll
            for (int pl = 0; pl < 3; ++pl)
#pragma unroll
                for (int rd = 0; rd < 2; ++rd) f[pl][rd] = lds_tr_x(pa[rd] + pl * abytes + ((t / 3) * Wi + (t % 3)) * 64);
        };
        auto cat = [&](const Frag &f, int pl) {
            return __builtin_bit_cast(bf16x8, (s16x8)__builtin_shufflevector(f[pl][0], f[pl][1], 0, 1, 2, 3, 4, 5, 6, 7));
        };
        const char *pz[2], *pa[2];
        Frag zr, x0, x1;
        locate(cw, pz, pa);
        rdz(pz, zr);
        rdx(pa, 0, x0);
        for (int k = cw; k < nsteps; k += 4) { // 16 strip pixels per step; all 64 lanes stay active (tr reads)
            const bf16x8 zf[3] = {cat(zr, 0), cat(zr, 1), cat(zr, 2)};
#pragma unroll
            for (int j = 0; j < 8; ++j) dbacc += ((float)zf[0][j] + (float)zf[1][j]) + (float)zf[2][j];
            const char *pzn[2], *pan[2];
            locate(k + 4 < nsteps ? k + 4 : k, pzn, pan); // no next step: re-read this one
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                Frag &cur = (t & 1) ? x1 : x0;
                Frag &nxt = (t & 1) ? x0 : x1;
                if (t < 8) rdx(pa, t + 1, nxt);
                else rdx(pan, 0, nxt);      // (9 is odd: the next step's tap 0 lands in x1 -- swapped back below)
                if (t == 7) rdz(pzn, zr);   // zf holds this step's fragments by value
                __builtin_amdgcn_sched_barrier(0);
                const bf16x8 xf[3] = {cat(cur, 0), cat(cur, 1), cat(cur, 2)};
                f32x16 &c = acc[t];
                // smallest products first: zm xm, zl xh, zh xl, zm xh, zh xm, zh xh
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(zf[1], xf[1], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(zf[2], xf[0], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(zf[0], xf[2], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(zf[1], xf[0], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(zf[0], xf[1], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(zf[0], xf[0], c, 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int pl = 0; pl < 3; ++pl)
#pragma unroll
                for (int rd = 0; rd < 2; ++rd) x0[pl][rd] = x1[pl][rd];
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
    float *slab = a.partial + (size_t)blockIdx.x * WGX_SLAB;
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
            for (int r = 0; r < 16; ++r) slab[(size_t)acc_row_x(r, h) * 288 + l31 * 9 + t] = wg_neg ? -acc[t][r] : acc[t][r];
        const float other = __shfl_xor(dbacc, 32, 64); // lanes (co, h=0/1) hold pixels 8h..8h+7 of every step
        if (h == 0) slab[32 * 288 + l31] = wg_neg ? -(dbacc + other) : dbacc + other;
    }
}


// Weight fragments of a whole stack in one small launch: block (operand i, layer): lane l computes the fragment
// W_eff[row l & 31][k = 16 s + 8 (l >> 5) + j][tap t] of operand i = 2 t + s exactly as k_conv32_x3's prologue does (odd operands
// negated) and stores its hi / mid / lo parts into that layer's block [3][18][64] x 16 B.
constexpr int X3_FRAG_VU4 = 3 * 18 * 64; // 16-byte entries per layer block (55296 bytes)
struct PrepX3Args {
    const float *w[32]; // (32,32,3,3) weights of the layers to prepare
    vu4 *out;           // [n][X3_FRAG_VU4]
    int mode;           // 0: forward fragments, 1: backward-data (transposed, flipped taps)
};

__global__ __launch_bounds__(64) void k_x3_prep(PrepX3Args a)
{
    const int lane = threadIdx.x, l31 = lane & 31, h = lane >> 5;
    const int i = blockIdx.x, t = i >> 1, s = i & 1;
    const float *w = a.w[blockIdx.y];
    vu4 fh, fm, fl;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        float v0, v1;
        if (a.mode == 0) { // W[co = l31][ci = k][t]
            v0 = w[l31 * 288 + (16 * s + 8 * h + 2 * e) * 9 + t];
            v1 = w[l31 * 288 + (16 * s + 8 * h + 2 * e + 1) * 9 + t];
        } else {           // W'[ci = l31][co = k][t] = W[co][ci][8 - t]
            v0 = w[(16 * s + 8 * h + 2 * e) * 288 + l31 * 9 + (8 - t)];
            v1 = w[(16 * s + 8 * h + 2 * e + 1) * 288 + l31 * 9 + (8 - t)];
        }
        if (s) { v0 = -v0; v1 = -v1; }
        uint32_t th, tm, tl;
        split2(v0, v1, th, tm, tl);
        fh[e] = th; fm[e] = tm; fl[e] = tl;
    }
    vu4 *o = a.out + (size_t)blockIdx.y * X3_FRAG_VU4;
    o[i * 64 + lane] = fh;
    o[(18 + i) * 64 + lane] = fm;
    o[(36 + i) * 64 + lane] = fl;
}

} // namespace augcl

using namespace augcl;

namespace {

int cu_count_x3()
{
    int dev = 0, v = 256;
    if (hipGetDevice(&dev) == hipSuccess) {
        int n = 0;
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0) v = n;
    }
    return v;
}

#ifdef AUGCL_TUNING
unsigned long long *g_x3_dbg_get();
#endif

// +1: the shape is outside what the x3 kernels cover (the caller runs the native f32 MFMA kernel instead)
template <int MODE>
int launch_conv_x3(ConvX3Args &a, const char *name, hipStream_t s)
{
    const int Wi = a.Wi;
    const int Wl = MODE ? Wi + 2 : Wi;
    if (Wi < (MODE ? 5 : 4)) return 1; // a quad of 4 pixels per unit; MODE 1 divides by QW >= 2
    // rows per strip: the tile fits its LDS buffer and one strip is X3_NU rounds of loader units
    int rt = (X3_MAXTILE / X3_SLOT - (MODE ? 2 : 0)) / Wl - 2;
    const int QW = (Wi + 3) >> 2;
    rt = std::min(rt, (MODE ? 64 * X3_NU / QW : 256 * X3_NU / Wi) - 2); // quads of the tile's RT + 2 input rows <= 64 * X3_NU
    if (rt < 1) return 1;
    if (rt > a.Ho) rt = a.Ho;
    const int cus = cu_count_x3();
    {   // Strip height: a strip's 32-pixel tiles are dealt to 4 compute waves, so it costs ceil(tiles / 4) rounds of 108 MFMAs
        // whatever its size.  Among the heights that fit, take the one with the fewest rounds per image (+ a little per strip
        // for its barrier and halo rows); small batches additionally need enough strips to go around the CUs.
        const int rt_max = rt;
        double best_cost = 1e30;
        int best = rt_max;
#ifdef AUGCL_TUNING
        const bool even_split = getenv("AUGCL_X3_EVEN") != nullptr; // tuning: the plain even split, for A/B
#else
        const bool even_split = false;
#endif
        for (int r = 1; r <= rt_max && !even_split; ++r) {
            const int S = (a.Ho + r - 1) / r, last = a.Ho - (S - 1) * r;
            if ((long)a.B * S < 2L * cus && r > 1) continue; // too few strips for this batch
            const double cost = (S - 1) * (double)(((r * a.Wo + 31) / 32 + 3) / 4) + (double)(((last * a.Wo + 31) / 32 + 3) / 4) + 0.12 * S;
            if (cost <= best_cost) { best_cost = cost; best = r; }
        }
        if (even_split) {
            a.S = (a.Ho + rt_max - 1) / rt_max;
            while ((long)a.B * a.S < 2L * cus && a.S < a.Ho) ++a.S;
            best = (a.Ho + a.S - 1) / a.S;
        }
        a.RT = best;
    }
    a.S = (a.Ho + a.RT - 1) / a.RT;
    a.n_items = a.B * a.S;
    a.magic_wo = (uint32_t)(0x100000000ull / (unsigned)a.Wo) + 1u;
    a.magic_qw = (uint32_t)(0x100000000ull / (unsigned)QW) + 1u;
    a.magic_s = (uint32_t)(0x100000000ull / (unsigned)a.S) + 1u; // S == 1 wraps to 1: handled in the kernel
    const int tilepix = (a.RT + 2) * Wl + (MODE ? 2 : 0);
    const int bufbytes = std::max(tilepix * X3_SLOT, X3_WST);
    const size_t lds = (size_t)X3_WLF + 128 + 2 * (size_t)bufbytes;
    if (lds > 160 * 1024) return 1;
#ifdef AUGCL_TUNING
    a.dbg = g_x3_dbg_get();
    a.abl = getenv("AUGCL_X3_ABL") ? atoi(getenv("AUGCL_X3_ABL")) : 0;
#endif
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_conv32_x3<MODE>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return augcl::fail(AUGCL_E_LAUNCH, "%s: cannot set dynamic LDS %zu: %s", name, lds, hipGetErrorString(e));
    int grid = a.n_items < cus ? a.n_items : cus;
    if (a.dual) grid &= ~1; // even / odd workgroups serve the two halves
    hipLaunchKernelGGL((k_conv32_x3<MODE>), dim3(grid), dim3(512), lds, s, a);
    return augcl::check_launch(name);
}

} // namespace

extern "C" {

int augcl_internal_conv32_forward_dual(const float *in, const float *w, const float *bias, const float *w2,
                                       const float *bias2, float *out, int B2, int Hi, int Wi, int relu,
                                       augcl_stream_t stream);

static int x3_forward_impl(const float *in, const float *w, const float *bias, const float *w2, const float *bias2, float *out,
                           int B, int Hi, int Wi, int relu, augcl_stream_t stream, const char *name,
                           const void *wfrag = nullptr, const void *wfrag2 = nullptr)
{
    AUGCL_REQUIRE(B >= 0 && Hi >= 3 && Wi >= 3, "%s: bad shape", name);
    if (B == 0) return AUGCL_OK;
    AUGCL_REQUIRE(in && w && out, "%s: null tensor", name);
    AUGCL_REQUIRE(aligned16(w) && aligned16(w2), "%s: weights must be 16-byte aligned", name);
    AUGCL_REQUIRE(aligned4(in) && aligned4(out), "%s: tensors must be 4-byte aligned", name);
    AUGCL_REQUIRE(w2 == nullptr || (B & 1) == 0, "%s: a two-weight-set batch must be even", name);
    AUGCL_REQUIRE((size_t)B * 32 * Hi * Wi < ((size_t)1 << 30) - 8, "%s: tensor too large for 32-bit byte offsets (2^30 elements)", name);
    ConvX3Args a{};
    a.in = in; a.w = w; a.bias = bias; a.mask = nullptr; a.out = out;
    a.w2 = w2; a.bias2 = bias2; a.dual = w2 != nullptr;
    a.wfrag = reinterpret_cast<const vu4 *>(wfrag);
    a.wfrag2 = reinterpret_cast<const vu4 *>(w2 ? wfrag2 : nullptr);
    if (w2 && (wfrag == nullptr) != (wfrag2 == nullptr)) a.wfrag = a.wfrag2 = nullptr; // both sets prepared, or neither
    a.B = B; a.Hi = Hi; a.Wi = Wi; a.Ho = Hi - 2; a.Wo = Wi - 2; a.relu = relu;
    const int rc = launch_conv_x3<0>(a, name, as_stream(stream));
    if (rc <= 0) return rc;
    // rows too wide for the split tile: the native f32 MFMA kernel computes the same layer
    if (w2) return augcl_internal_conv32_forward_dual(in, w, bias, w2, bias2, out, B, Hi, Wi, relu, stream);
    return augcl_enc_conv32_forward(in, w, bias, out, B, Hi, Wi, relu, stream);
}

#ifdef AUGCL_TUNING
static thread_local unsigned long long *g_x3_dbg = nullptr;
} // extern "C"
namespace {
unsigned long long *g_x3_dbg_get() { return g_x3_dbg; }
} // namespace
extern "C" {
// tuning builds: the next x3 forward / backward-data launch of this thread writes its phase stamps to dbg (8192 u64, device)
int augcl_tuning_x3_set_stamps(void *dbg)
{
    g_x3_dbg = reinterpret_cast<unsigned long long *>(dbg);
    return AUGCL_OK;
}
#endif

int augcl_enc_x3_conv32_forward(const float *in, const float *w, const float *bias, float *out, int B, int Hi, int Wi,
                                int relu, augcl_stream_t stream)
{
    return x3_forward_impl(in, w, bias, nullptr, nullptr, out, B, Hi, Wi, relu, stream, "augcl_enc_x3_conv32_forward");
}

// internal (encoder_stack.cpp): one layer with PREPARED weight fragments (augcl_internal_x3_prep_weights; null = the kernel
// prepares its own); w2 != null: B images, the first half with (w, bias), the second with (w2, bias2)
int augcl_internal_x3_conv32_forward(const float *in, const float *w, const float *bias, const float *w2, const float *bias2,
                                     float *out, int B, int Hi, int Wi, int relu, const void *wfrag, const void *wfrag2,
                                     augcl_stream_t stream)
{
    return x3_forward_impl(in, w, bias, w2, bias2, out, B, Hi, Wi, relu, stream,
                           w2 ? "augcl_enc_stack_forward_dual" : "augcl_enc_stack_forward", wfrag, wfrag2);
}

size_t augcl_enc_frag_workspace_bytes(int n_layers, int weight_sets)
{
    if (n_layers < 1 || weight_sets < 1) return 0;
    return (size_t)n_layers * (size_t)weight_sets * X3_FRAG_VU4 * 16;
}

// internal: fragments of n (<= 32) 32 -> 32 layers into out[n][X3_FRAG_VU4] (mode 0 forward, 1 backward-data): ONE launch
int augcl_internal_x3_prep_weights(const float *const *w, int n, int mode, void *out, augcl_stream_t stream)
{
    const char *name = "augcl_internal_x3_prep_weights";
    AUGCL_REQUIRE(w && out && n >= 1 && n <= 32 && (mode == 0 || mode == 1) && aligned16(out), "%s: bad argument", name);
    PrepX3Args a{};
    for (int i = 0; i < n; ++i) {
        AUGCL_REQUIRE(w[i] != nullptr, "%s: null weight tensor", name);
        a.w[i] = w[i];
    }
    a.out = reinterpret_cast<vu4 *>(out);
    a.mode = mode;
    hipLaunchKernelGGL(k_x3_prep, dim3(18, n), dim3(64), 0, as_stream(stream), a);
    return augcl::check_launch(name);
}

static int x3_backward_data_impl(const float *dz, const float *w, const float *act_prev, float *dact, int B, int Ho, int Wo,
                                 const void *wfrag, augcl_stream_t stream);

int augcl_enc_x3_conv32_backward_data(const float *dz, const float *w, const float *act_prev, float *dact, int B, int Ho,
                                      int Wo, augcl_stream_t stream)
{
    return x3_backward_data_impl(dz, w, act_prev, dact, B, Ho, Wo, nullptr, stream);
}

// internal (encoder_stack.cpp): with prepared mode-1 fragments
int augcl_internal_x3_conv32_backward_data(const float *dz, const float *w, const float *act_prev, float *dact, int B, int Ho,
                                           int Wo, const void *wfrag, augcl_stream_t stream)
{
    return x3_backward_data_impl(dz, w, act_prev, dact, B, Ho, Wo, wfrag, stream);
}

static int x3_backward_data_impl(const float *dz, const float *w, const float *act_prev, float *dact, int B, int Ho, int Wo,
                                 const void *wfrag, augcl_stream_t stream)
{
    const char *name = "augcl_enc_x3_conv32_backward_data";
    AUGCL_REQUIRE(B >= 0 && Ho >= 1 && Wo >= 1, "%s: bad shape", name);
    if (B == 0) return AUGCL_OK;
    AUGCL_REQUIRE(dz && w && dact, "%s: null tensor", name);
    AUGCL_REQUIRE(aligned16(w), "%s: weights must be 16-byte aligned", name);
    AUGCL_REQUIRE(aligned4(dz) && aligned4(dact) && aligned4(act_prev), "%s: tensors must be 4-byte aligned", name);
    AUGCL_REQUIRE((size_t)B * 32 * (Ho + 2) * (Wo + 2) < ((size_t)1 << 30) - 8, "%s: tensor too large for 32-bit byte offsets (2^30 elements)", name);
    ConvX3Args a{};
    a.in = dz; a.w = w; a.bias = nullptr; a.mask = act_prev; a.out = dact;
    a.wfrag = reinterpret_cast<const vu4 *>(wfrag);
    a.B = B; a.Hi = Ho; a.Wi = Wo; a.Ho = Ho + 2; a.Wo = Wo + 2; a.relu = 0;
    const int rc = launch_conv_x3<1>(a, name, as_stream(stream));
    if (rc <= 0) return rc;
    return augcl_enc_conv32_backward_data(dz, w, act_prev, dact, B, Ho, Wo, stream);
}

int augcl_enc_x3_conv32_backward_weights(const float *act_prev, const float *dz, float *dw, float *db, float *workspace,
                                         size_t workspace_bytes, int B, int Hi, int Wi, augcl_stream_t stream)
{
    const char *name = "augcl_enc_x3_conv32_backward_weights";
    AUGCL_REQUIRE(B >= 1 && Hi >= 3 && Wi >= 3, "%s: bad shape", name);
    AUGCL_REQUIRE(act_prev && dz && dw && db && workspace, "%s: null tensor", name);
    AUGCL_REQUIRE(aligned4(act_prev) && aligned4(dz), "%s: tensors must be 4-byte aligned", name);
    if ((size_t)B * 32 * Hi * Wi >= ((size_t)1 << 30)) // beyond 32-bit byte offsets: the native f32 MFMA kernel
        return augcl_enc_conv32_backward_weights(act_prev, dz, dw, db, workspace, workspace_bytes, B, Hi, Wi, stream);
    WgX3Args a{};
    a.act = act_prev; a.dz = dz; a.partial = workspace;
    a.B = B; a.Hi = Hi; a.Wi = Wi; a.Ho = Hi - 2; a.Wo = Wi - 2;
    // per buffer: 3 planes x ((RT+2)*Wi + roundup16(RT*Wo)) pixels of 64 bytes; one item = WGX_NU rounds of loader units
    const int budget_pix = 80 * 1024 / 192;
    int rt = (budget_pix - 2 * Wi - 16) / (Wi + a.Wo);
    rt = std::min(rt, std::min(256 / Wi - 2, 256 / a.Wo)); // one loader round (64 quads) per operand: (RT+2)*Wi, RT*Wo <= 256
    if (rt < 1 || a.Wo < 4) // rows too wide (or too narrow) for the split tiles: the native f32 MFMA kernel
        return augcl_enc_conv32_backward_weights(act_prev, dz, dw, db, workspace, workspace_bytes, B, Hi, Wi, stream);
    if (rt > a.Ho) rt = a.Ho;
    const int cus = cu_count_x3();
    // Strip height: the fewest strips that fit, split evenly.  (Choosing the height by rounds of 4-wave steps as launch_conv_x3
    // does measured 4 % SLOWER over the ten layer sizes -- 717 vs 691 us at B = 512, tools/x3_wg_sweep.py: this kernel is bound by
    // its loaders, which split BOTH operands, and shorter strips stage more halo rows.)
    a.S = (a.Ho + rt - 1) / rt;
    while ((long)B * a.S < 2L * cus && a.S < a.Ho) ++a.S;
    a.RT = (a.Ho + a.S - 1) / a.S;
    a.S = (a.Ho + a.RT - 1) / a.RT;
    a.n_items = B * a.S;
    a.magic_wo = (uint32_t)(0x100000000ull / (unsigned)a.Wo) + 1u;
    a.magic_s = (uint32_t)(0x100000000ull / (unsigned)a.S) + 1u; // (S == 1 is tested for in the kernel)
    const int apix = (a.RT + 2) * Wi, zpix = (a.RT * a.Wo + 15) & ~15;
    size_t lds = (size_t)2 * 3 * (apix + zpix) * 64;
    const size_t red = (size_t)(145 * 64) * 4;
    if (lds < red) lds = red;
    const int grid = a.n_items < cus ? a.n_items : cus;
    AUGCL_REQUIRE(workspace_bytes >= (size_t)(grid + 8) * WGX_SLAB * 4, "%s: workspace too small", name);
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_wgrad32_x3),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return augcl::fail(AUGCL_E_LAUNCH, "%s: cannot set dynamic LDS %zu: %s", name, lds, hipGetErrorString(e));
    hipStream_t s = as_stream(stream);
    hipLaunchKernelGGL(k_wgrad32_x3, dim3(grid), dim3(512), lds, s, a);
    if (int rc = augcl::check_launch(name)) return rc;
    return augcl::reduce_slabs(ReduceDesc{workspace, workspace + (size_t)grid * WGX_SLAB, dw, db, grid, WGX_SLAB, 32 * 288}, s, name);
}

} // extern "C"
