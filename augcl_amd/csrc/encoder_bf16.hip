// encoder_bf16.hip -- bf16 PERFORMANCE MODE of the shared conv encoder (BASELINE config 2 "bf16").
//
// Same algorithm as encoder_kernels.hip (f32, the parity path), different arithmetic and layout:
//   * activations live in HBM as bf16 NHWC (one pixel = its 32 channels = 64 contiguous bytes),
//   * products run on v_mfma_f32_32x32x16_bf16 (16x the f32 MFMA rate), accumulation stays f32,
//   * weights / biases stay f32 masters and are rounded to bf16 (RNE) into registers per launch.
// With 18 MFMAs (576 cycles) per 32-pixel x 32-channel tile the layers become HBM-bound, so the kernel is
// organised around the data movement: 4 LOADER waves stream the next strip's pixels (16-byte chunks,
// 10 in flight per lane) into the other half of a double-buffered, XOR-swizzled LDS tile while 4 COMPUTE
// waves read their MFMA B-fragments with one conflict-free ds_read_b128 each (8 channels of one pixel).
// NOT the parity path: results differ from the f32 reference at bf16 resolution (~3e-3 relative).
#include "augcl_common.h"

#include <algorithm>

namespace augcl {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ int acc_row_b(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

struct ConvBfArgs {
    const uint4 *in;      // bf16 NHWC (B,Hi,Wi,32) as 16-byte chunks (4 per pixel)
    const float *w;       // (32,32,3,3) f32 master weights
    const float *bias;    // (32) or null              [MODE 0]
    const uint4 *mask;    // a_prev bf16 NHWC (B,Ho,Wo,32) or null [MODE 1]
    void *out;            // bf16 NHWC (B,Ho,Wo,32), or f32 NCHW (B,32,Ho,Wo) when out_f32
    int B, Hi, Wi, Ho, Wo, RT, S, relu, n_items, out_f32;
    uint32_t magic_wi, magic_wo; // floor(2^32 / W) + 1: x / W == umulhi(x, magic) for the pixel counts used here
    // two-weight-set forward: images [B/2, B) use w2 / bias2; even workgroups serve the first half, odd the second
    const float *w2, *bias2;
    int dual;
};

typedef uint32_t vu4 __attribute__((ext_vector_type(4))); // native vectors: register arrays of them stay in registers
typedef float vf4 __attribute__((ext_vector_type(4)));
constexpr int CV_U = 10;        // tile chunks (16 B) per loader lane per strip: 256 x 10 x 16 B = 40 KB
constexpr int PIXB = 80;        // LDS bytes per pixel slot: 64 data + 16 pad.  16 consecutive pixels x one 16-byte
                                // chunk then fall into 16 distinct bank quads (20 p mod 64), and every tap of a
                                // tile is base + a constant: no per-read address arithmetic
constexpr int WST_ROW = 292;    // weight staging row pitch in floats (16-byte aligned rows, conflict-free reads)
constexpr int WST_BYTES = 32 * WST_ROW * 4;

// MODE 0: forward (valid 3x3).  MODE 1: backward-data = full correlation of dZ with the flipped, transposed
// weights; its LDS tile keeps every dZ row with 2 zero pixels in front (which double as the 2 behind the previous
// row) and zero rows outside the image, so no tap needs masking.
typedef short s16x2 __attribute__((ext_vector_type(2)));
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
typedef uint32_t vu2 __attribute__((ext_vector_type(2)));

template <int MODE, bool RELU, bool OUT_F32>
__global__ __launch_bounds__(512, 2) void k_conv32_bf16(ConvBfArgs a)
{
    extern __shared__ __attribute__((aligned(16))) char ldsc[]; // 2 x {tile [(RT+2) rows][Wl pixels][80 B], gate words}
    constexpr int P = MODE ? 2 : 0;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool loader = wave >= 4;
    const int cw = wave & 3;
    const int l31 = lane & 31, h = lane >> 5;
    const int Wi = a.Wi;
    const int Wl = MODE ? Wi + 2 : Wi;                     // LDS row pitch in pixels
    const int tilepix = (a.RT + 2) * Wl + (MODE ? 2 : 0);
    const int bufbytes = tilepix * PIXB;
    float *lbias = reinterpret_cast<float *>(ldsc + max(2 * bufbytes, bufbytes + WST_BYTES));
    const int grp = a.dual ? (int)(blockIdx.x & 1) : 0, nhalf = a.n_items >> 1;
    const int first = a.dual ? grp * nhalf + (int)(blockIdx.x >> 1) : xcd_slot((int)blockIdx.x, (int)gridDim.x);
    const int NG = a.dual ? (int)(gridDim.x >> 1) : (int)gridDim.x;
    const int n = a.dual ? (grp + 1) * nhalf : a.n_items;

    // ---- weights: all 512 threads copy the 36 KB of f32 masters into LDS with coalesced 16-byte loads (the region
    // of tile buffer 1, which is first written after the second barrier); the compute waves then pick their
    // MFMA A-fragments out of LDS.  (Per-lane global gathers cost ~4 us per launch: 2304 cache lines per wave.)
    {
        float *wst = reinterpret_cast<float *>(ldsc + bufbytes);
        const vf4 *w4 = reinterpret_cast<const vf4 *>(grp ? a.w2 : a.w);
        const float *bsrc = grp ? a.bias2 : a.bias;
        for (int i = tid; i < 32 * 288 / 4; i += 512) {
            const int row = (4 * i) / 288, col = 4 * i - row * 288;
            *reinterpret_cast<vf4 *>(wst + row * WST_ROW + col) = w4[i];
        }
        if (tid < 32) lbias[tid] = (MODE == 0 && bsrc) ? bsrc[tid] : 0.0f;
    }
    __syncthreads();

    // The two roles run DIFFERENT code with the same barrier sequence (2 + one per item): the branch is
    // wave-uniform, and the separate branches keep the prefetch registers and the weight fragments apart.
    if (loader) {
        // One strip = ONE round of loads (the host sizes RT for it): CV_U tile chunks per lane.
        // The loader runs two strips ahead of the compute
        // waves: strip i is read from LDS buffer i&1, strip i+1 is written into the other buffer from registers
        // whose loads were issued two iterations ago, strips i+2 / i+3 are in flight in the two register sets.
        const int lt = cw * 64 + lane; // 0..255 among the loader threads
        struct Regs {
            vu4 d[CV_U];
        };
        struct Geo {
            int rows, r0, r1, rbase, nchunk;
            const vu4 *src;
        };
        auto geom = [&](int item) {
            Geo g;
            const int b = item / a.S, s = item - b * a.S;
            const int y0 = s * a.RT;
            g.rows = min(a.RT, a.Ho - y0);
            g.rbase = y0 - P;                                   // image row of tile row 0
            g.r0 = MODE ? max(0, g.rbase) : y0;                 // image rows [r0, r1) exist
            g.r1 = MODE ? min(a.Hi, y0 + g.rows) : y0 + g.rows + 2;
            g.nchunk = (g.r1 - g.r0) * Wi * 4;
            g.src = reinterpret_cast<const vu4 *>(a.in) + ((size_t)(b * a.Hi + g.r0) * Wi) * 4;
            return g;
        };
        auto issue = [&](int item, Regs &R) {
            const Geo g = geom(item);
#pragma unroll
            for (int u = 0; u < CV_U; ++u) R.d[u] = g.src[min(u * 256 + lt, g.nchunk - 1)]; // branch-free (idempotent tail)
        };
        auto commit = [&](int item, int bi, const Regs &R) {
            const Geo g = geom(item);
            char *dst = ldsc + bi * bufbytes;
#pragma unroll
            for (int u = 0; u < CV_U; ++u) {
                const int q = min(u * 256 + lt, g.nchunk - 1);
                int lp = q >> 2; // MODE 0: the tile is the contiguous run of rows
                if (MODE == 1) {
                    const int rr = (int)__umulhi((uint32_t)lp, a.magic_wi), x = lp - rr * Wi;
                    lp = (rr + g.r0 - g.rbase) * Wl + x + 2;
                }
                *reinterpret_cast<vu4 *>(dst + lp * PIXB + (q & 3) * 16) = R.d[u];
            }
            if (MODE == 1) {
                const vu4 zero = {0u, 0u, 0u, 0u};
                // the 2 pad pixels in front of every tile row (and behind the last one)
                for (int id = lt; id < (g.rows + 3) * 8; id += 256)
                    *reinterpret_cast<vu4 *>(dst + ((id >> 3) * Wl + ((id >> 2) & 1)) * PIXB + (id & 3) * 16) = zero;
                // tile rows above / below the image
                const int top = g.r0 - g.rbase, bot = g.rbase + g.rows + 2 - g.r1, rowch = Wi * 4;
                for (int id = lt; id < (top + bot) * rowch; id += 256) {
                    const int jr = (int)__umulhi((uint32_t)(id >> 2), a.magic_wi), x = (id >> 2) - jr * Wi;
                    const int j = jr < top ? jr : (g.r1 - g.rbase) + (jr - top);
                    *reinterpret_cast<vu4 *>(dst + (j * Wl + x + 2) * PIXB + (id & 3) * 16) = zero;
                }
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
            if (item + NG < n) commit(item + NG, 1, ra); // even half: the compute waves read buffer 0
            if (item + 3 * NG < n) issue(item + 3 * NG, ra);
            __syncthreads();
            if (item + NG >= n) break;
            if (item + 2 * NG < n) commit(item + 2 * NG, 0, rb); // odd half: they read buffer 1
            if (item + 4 * NG < n) issue(item + 4 * NG, rb);
            __syncthreads();
        }
        return;
    }

    // ---- compute waves.  A fragment of (tap t, channel half s): W_eff[row l31][k = 16s + 8h + j][t], j = 0..7
    bf16x8 wfrag[18];
    {
        const float *wst = reinterpret_cast<const float *>(ldsc + bufbytes);
        if (MODE == 0) { // W[co = l31][ci = k][t]: 72 contiguous floats per channel half
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                vf4 tmp[18];
                const vf4 *src = reinterpret_cast<const vf4 *>(wst + l31 * WST_ROW + (16 * s + 8 * h) * 9);
#pragma unroll
                for (int v = 0; v < 18; ++v) tmp[v] = src[v];
#pragma unroll
                for (int t = 0; t < 9; ++t)
#pragma unroll
                    for (int j = 0; j < 8; ++j) wfrag[t * 2 + s][j] = (__bf16)tmp[(j * 9 + t) >> 2][(j * 9 + t) & 3];
            }
        } else { // W[co = k][ci = l31][8 - t]
#pragma unroll
            for (int t = 0; t < 9; ++t)
#pragma unroll
                for (int s = 0; s < 2; ++s)
#pragma unroll
                    for (int j = 0; j < 8; ++j)
                        wfrag[t * 2 + s][j] = (__bf16)wst[(16 * s + 8 * h + j) * WST_ROW + l31 * 9 + (8 - t)];
        }
    }
    const int rowb = Wl * PIXB; // LDS bytes between tile rows
    f32x16 bias16; // accumulator seed: the first MFMA of a tile takes it as its C operand (no per-tile copies)
#pragma unroll
    for (int r = 0; r < 16; ++r) bias16[r] = lbias[acc_row_b(r, h)];
    auto compute = [&](int item, int bi) {
        const int b = item / a.S, s = item - b * a.S;
        const int y0 = s * a.RT;
        const int rows = min(a.RT, a.Ho - y0);
        const int npix = rows * a.Wo;
        const char *tile = ldsc + bi * bufbytes;
        const int ntiles = (npix + 31) >> 5;
        if (cw >= ntiles) return; // wave-uniform
        // this lane's pixel of tile tl: index m in the strip, (yy, xx), and the LDS byte of its top-left tap
        // (both modes: tile row yy + ky, tile pixel xx + kx -- backward-data's +2 shift cancels its -2 taps)
        auto locate = [&](int tl, int &m_, bool &valid_, int &yy_, int &xx_, int &off_) {
            m_ = tl * 32 + l31;
            valid_ = m_ < npix;
            if (!valid_) m_ = npix - 1;
            yy_ = (int)__umulhi((uint32_t)m_, a.magic_wo);
            xx_ = m_ - yy_ * a.Wo;
            off_ = (yy_ * Wl + xx_) * PIXB + h * 16;
        };
        // operand i = (tap t = i/2, channel half s2 = i&1): one ds_read_b128 at row base + constant.  The 18 reads
        // of a tile run NINE MFMAs ahead of their use, across tile boundaries (an MFMA never waits for LDS).
        auto rd = [&](int off_, int i) {
            const int t = i >> 1;
            return *reinterpret_cast<const vu4 *>(tile + off_ + (t / 3) * rowb + ((t % 3) * PIXB + (i & 1) * 32));
        };
        int m, yy, xx, off;
        bool valid;
        locate(cw, m, valid, yy, xx, off);
        vu4 raw[18];
#pragma unroll
        for (int i = 0; i < 9; ++i) raw[i] = rd(off, i);
        for (int tl = cw; tl < ntiles; tl += 4) {
            int m2, yy2, xx2, off2;
            bool valid2;
            locate(tl + 4 < ntiles ? tl + 4 : tl, m2, valid2, yy2, xx2, off2); // no next tile: re-read this one
            // backward-data: the ReLU gate of the layer below is (a_prev != 0); this lane's 16 channels of its pixel
            // are 4 x 8 bytes of a_prev, fetched now and used after the MFMAs
            vu2 gsrc[4];
            if (MODE == 1 && a.mask != nullptr) {
                const vu2 *gp = reinterpret_cast<const vu2 *>(a.mask) + ((size_t)(b * a.Ho + y0 + yy) * a.Wo + xx) * 8 + h;
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) gsrc[g4] = gp[2 * g4];
            }
            f32x16 acc;
#pragma unroll
            for (int i = 0; i < 18; ++i) {
                if (i < 9) raw[i + 9] = rd(off, i + 9);
                else raw[i - 9] = rd(off2, i - 9);
                __builtin_amdgcn_sched_barrier(0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wfrag[i], __builtin_bit_cast(bf16x8, raw[i]),
                                                              i == 0 ? bias16 : acc, 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
            if (valid) {
                const int y = y0 + yy;
                if (OUT_F32) { // last layer: f32 NCHW for the projection
                    float *o = reinterpret_cast<float *>(a.out) + ((size_t)(b * 32) * a.Ho + y) * a.Wo + xx;
                    const int chw = a.Ho * a.Wo;
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        float v = acc[r];
                        if (RELU) v = v > 0.0f ? v : 0.0f;
                        if (MODE == 1 && a.mask != nullptr) {
                            const uint32_t d = gsrc[r >> 2][(r >> 1) & 1];
                            v = ((r & 1) ? (d >> 16) & 0x7fffu : (d & 0x7fffu)) != 0 ? v : 0.0f;
                        }
                        o[acc_row_b(r, h) * chw] = v;
                    }
                } else { // bf16 NHWC: 4 x (4 consecutive channels = 8 bytes) per lane
                    vu2 *o = reinterpret_cast<vu2 *>(a.out) + ((size_t)(b * a.Ho + y) * a.Wo + xx) * 8 + h;
#pragma unroll
                    for (int g4 = 0; g4 < 4; ++g4) {
                        vu2 pk;
#pragma unroll
                        for (int d = 0; d < 2; ++d) {
                            typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
                            typedef float f32x2 __attribute__((ext_vector_type(2)));
                            const bf16x2 two = __builtin_convertvector(f32x2{acc[g4 * 4 + 2 * d], acc[g4 * 4 + 2 * d + 1]}, bf16x2);
                            uint32_t bits = __builtin_bit_cast(uint32_t, two);
                            if (RELU) // bf16 pairs as signed 16-bit integers: negative values (sign bit) -> 0
                                bits = __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(s16x2, bits), s16x2{0, 0}));
                            if (MODE == 1 && a.mask != nullptr) { // keep where a_prev != 0: x * min(a_prev bits, 1)
                                const u16x2 keep = __builtin_elementwise_min(__builtin_bit_cast(u16x2, gsrc[g4][d] & 0x7fff7fffu), u16x2{1, 1}); // -0 counts as 0
                                bits = __builtin_bit_cast(uint32_t, (u16x2)(__builtin_bit_cast(u16x2, bits) * keep));
                            }
                            pk[d] = bits;
                        }
                        o[2 * g4] = pk;
                    }
                }
            }
            m = m2; valid = valid2; yy = yy2; xx = xx2; off = off2;
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
}

// f32 NCHW (B,32,H,W) -> bf16 NHWC (B,H,W,32): the gradient entering the stack from the projection
__global__ __launch_bounds__(256) void k_nchw_f32_to_nhwc_bf16(const float *__restrict__ in, __bf16 *__restrict__ out,
                                                               int B, int HW)
{
    // one thread per (pixel, 8-channel chunk): reads are plane-strided but every 128-byte line is fully used by the
    // 32 lanes that walk consecutive pixels of one channel
    const size_t total = (size_t)B * HW * 4;
    for (size_t q = (size_t)blockIdx.x * 256 + threadIdx.x; q < total; q += (size_t)gridDim.x * 256) {
        const int c = (int)(q / ((size_t)B * HW));        // chunk-major: consecutive threads = consecutive pixels
        const size_t pix = q - (size_t)c * B * HW;        // b*HW + p
        const size_t b = pix / HW, p = pix - b * HW;
        bf16x8 v;
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = (__bf16)in[(b * 32 + 8 * c + j) * HW + p];
        reinterpret_cast<bf16x8 *>(out)[pix * 4 + c] = v;
    }
}

} // namespace augcl

using namespace augcl;

namespace {
constexpr int kBufBytes = 78 * 1024;

int cu_count_bf()
{
    int dev = 0, v = 256;
    if (hipGetDevice(&dev) == hipSuccess) {
        int n = 0;
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0) v = n;
    }
    return v;
}

template <int MODE, bool RELU, bool OUT_F32>
int launch_conv_bf16_t(ConvBfArgs &a, const char *name, hipStream_t s)
{
    using namespace augcl;
    const int Wi = a.Wi;
    const int Wl = MODE ? Wi + 2 : Wi;
    // LDS per output row: Wl pixel slots of 80 B; 2 halo rows
    const int per_row = Wl * PIXB;
    int rt = (kBufBytes - 2 * Wl * PIXB - 2 * PIXB - 16) / per_row;
    // a strip is one round of loader loads: (RT+2)*Wi pixels <= 64*CV_U
    rt = std::min(rt, 64 * CV_U / Wi - 2);
    if (rt < 1) return augcl::fail(AUGCL_E_UNSUPPORTED, "%s: image row too wide for LDS (W=%d)", name, Wi);
    if (rt > a.Ho) rt = a.Ho;
    a.S = (a.Ho + rt - 1) / rt;
    // keep every CU busy: at least 2 items per CU when the batch allows it
    const int cus = cu_count_bf();
    while ((long)a.B * a.S < 2L * cus && a.S < a.Ho) ++a.S;
    a.S = std::min(a.Ho, a.S + augcl::extra_strips(MODE ? 3 : 2, Wi)); // measured per layer width
    a.RT = (a.Ho + a.S - 1) / a.S;
    a.S = (a.Ho + a.RT - 1) / a.RT;
    a.n_items = a.B * a.S;
    a.magic_wi = (uint32_t)(0x100000000ull / (unsigned)Wi) + 1u;
    a.magic_wo = (uint32_t)(0x100000000ull / (unsigned)a.Wo) + 1u;
    const int tilepix = (a.RT + 2) * Wl + (MODE ? 2 : 0);
    const int bufbytes = tilepix * PIXB;
    const size_t lds = (size_t)std::max(2 * bufbytes, bufbytes + WST_BYTES) + 32 * 4;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_conv32_bf16<MODE, RELU, OUT_F32>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return augcl::fail(AUGCL_E_LAUNCH, "%s: cannot set dynamic LDS %zu: %s", name, lds, hipGetErrorString(e));
    int grid = a.n_items < cus ? a.n_items : cus;
    if (a.dual) grid &= ~1; // even / odd workgroups serve the two halves
    hipLaunchKernelGGL((k_conv32_bf16<MODE, RELU, OUT_F32>), dim3(grid), dim3(512), lds, s, a);
    return augcl::check_launch(name);
}

template <int MODE>
int launch_conv_bf16(ConvBfArgs &a, const char *name, hipStream_t s)
{
    if (MODE == 1) return launch_conv_bf16_t<1, false, false>(a, name, s);
    if (a.relu) return a.out_f32 ? launch_conv_bf16_t<0, true, true>(a, name, s) : launch_conv_bf16_t<0, true, false>(a, name, s);
    return a.out_f32 ? launch_conv_bf16_t<0, false, true>(a, name, s) : launch_conv_bf16_t<0, false, false>(a, name, s);
}
} // namespace

extern "C" {

int augcl_enc_bf16_conv32_forward(const void *in_bf16_nhwc, const float *w, const float *bias, void *out, int B, int Hi,
                                  int Wi, int relu, int out_f32_nchw, augcl_stream_t stream)
{
    const char *name = "augcl_enc_bf16_conv32_forward";
    AUGCL_REQUIRE(B >= 0 && Hi >= 3 && Wi >= 3, "%s: bad shape", name);
    if (B == 0) return AUGCL_OK;
    AUGCL_REQUIRE(in_bf16_nhwc && w && out, "%s: null tensor", name);
    AUGCL_REQUIRE(aligned16(in_bf16_nhwc) && aligned16(out) && aligned16(w), "%s: tensors must be 16-byte aligned", name);
    AUGCL_REQUIRE((size_t)B * 32 * Hi * Wi < ((size_t)1 << 31), "%s: tensor too large for 32-bit offsets", name);
    ConvBfArgs a{};
    a.in = reinterpret_cast<const uint4 *>(in_bf16_nhwc); a.w = w; a.bias = bias; a.mask = nullptr; a.out = out;
    a.B = B; a.Hi = Hi; a.Wi = Wi; a.Ho = Hi - 2; a.Wo = Wi - 2; a.relu = relu; a.out_f32 = out_f32_nchw;
    return launch_conv_bf16<0>(a, name, as_stream(stream));
}

int augcl_enc_bf16_conv32_backward_data(const void *dz_bf16_nhwc, const float *w, const void *act_prev_bf16_nhwc,
                                        void *dact_bf16_nhwc, int B, int Ho, int Wo, augcl_stream_t stream)
{
    const char *name = "augcl_enc_bf16_conv32_backward_data";
    AUGCL_REQUIRE(B >= 0 && Ho >= 1 && Wo >= 1, "%s: bad shape", name);
    if (B == 0) return AUGCL_OK;
    AUGCL_REQUIRE(dz_bf16_nhwc && w && dact_bf16_nhwc, "%s: null tensor", name);
    AUGCL_REQUIRE(aligned16(dz_bf16_nhwc) && aligned16(dact_bf16_nhwc) && aligned16(act_prev_bf16_nhwc) && aligned16(w),
                  "%s: tensors must be 16-byte aligned", name);
    AUGCL_REQUIRE((size_t)B * 32 * (Ho + 2) * (Wo + 2) < ((size_t)1 << 31), "%s: tensor too large", name);
    ConvBfArgs a{};
    a.in = reinterpret_cast<const uint4 *>(dz_bf16_nhwc); a.w = w; a.bias = nullptr;
    a.mask = reinterpret_cast<const uint4 *>(act_prev_bf16_nhwc); a.out = dact_bf16_nhwc;
    a.B = B; a.Hi = Ho; a.Wi = Wo; a.Ho = Ho + 2; a.Wo = Wo + 2; a.relu = 0; a.out_f32 = 0;
    return launch_conv_bf16<1>(a, name, as_stream(stream));
}

// internal (encoder_stack.cpp): B2 images, the first half with (w, bias), the second with (w2, bias2)
int augcl_internal_bf16_conv32_forward_dual(const void *in_bf16_nhwc, const float *w, const float *bias, const float *w2,
                                            const float *bias2, void *out, int B2, int Hi, int Wi, int relu,
                                            int out_f32_nchw, augcl_stream_t stream)
{
    const char *name = "augcl_enc_stack_forward_dual";
    AUGCL_REQUIRE(B2 >= 2 && (B2 & 1) == 0 && Hi >= 3 && Wi >= 3, "%s: bad shape", name);
    AUGCL_REQUIRE(in_bf16_nhwc && w && w2 && out, "%s: null tensor", name);
    AUGCL_REQUIRE(aligned16(in_bf16_nhwc) && aligned16(out) && aligned16(w) && aligned16(w2), "%s: tensors must be 16-byte aligned", name);
    AUGCL_REQUIRE((size_t)B2 * 32 * Hi * Wi < ((size_t)1 << 31), "%s: tensor too large for 32-bit offsets", name);
    ConvBfArgs a{};
    a.in = reinterpret_cast<const uint4 *>(in_bf16_nhwc); a.w = w; a.bias = bias; a.mask = nullptr; a.out = out;
    a.w2 = w2; a.bias2 = bias2; a.dual = 1;
    a.B = B2; a.Hi = Hi; a.Wi = Wi; a.Ho = Hi - 2; a.Wo = Wi - 2; a.relu = relu; a.out_f32 = out_f32_nchw;
    return launch_conv_bf16<0>(a, name, as_stream(stream));
}

int augcl_enc_nchw_f32_to_nhwc_bf16(const float *in, void *out_bf16_nhwc, int B, int H, int W, augcl_stream_t stream)
{
    const char *name = "augcl_enc_nchw_f32_to_nhwc_bf16";
    AUGCL_REQUIRE(B >= 0 && H > 0 && W > 0, "%s: bad shape", name);
    if (B == 0) return AUGCL_OK;
    AUGCL_REQUIRE(in && out_bf16_nhwc && aligned16(out_bf16_nhwc), "%s: bad tensor", name);
    const size_t total = (size_t)B * H * W * 4;
    const size_t blocks = (total + 255) / 256;
    hipLaunchKernelGGL(k_nchw_f32_to_nhwc_bf16, dim3((unsigned)(blocks < 8192 ? blocks : 8192)), dim3(256), 0,
                       as_stream(stream), in, reinterpret_cast<__bf16 *>(out_bf16_nhwc), B, H * W);
    return augcl::check_launch(name);
}

} // extern "C"

// =========================================================================================
// backward-weights in bf16: D_t[co][ci] = sum_p dZ[p][co] * A[p + tap_t][ci], K = 16 pixels per
// v_mfma_f32_32x32x16_bf16 (8x fewer MFMAs than the f32 path).  Both operands need 8 consecutive PIXELS
// of one channel per lane, while the NHWC tiles in LDS hold 32 channels of one pixel per 64-byte row:
// ds_read_b64_tr_b16 delivers exactly that transpose (4 rows x 16 columns per 16-lane group, column-major),
// so the tiles are staged by plain 16-byte copies and every tap is just a different row address.
// 4 loader + 4 compute waves, double-buffered {A tile, dZ tile}; partial slabs + fixed-order reduction
// as in the f32 path (bitwise reproducible).
// =========================================================================================
namespace augcl {

typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

struct WgBfArgs {
    const uint4 *act; // a_{l-1} bf16 NHWC (B,Hi,Wi,32)
    const uint4 *dz;  // bf16 NHWC (B,Ho,Wo,32)
    float *partial;   // [gridDim.x][32*288 + 32]
    int B, Hi, Wi, Ho, Wo, RT, S, n_items;
};

constexpr int WGB_SLAB = 32 * 288 + 32;

__device__ __forceinline__ s16x4 lds_tr(const char *p)
{
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3))) *)(p));
}

constexpr int WG_U = 18; // 16-byte chunks per loader lane per item: 256 lanes x 18 x 16 B = 72 KB

__global__ __launch_bounds__(512, 2) void k_wgrad32_bf16(WgBfArgs a)
{
    extern __shared__ __attribute__((aligned(16))) char ldsb[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool loader = wave >= 4;
    const int cw = wave & 3;
    const int l31 = lane & 31, h = lane >> 5;
    const int Wi = a.Wi, Wo = a.Wo;
    const int apix = (a.RT + 2) * Wi;                // A tile pixels
    const int zpix = (a.RT * Wo + 15) & ~15;         // dZ tile pixels, zero-padded to whole 16-pixel steps
    const int bufbytes = (apix + zpix) * 64;
    // this lane's part in the transposed reads: it SUPPLIES row q, columns 4pp..4pp+3 of its group's 16-channel block
    const int q = (lane & 15) >> 2, pp = lane & 3, cblk = (lane >> 4) & 1;
    const int colbyte = (16 * cblk + 4 * pp) * 2;

    // ---- loader side.  One item's A + dZ chunks fit WG_U 16-byte loads per loader lane (the host sizes RT for it),
    // so an item is ONE round of loads.  The loader runs two items ahead of the compute waves: item i is being
    // read from LDS buffer i&1, item i+1 is written into the other buffer at the start of the iteration from
    // registers whose loads were issued TWO iterations ago, and the loads of items i+2 / i+3 are in flight in the
    // two register sets -- ~2 x 70 KB per CU always outstanding, which is what hides the HBM latency.
    const int lt = cw * 64 + lane; // 0..255 among the loader threads
    auto geom = [&](int item, const vu4 *&ga, const vu4 *&gz, int &na, int &nz) {
        const int b = item / a.S, s = item - b * a.S;
        const int y0 = s * a.RT;
        const int rows = min(a.RT, a.Ho - y0);
        na = (rows + 2) * Wi * 4; // 16-byte chunks
        nz = rows * Wo * 4;
        ga = reinterpret_cast<const vu4 *>(a.act) + ((size_t)(b * a.Hi + y0) * Wi) * 4;
        gz = reinterpret_cast<const vu4 *>(a.dz) + ((size_t)(b * a.Ho + y0) * Wo) * 4;
    };
    auto issue = [&](int item, vu4(&v)[WG_U]) {
        const vu4 *ga, *gz;
        int na, nz;
        geom(item, ga, gz, na, nz);
#pragma unroll
        for (int u = 0; u < WG_U; ++u) { // branch-free: the tail re-reads the last chunk
            const int c = min(u * 256 + lt, na + nz - 1);
            v[u] = c < na ? ga[c] : gz[c - na];
        }
    };
    auto commit = [&](int item, int bi, const vu4(&v)[WG_U]) {
        const vu4 *ga, *gz;
        int na, nz;
        geom(item, ga, gz, na, nz);
        vu4 *da = reinterpret_cast<vu4 *>(ldsb + bi * bufbytes);
#pragma unroll
        for (int u = 0; u < WG_U; ++u) {
            const int c = min(u * 256 + lt, na + nz - 1);
            da[c < na ? c : apix * 4 + (c - na)] = v[u];
        }
        vu4 *dzl = da + apix * 4;
        for (int c = nz + lt; c < zpix * 4; c += 256) dzl[c] = vu4{0u, 0u, 0u, 0u}; // zero the padded pixels
    };
    const int G = gridDim.x, n = a.n_items, first = xcd_slot((int)blockIdx.x, G);
    // The two roles run DIFFERENT code with the same barrier sequence (1 + one per item + 6): the branch is
    // wave-uniform, and keeping the accumulators and the prefetch registers in separate branches keeps their
    // live ranges apart (together they would not fit the register file).
    if (loader) {
        vu4 ra[WG_U], rb[WG_U];
        if (first < n) {
            issue(first, ra);
            commit(first, 0, ra);
        }
        if (first + G < n) issue(first + G, ra);
        if (first + 2 * G < n) issue(first + 2 * G, rb);
        __syncthreads();
        for (int item = first; item < n; item += 2 * G) {
            if (item + G < n) commit(item + G, 1, ra); // even half: the compute waves read buffer 0
            if (item + 3 * G < n) issue(item + 3 * G, ra);
            __syncthreads();
            if (item + G >= n) break;
            if (item + 2 * G < n) commit(item + 2 * G, 0, rb); // odd half: they read buffer 1
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
        const int s = item % a.S;
        const int rows = min(a.RT, a.Ho - s * a.RT);
        const int npix = rows * Wo;
        const char *At = ldsb + bi * bufbytes;
        const char *Zt = At + apix * 64;
        const int nsteps = (npix + 15) >> 4;
        for (int k = cw; k < nsteps; k += 4) { // 16 strip pixels per step; all 64 lanes stay active (tr reads)
            s16x4 zf[2];
            const char *pa[2];
#pragma unroll
            for (int rd = 0; rd < 2; ++rd) {
                const int m = 16 * k + 8 * h + 4 * rd + q; // the pixel whose row this lane supplies
                zf[rd] = lds_tr(Zt + m * 64 + colbyte);
                const int mc = min(m, npix - 1);             // padded pixels: dZ is zero, read any real activation
                const int yy = mc / Wo, xx = mc - yy * Wo;
                pa[rd] = At + (yy * Wi + xx) * 64 + colbyte;
            }
            const s16x8 zfrag = __builtin_shufflevector(zf[0], zf[1], 0, 1, 2, 3, 4, 5, 6, 7);
            const bf16x8 za = __builtin_bit_cast(bf16x8, zfrag);
#pragma unroll
            for (int j = 0; j < 8; ++j) dbacc += (float)za[j];
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) { // one kernel row at a time keeps only 6 operand fragments live
                s16x4 xf[3][2];
#pragma unroll
                for (int kx = 0; kx < 3; ++kx)
#pragma unroll
                    for (int rd = 0; rd < 2; ++rd) xf[kx][rd] = lds_tr(pa[rd] + (ky * Wi + kx) * 64);
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    const s16x8 xfrag = __builtin_shufflevector(xf[kx][0], xf[kx][1], 0, 1, 2, 3, 4, 5, 6, 7);
                    acc[ky * 3 + kx] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(za, __builtin_bit_cast(bf16x8, xfrag),
                                                                               acc[ky * 3 + kx], 0, 0, 0);
                }
            }
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
    float *lds = reinterpret_cast<float *>(ldsb);
    float *slab = a.partial + (size_t)blockIdx.x * WGB_SLAB;
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
            for (int r = 0; r < 16; ++r) slab[(size_t)acc_row_b(r, h) * 288 + l31 * 9 + t] = acc[t][r];
        const float other = __shfl_xor(dbacc, 32, 64); // lanes (co, h=0/1) hold pixels 8h..8h+7 of every step
        if (h == 0) slab[32 * 288 + l31] = dbacc + other;
    }
}

} // namespace augcl

extern "C" int augcl_enc_bf16_conv32_backward_weights_mfma(const void *act_prev_bf16_nhwc, const void *dz_bf16_nhwc,
                                                           float *dw, float *db, float *workspace,
                                                           size_t workspace_bytes, int B, int Hi, int Wi,
                                                           augcl_stream_t stream)
{
    const char *name = "augcl_enc_bf16_conv32_backward_weights_mfma";
    AUGCL_REQUIRE(B >= 1 && Hi >= 3 && Wi >= 3, "%s: bad shape", name);
    AUGCL_REQUIRE(act_prev_bf16_nhwc && dz_bf16_nhwc && dw && db && workspace, "%s: null tensor", name);
    AUGCL_REQUIRE(aligned16(act_prev_bf16_nhwc) && aligned16(dz_bf16_nhwc), "%s: tensors must be 16-byte aligned", name);
    WgBfArgs a{};
    a.act = reinterpret_cast<const uint4 *>(act_prev_bf16_nhwc);
    a.dz = reinterpret_cast<const uint4 *>(dz_bf16_nhwc);
    a.partial = workspace;
    a.B = B; a.Hi = Hi; a.Wi = Wi; a.Ho = Hi - 2; a.Wo = Wi - 2;
    // per buffer: (RT+2)*Wi + roundup16(RT*Wo) pixels of 64 bytes
    int rt = (kBufBytes / 64 - 2 * Wi - 16) / (Wi + a.Wo);
    const int rt_round = (256 * WG_U / 4 - 2 * Wi) / (Wi + a.Wo); // (RT+2)*Wi + RT*Wo pixels <= one round of loads
    if (rt > rt_round) rt = rt_round;
    AUGCL_REQUIRE(rt >= 1, "%s: image row too wide for LDS", name);
    if (rt > a.Ho) rt = a.Ho;
    a.S = (a.Ho + rt - 1) / rt;
    const int cus = cu_count_bf();
    while ((long)B * a.S < 2L * cus && a.S < a.Ho) ++a.S;
    a.RT = (a.Ho + a.S - 1) / a.S;
    a.S = (a.Ho + a.RT - 1) / a.RT;
    a.n_items = B * a.S;
    const int apix = (a.RT + 2) * Wi, zpix = (a.RT * a.Wo + 15) & ~15;
    size_t lds = (size_t)2 * (apix + zpix) * 64;
    const size_t red = (size_t)(145 * 64) * 4;
    if (lds < red) lds = red;
    const int grid = a.n_items < cus ? a.n_items : cus;
    AUGCL_REQUIRE(workspace_bytes >= (size_t)(grid + 8) * WGB_SLAB * 4, "%s: workspace too small", name);
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_wgrad32_bf16),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return augcl::fail(AUGCL_E_LAUNCH, "%s: cannot set dynamic LDS %zu: %s", name, lds, hipGetErrorString(e));
    hipStream_t s = as_stream(stream);
    hipLaunchKernelGGL(k_wgrad32_bf16, dim3(grid), dim3(512), lds, s, a);
    if (int rc = augcl::check_launch(name)) return rc;
    return augcl::reduce_slabs(ReduceDesc{workspace, workspace + (size_t)grid * WGB_SLAB, dw, db, grid, WGB_SLAB, 32 * 288}, s, name);
}
