// encoder_first_x3.hip -- the encoder's first layer (src/algorithms/modules.py:85-90, 145-155: NormalizeImg, Conv2d(9, 32, 3,
// stride 2)) for FLOAT32 observations (augmented batches: random_conv / overlay outputs are not integers) with float32-accurate
// products on the bf16 matrix pipe.  The loader waves cut every input element into three bf16 pieces (round-to-nearest, exact:
// x = x0 + x1 + x2) as it passes through them -- each element is cut ONCE and feeds ~2.25 output pixels x 32 filters, so the cut
// costs 2 us per 256 images -- the weights are cut the same way when the kernel starts, and a 32-pixel tile runs the six piece
// products that carry everything above 2^-24: 42 v_mfma_f32_32x32x16_bf16 = 1344 matrix cycles where the float32 pipe needs 41
// v_mfma_f32_32x32x2_f32 = 2624, with no per-tile conversion work in the waves that issue them.
//
// Work item = (image, strip of 6 output rows): 13 input rows x 9 channels -> three [channel][row][84] bf16 planes (57.6 KiB,
// double buffered).  K = 28 (channel, ky) PAIRS x 4 (kx = 0, 1, 2 and a pad column with a zero weight): one 8-byte LDS read per pair
// and plane at the lane's pixel base + a compile-time offset.  Epilogue as in encoder_first.hip (x 1 / scale, bias, ReLU, maximum,
// gate bits).  Two weight sets: workgroup pairs (i, i + 8) -- one XCD -- take the two sets of one input image.
#include "augcl_common.h"

namespace augcl {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float vf4 __attribute__((ext_vector_type(4)));
typedef uint32_t vu2 __attribute__((ext_vector_type(2)));
typedef uint32_t vu4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int X3F_W = 84, X3F_C = 9, X3F_HO = 41, X3F_WO = 41;
constexpr int X3F_RT = 6, X3F_S = 7;                 // 6 output rows per strip, 7 strips (the last has 5)
constexpr int X3F_ROWS = 2 * X3F_RT + 1;             // 13 input rows
constexpr int X3F_RPITCH = X3F_W * 2;                // 168 bytes per row of a plane
constexpr int X3F_CPITCH = X3F_ROWS * X3F_RPITCH;    // 2184
constexpr int X3F_PLANE = X3F_C * X3F_CPITCH;        // 19656
constexpr int X3F_TILE = 3 * X3F_PLANE;              // 58968
constexpr int X3F_NCH = 7;
constexpr int X3F_FRAG = 3 * X3F_NCH * 64 * 16;      // 21504
constexpr int X3F_LU = 10;                           // float4 per loader lane and strip (9 x 273 = 2457 <= 2560)
constexpr int X3F_LDS = 2 * X3F_TILE + X3F_FRAG + 32 * 4 + 16;

struct FirstX3Args {
    const float *in; // (B or B/2, 9, 84, 84) raw
    const float *w, *bias, *w2, *bias2;
    float *out;      // (B,32,41,41) f32 NCHW
    int B, dual, relu, n_items;
    float inv_scale;
    uint32_t *amax_out, *bits_out;
};

__device__ __forceinline__ int x3f_acc_row(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

// three bf16 pieces of four floats (round to nearest even; x = p0 + p1 + p2 exactly) -> three 8-byte quads
__device__ __forceinline__ void x3f_split4(vf4 x, vu2 &q0, vu2 &q1, vu2 &q2)
{
    typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
    const bf16x4 p0 = __builtin_convertvector(x, bf16x4);
    const vf4 r1 = x - __builtin_convertvector(p0, vf4);
    const bf16x4 p1 = __builtin_convertvector(r1, bf16x4);
    const vf4 r2 = r1 - __builtin_convertvector(p1, vf4);
    const bf16x4 p2 = __builtin_convertvector(r2, bf16x4);
    q0 = __builtin_bit_cast(vu2, p0);
    q1 = __builtin_bit_cast(vu2, p1);
    q2 = __builtin_bit_cast(vu2, p2);
}

__global__ __launch_bounds__(512, 2) void k_conv_first9_x3(FirstX3Args a)
{
    extern __shared__ __attribute__((aligned(16))) char ldsx3[];
    char *tiles = ldsx3;
    char *frags = ldsx3 + 2 * X3F_TILE;
    float *lbias = reinterpret_cast<float *>(ldsx3 + 2 * X3F_TILE + X3F_FRAG);
    uint32_t *lred = reinterpret_cast<uint32_t *>(lbias + 32);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool loader = wave >= 4;
    const int cw = wave & 3;
    const int l31 = lane & 31, h = lane >> 5;
    // item distribution as in encoder_first.hip (two weight sets: the two readers of an input image on one XCD)
    const bool pair_xcd = a.dual && (gridDim.x & 15) == 0;
    const int grp = a.dual ? (pair_xcd ? (int)((blockIdx.x >> 3) & 1) : (int)(blockIdx.x & 1)) : 0, nhalf = a.n_items >> 1;
    const int worker = pair_xcd ? (int)(blockIdx.x & 7) * (int)(gridDim.x >> 4) + (int)(blockIdx.x >> 4) : (int)(blockIdx.x >> 1);
    const int first = a.dual ? grp * nhalf + worker : xcd_slot((int)blockIdx.x, (int)gridDim.x);
    const int NG = a.dual ? (int)(gridDim.x >> 1) : (int)gridDim.x;
    const int n = a.dual ? (grp + 1) * nhalf : a.n_items;

    // every thread runs prep(); the loader waves call it BEHIND their first strip's loads
    auto prep = [&]() {
        const float *wg = grp ? a.w2 : a.w, *bsrc = grp ? a.bias2 : a.bias;
        // the 2592 weights: one coalesced copy into LDS (tile buffer 1 is first written behind the next barrier but one), then the
        // fragments are built from there (lanes = filters, stride 81 words: every bank once) -- not from 8 scattered global loads per
        // vector, three dependent rounds of them per thread (~3 us of every launch)
        float *wsrc = reinterpret_cast<float *>(tiles + X3F_TILE);
        for (int i = tid; i < 32 * 81; i += 512) wsrc[i] = wg[i];
        __syncthreads();
        for (int v = tid; v < 3 * X3F_NCH * 64; v += 512) {
            const int ln = v & 63, i = (v >> 6) % X3F_NCH, p = (v >> 6) / X3F_NCH;
            const int co = ln & 31, hh = ln >> 5;
            bf16x8 f;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int P = 4 * i + 2 * hh + (j >> 2), kx = j & 3;
                float x = (P < 27 && kx < 3) ? wsrc[co * 81 + P * 3 + kx] : 0.0f;
                __bf16 q = (__bf16)x;
                if (p >= 1) {
                    x -= (float)q;
                    q = (__bf16)x;
                }
                if (p >= 2) {
                    x -= (float)q;
                    q = (__bf16)x;
                }
                f[j] = q;
            }
            *reinterpret_cast<bf16x8 *>(frags + ((p * X3F_NCH + i) * 64 + ln) * 16) = f;
        }
        if (tid < 32) lbias[tid] = bsrc ? bsrc[tid] : 0.0f;
        if (tid < 2) lred[tid] = 0u;
    };

    if (loader) {
        const int lt = cw * 64 + lane;
        struct Regs {
            vf4 d[X3F_LU];
        };
        auto geom = [&](int item, int &b, int &s, int &upc, uint32_t &magic) {
            b = item / X3F_S;
            s = item - b * X3F_S;
            const bool last = s == X3F_S - 1;
            upc = last ? 11 * 21 : 13 * 21; // float4 per channel run: rows x 84 / 4
            magic = last ? (0xffffffffu / 231u + 1u) : (0xffffffffu / 273u + 1u);
        };
        auto issue = [&](int item, Regs &R) {
            int b, s, upc;
            uint32_t magic;
            geom(item, b, s, upc, magic);
            const int bi = (a.dual && b >= (a.B >> 1)) ? b - (a.B >> 1) : b; // both halves read the same input
            const float *src = a.in + ((size_t)bi * X3F_C * X3F_W + 2 * X3F_RT * s) * X3F_W;
#pragma unroll
            for (int u = 0; u < X3F_LU; ++u) {
                const int id = min(u * 256 + lt, X3F_C * upc - 1);
                const int c = (int)__umulhi((uint32_t)id, magic), rem = id - c * upc;
                R.d[u] = *reinterpret_cast<const vf4 *>(src + (size_t)c * (X3F_W * X3F_W) + rem * 4);
            }
        };
        auto commit = [&](int item, int bufi, const Regs &R) {
            int b, s, upc;
            uint32_t magic;
            geom(item, b, s, upc, magic);
            char *dst = tiles + bufi * X3F_TILE;
#pragma unroll
            for (int u = 0; u < X3F_LU; ++u) {
                const int id = min(u * 256 + lt, X3F_C * upc - 1);
                const int c = (int)__umulhi((uint32_t)id, magic), rem = id - c * upc;
                vu2 q0, q1, q2;
                x3f_split4(R.d[u], q0, q1, q2);
                char *o = dst + c * X3F_CPITCH + rem * 8; // (rem float4 = 4 rem elements = 8 rem bytes of bf16: rows are contiguous)
                *reinterpret_cast<vu2 *>(o) = q0;
                *reinterpret_cast<vu2 *>(o + X3F_PLANE) = q1;
                *reinterpret_cast<vu2 *>(o + 2 * X3F_PLANE) = q2;
            }
        };
        Regs ra, rb;
        if (first < n) issue(first, ra);
        prep();
        if (first < n) commit(first, 0, ra);
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

    // ---- compute waves
    prep();
    const __amdgpu_buffer_rsrc_t rs_out =
        __builtin_amdgcn_make_buffer_rsrc(a.out, 0, (int)((unsigned)a.B * 32u * X3F_HO * X3F_WO * 4u), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_bits = __builtin_amdgcn_make_buffer_rsrc(
        a.bits_out ? a.bits_out : reinterpret_cast<uint32_t *>(a.out), 0, (int)((unsigned)a.B * X3F_HO * X3F_WO * 4u), 0x00020000);
    constexpr unsigned kNoStore = 0xfffffff0u;
    constexpr int chw = X3F_HO * X3F_WO;
    const int relu_floor = __builtin_amdgcn_readfirstlane(a.relu ? 0 : (int)0x80000000);
    float vmax = 0.0f;
    __syncthreads();
    bf16x8 wf[3][X3F_NCH];
#pragma unroll
    for (int p = 0; p < 3; ++p)
#pragma unroll
        for (int i = 0; i < X3F_NCH; ++i) wf[p][i] = *reinterpret_cast<const bf16x8 *>(frags + ((p * X3F_NCH + i) * 64 + lane) * 16);
    float bias16[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) bias16[r] = lbias[x3f_acc_row(r, h)];

    auto compute = [&](int item, int bufi) {
        const int b = item / X3F_S, s = item - b * X3F_S;
        const int y0 = s * X3F_RT, rows = min(X3F_RT, X3F_HO - y0);
        const int npix = rows * X3F_WO, ntiles = (npix + 31) >> 5;
        const char *tile = tiles + bufi * X3F_TILE;
        for (int tl = cw; tl < ntiles; tl += 4) {
            int m = tl * 32 + l31;
            const bool valid = m < npix;
            if (!valid) m = npix - 1;
            const int yy = m / X3F_WO, xx = m - yy * X3F_WO;
            const char *pb = tile + (2 * yy) * X3F_RPITCH + 4 * xx; // input row 2 yy of the strip, column 2 xx (2 bytes each)
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
#pragma unroll
            for (int i = 0; i < X3F_NCH; ++i) {
                vu4 op[3];
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const int P0 = 4 * i + e, P1 = 4 * i + 2 + e;
                    const int Q0 = P0 < 27 ? P0 : 26, Q1 = P1 < 27 ? P1 : 26; // pair 27 has a zero weight: any finite operand
                    const int off0 = (Q0 / 3) * X3F_CPITCH + (Q0 % 3) * X3F_RPITCH, off1 = (Q1 / 3) * X3F_CPITCH + (Q1 % 3) * X3F_RPITCH;
                    const char *p = pb + (h ? off1 : off0);
#pragma unroll
                    for (int q = 0; q < 3; ++q) {
                        vu2 two;
                        __builtin_memcpy(&two, __builtin_assume_aligned(p + q * X3F_PLANE, 4), 8);
                        op[q][2 * e] = two[0];
                        op[q][2 * e + 1] = two[1];
                    }
                }
                const bf16x8 x0 = __builtin_bit_cast(bf16x8, op[0]), x1 = __builtin_bit_cast(bf16x8, op[1]), x2 = __builtin_bit_cast(bf16x8, op[2]);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[2][i], x0, acc, 0, 0, 0); // smallest pieces first
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[0][i], x2, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[1][i], x1, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[1][i], x0, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[0][i], x1, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[0][i], x0, acc, 0, 0, 0);
            }
            const unsigned boff = valid ? (unsigned)(((b * 32 + 4 * h) * X3F_HO + (y0 + yy)) * X3F_WO + xx) * 4u : kNoStore;
            uint32_t gbits = 0u;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = (r & 3) + 8 * (r >> 2);
                float v = __builtin_fmaf(acc[r], a.inv_scale, bias16[r]);
                v = __int_as_float(max(__float_as_int(v), relu_floor));
                vmax = fmaxf(vmax, fabsf(v));
                gbits |= min(__float_as_uint(v) & 0x7fffffffu, 1u) << co;
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rs_out, boff, (unsigned)(co * chw) * 4u, 0);
            }
            if (a.bits_out != nullptr) {
                const uint32_t mine = gbits << (4 * h);
                const auto sw = __builtin_amdgcn_permlane32_swap(mine, mine, false, false);
                const unsigned woff = (h == 0 && valid) ? (unsigned)((b * X3F_HO + (y0 + yy)) * X3F_WO + xx) * 4u : kNoStore;
                __builtin_amdgcn_raw_buffer_store_b32(mine | (uint32_t)sw[1], rs_bits, woff, 0, 0);
            }
        }
    };
    for (int item = first; item < n; item += 2 * NG) {
        compute(item, 0);
        __syncthreads();
        if (item + NG >= n) break;
        compute(item + NG, 1);
        __syncthreads();
    }
    if (a.amax_out != nullptr) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) vmax = fmaxf(vmax, __shfl_xor(vmax, o, 64));
        if (lane == 0) {
            atomicMax(lred, __float_as_uint(vmax));
            if (atomicAdd(lred + 1, 1u) == 3u) atomicMax(a.amax_out + grp, lred[0]);
        }
    }
}

static int g_first_x3 = 1; // (A/B switch for tools: augcl_internal_first_layer_x3)

// AUGCL_OK / negative error after launching, +1 when not covered (f32 NCHW output, 9 x 84 x 84, stride 2)
int conv_first9_x3_try(const float *in, const float *w, const float *bias, float *out, int B, int C, int Hi, int Wi, int stride,
                       float scale, int relu, hipStream_t s, const char *name, const float *w2, const float *bias2, float *amax_out,
                       void *gate_bits_out)
{
    if (!g_first_x3 || C != X3F_C || Hi != X3F_W || Wi != X3F_W || stride != 2 || scale <= 0.0f) return 1;
    if (w2 != nullptr && (B & 1)) return 1;
    if ((size_t)B * 32 * X3F_HO * X3F_WO >= ((size_t)1 << 30) || (reinterpret_cast<uintptr_t>(in) & 15u) != 0) return 1;
    FirstX3Args a{};
    a.in = in; a.w = w; a.bias = bias; a.w2 = w2; a.bias2 = bias2; a.out = out;
    a.B = B; a.dual = w2 != nullptr; a.relu = relu; a.n_items = B * X3F_S;
    a.inv_scale = 1.0f / scale;
    a.amax_out = reinterpret_cast<uint32_t *>(amax_out);
    a.bits_out = relu ? reinterpret_cast<uint32_t *>(gate_bits_out) : nullptr;
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) {
        int v = 0;
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) cus = v;
    }
    int grid = a.n_items < cus ? a.n_items : cus;
    if (a.dual) grid &= ~1;
    if (grid < 1) return 1;
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_conv_first9_x3), hipFuncAttributeMaxDynamicSharedMemorySize, X3F_LDS);
        if (e != hipSuccess) return fail(AUGCL_E_LAUNCH, "%s: cannot set dynamic LDS %d: %s", name, X3F_LDS, hipGetErrorString(e));
        attr_done = true;
    }
    hipLaunchKernelGGL(k_conv_first9_x3, dim3(grid), dim3(512), X3F_LDS, s, a);
    return check_launch(name);
}

} // namespace augcl

extern "C" int augcl_internal_first_layer_x3(int on)
{
    const int was = augcl::g_first_x3;
    augcl::g_first_x3 = on ? 1 : 0;
    return was;
}
