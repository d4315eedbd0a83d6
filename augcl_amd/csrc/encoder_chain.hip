// encoder_chain.hip -- SEVERAL consecutive 32 -> 32 layers of the shared conv encoder (src/algorithms/modules.py:145-165) in
// ONE launch, for passes whose intermediate activations nobody reads again: the two no-grad passes of
// SAC.calculate_target_Q (src/algorithms/sac.py:94-101), the actor step's detached pass (sac.py:119-122) and acting
// (sac.py:73-92).  Arithmetic = encoder_x2.hip's f16 pairs (float32-accurate products, three v_mfma_f32_32x32x16_f16 per
// product group) with the residuals UNSCALED (x s = hi + lo, all three products into one sum: scales are per work item here,
// see below); what changes is where the data lives.
//
// A work item is (image, part): the rows of the chain's LAST layer that one workgroup produces.  Its input -- those rows plus
// two halo rows per layer, all 32 channels -- is staged ONCE into a pixel-major LDS image (slot = f16 hi | lo x 32 channels +
// pad = 144 B, pitch = the input's width) and then every layer of the chain runs IN PLACE on that image: output pixel (y, x)
// of layer j overwrites input pixel (y, x), which is dead as soon as every output pixel up to (y, x) in raster order has been
// computed (a valid 3x3 convolution only looks right and down).  The four compute waves take the layer's 32-pixel tiles in
// raster order, four at a time; ONE workgroup barrier per round separates the round's operand reads from its in-place writes
// (a round only writes pixels in front of every later round's reads, and later tiles' operands are never behind a pixel a
// round writes), one more per layer hands the layer's output over.  Nothing but the chain's input and its last layer's
// output touches HBM: for the encoder's ten 32 -> 32 layers (chains 41 -> 33 in two half-image parts, 33 -> 21 whole
// images) that is 1/5 of the per-layer kernels' bytes, 3 launches instead of 11, and no loader work beside the matrix waves
// except the one staging pass per item.
//
// Scales.  An intermediate that never leaves the workgroup needs no tensor-wide maximum: layer j's output is stored under
// 2^k with k from an A-PRIORI bound of THIS item's values, |y| <= max_co sum_{ci,t} |w[co][ci][t]| max|x| + max|b| (the
// weights' row sums come with the prepared fragments, k_x2_prep), where max|x| is the exact maximum of the item's previous
// layer (every epilogue takes the maximum of what it stores; one LDS atomic per wave and layer).  The bound is rigorous (no
// overflow, no fallback) and does not compound: each layer's looseness (2^3..2^5) only costs that many bits at the BOTTOM
// of f16's range, none of float32's 24.  The chain's INPUT is scaled by the item's own maximum as well (the loader waves take
// it over the registers they stage): no image's accuracy depends on what else is in the batch, which is what allows the
// unscaled residuals (the f16 matrix pipe drops subnormal operands: under a TENSOR-wide scale an image 2^20 dimmer than the
// brightest one came out 1e-4 off; tests/test_referee_gpu.py::test_stack_with_one_outlier_sample...).  A one-layer chain,
// which has no barriers for the loaders to agree on a maximum, uses the tensor's (amax_in).
//
// Loader waves: they fetch the NEXT item's whole input into registers (160 per lane) while the compute waves work, split it
// there, and dump it into LDS between two barriers when the image is free -- the global loads of an item hide behind the
// previous item's ~25 us of matrix work, and the only loader instructions that meet the matrix waves are ~450 per item.
#include "encoder_x2_common.h"

#include <algorithm>
#include <type_traits>

namespace augcl {

constexpr int CH_MAXL = 6;    // layers per launch
constexpr int CH_NU = 5;      // loader rounds per item: 256 lanes x CH_NU units of (4 pixels x 8 channels) = 1280 pixels
constexpr int CH_HDR = 1280;  // LDS in front of the image: 4 x 128 B bias seeds (one copy per compute wave), the maxima, 4 dump slots

struct ChainArgs {
    const float *in;              // (B,32,Hi,Wi) f32 NCHW: input of the chain's first layer (post-ReLU activations)
    float *out;                   // (B,32,Hi-2nl,Wi-2nl) f32 NCHW: output of its last layer
    const float *bias[2][CH_MAXL]; // [weight set][layer] (32) or null
    const vu4 *wfrag[2][CH_MAXL];  // prepared by k_x2_prep (mode 2: forward fragments, unscaled residuals)
    int nl, B, Hi, Wi, parts, rows_out, relu_last, dual, n_items;
    uint32_t magic_w[CH_MAXL];    // floor(2^32 / width of layer j's output) + 1
    uint32_t magic_parts;         // floor(2^32 / parts) + 1 (parts == 1 is tested for)
    const uint32_t *amax_in;      // max |in| (float bits), used by one-layer chains only; dual: [0] images [0, B/2), [1] images [B/2, B)
    uint32_t *amax_out;           // max |out| is atomically max-ed into it (zeroed by the caller), or null; dual: [0], [1]
    // KEEP (the gradient-carrying forward, src/algorithms/sac.py:103-116 -> modules.py:145-165 with autograd on): every layer's
    // output is ALSO stored for the backward pass -- float32 NCHW, each row once (a part stores the rows it owns) -- together
    // with its ReLU gate bits (one word per pixel, bit c = channel c > 0: encoder_x2.hip's format) and its tensor-wide maximum
    float *acts[CH_MAXL];         // [j], j < nl - 1: (B,32,Hj,Wj) output of layer j (the last layer's is `out`)
    uint32_t *bits[CH_MAXL];      // [j], j < nl: (B,Hj,Wj) gate words of layer j's output, or null
    uint32_t *amax_acts[CH_MAXL]; // [j], j < nl - 1: max |output of layer j| (float bits) is atomically max-ed into it
};

#ifdef AUGCL_TUNING
// tuning builds: compute wave 0 of workgroup 0 adds up where its cycles go (s_memtime): [0] matrix loops, [1] waiting at the
// round barriers, [2] epilogues, [3] layer hand-over (barrier + next layer's set-up), [4] item switch (F .. A), [5] rounds,
// [6] whole kernel, [7] 100 MHz ticks of the whole kernel
__device__ unsigned long long g_chain_clk[8];
#endif

struct ChainItem {
    int b, y0, ny, grp;
};

__device__ __forceinline__ ChainItem chain_item(const ChainArgs &a, int item)
{
    ChainItem it;
    it.b = a.parts == 1 ? item : (int)__umulhi((uint32_t)item, a.magic_parts);
    const int p = item - it.b * a.parts;
    it.y0 = p * a.rows_out;
    it.ny = min(a.rows_out, a.Hi - 2 * a.nl - it.y0);
    it.grp = (a.dual && it.b >= (a.B >> 1)) ? 1 : 0;
    return it;
}

// rounds of four 32-pixel tiles that layer j of an item with ny final rows takes
__device__ __forceinline__ int chain_rounds(const ChainArgs &a, int j, int ny)
{
    const int npix = (ny + 2 * (a.nl - 1 - j)) * (a.Wi - 2 * (j + 1));
    return (((npix + 31) >> 5) + 3) >> 2;
}

template <bool KEEP>
__global__ __launch_bounds__(512, 2) void k_conv32_chain(ChainArgs a)
{
    extern __shared__ __attribute__((aligned(16))) char ldsc[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool loader = wave >= 4;
    const int cw = wave & 3;
    const int l31 = lane & 31, h = lane >> 5;
    const int Wi = a.Wi, nl = a.nl;
    const int Hf = a.Hi - 2 * nl, Wf = Wi - 2 * nl;
    uint32_t *lred = reinterpret_cast<uint32_t *>(ldsc + 512); // [j] max of layer j's stored (scaled) output of this item; [8..9]: kernel end
    char *img = ldsc + CH_HDR;
    const int G = (int)gridDim.x, n = a.n_items, first = xcd_slot((int)blockIdx.x, G);
    const int rowb = Wi * X2_SLOT;

    if (loader) {
        const int lt = cw * 64 + lane;
        const int cg = lt & 3, q0 = lt >> 2; // this lane's channel group (8 channels) and first pixel quad
        const unsigned cs4 = (unsigned)(a.Hi * Wi) * 4u; // bytes per channel plane
        const unsigned lane_c = (unsigned)cg * 8u * cs4;
        const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a.in), 0, -1, 0x00020000);
        vu4 d[CH_NU][8]; // the next item: raw f32 (8 channels x 4 pixels per unit), then in place its f16 (hi | lo) pairs
        int npx = 0;     // pixels of the staged item
        auto issue = [&](int item) {
            const ChainItem it = chain_item(a, item);
            npx = (it.ny + 2 * nl) * Wi; // rows [y0, y0 + ny + 2 nl) of every channel plane: one contiguous run each
            const unsigned base = (unsigned)((it.b * 32 * a.Hi + it.y0) * Wi) * 4u;
#pragma unroll
            for (int u = 0; u < CH_NU; ++u) {
                if (256 * u < npx) { // wave-uniform: this round holds pixels of the run
                    const int p0 = min(4 * (q0 + 64 * u), npx - 4); // units past the run repeat its last quad (idempotent)
#pragma unroll
                    for (int c = 0; c < 8; ++c)
                        d[u][c] = __builtin_bit_cast(vu4, __builtin_amdgcn_raw_buffer_load_b128(rs_in, lane_c + 4u * (unsigned)p0,
                                                                                                 base + (unsigned)c * cs4, 0));
                }
            }
        };
        uint32_t amax_bits = 0u; // max |x| of the staged item (float bits): its scale, handed to the compute waves with the image
        auto item_max = [&]() {  // this wave's share of it, into lred[12] (zeroed at the previous dump)
            float m = 0.0f;
#pragma unroll
            for (int u = 0; u < CH_NU; ++u) {
                if (256 * u < npx) {
#pragma unroll
                    for (int c = 0; c < 8; c += 2)
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            m = fmaxf(m, fmaxf(fabsf(__uint_as_float(d[u][c][j])), fabsf(__uint_as_float(d[u][c + 1][j]))));
                }
            }
            m = wave_max(m);
            if (lane == 0) atomicMax(lred + 12, __float_as_uint(m));
        };
        auto split = [&]() { // in place: d[u][2e], d[u][2e+1] (channels 2e, 2e+1 of 4 pixels) -> their hi dwords in d[u][2e], (unscaled) lo in d[u][2e+1]
            const float s = pow2f(scale_exp(amax_bits));
#pragma unroll
            for (int u = 0; u < CH_NU; ++u) {
                if (256 * u < npx) {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            uint32_t th, tl;
                            split2u(__uint_as_float(d[u][2 * e][j]) * s, __uint_as_float(d[u][2 * e + 1][j]) * s, th, tl);
                            d[u][2 * e][j] = th; d[u][2 * e + 1][j] = tl;
                        }
                }
            }
        };
        auto dump = [&]() { // registers -> the LDS image (slot = pixel of the run: the image's pitch is the input's width)
#pragma unroll
            for (int u = 0; u < CH_NU; ++u) {
                if (256 * u < npx) {
                    const int p0 = min(4 * (q0 + 64 * u), npx - 4);
                    char *o = img + p0 * X2_SLOT + cg * 16;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        *reinterpret_cast<vu4 *>(o + j * X2_SLOT) = vu4{d[u][0][j], d[u][2][j], d[u][4][j], d[u][6][j]};
                        *reinterpret_cast<vu4 *>(o + j * X2_SLOT + 64) = vu4{d[u][1][j], d[u][3][j], d[u][5][j], d[u][7][j]};
                    }
                }
            }
            if (lt < 16) lred[lt] = lt == 13 ? amax_bits : 0u; // the item's maxima start at zero (every read of the previous item's is
        };                                                      // behind us: barrier F); [13] = the image's own maximum
        // The staged item's scale is ITS OWN maximum (not the tensor's: an outlier image elsewhere in the batch must not push this
        // one's values down f16's range -- the residuals are unscaled here): every loader wave adds its share to an LDS word, a
        // workgroup barrier later all four read the total.  The first item pays two barriers of its own (P0, P1: the compute waves
        // execute them too); later items use two of the barriers the compute waves go through anyway.  A one-layer chain has no
        // such barriers: its items are scaled by the tensor's maximum (amax_in), like encoder_x2.hip's.
        const bool own_max = nl >= 2;
        if (lt < 16) lred[lt] = 0u;
        __syncthreads(); // P0
        if (first < n) {
            issue(first);
            if (own_max) item_max();
        }
        __syncthreads(); // P1
        if (first < n) {
            amax_bits = own_max ? (uint32_t)__builtin_amdgcn_readfirstlane((int)lred[12])
                                : (uint32_t)__builtin_amdgcn_readfirstlane((int)a.amax_in[chain_item(a, first).grp]);
            split();
            dump();
        }
        for (int item = first; item < n; item += G) {
            __syncthreads(); // A: the image is ready
            const int nxt = item + G;
            if (nxt < n) issue(nxt);
            const int ny = chain_item(a, item).ny;
            int nb = 0; // the compute waves' barriers inside this item: one per round and one per layer, last layer excepted
            for (int j = 0; j + 1 < nl; ++j) nb += chain_rounds(a, j, ny) + 1;
            const bool have = nxt < n;
            const int at = nb >= 2 ? nb / 2 : 0; // half way: the loads issued at the item's start have long landed
            for (int i = 0; i < nb; ++i) {
                if (have && own_max && i + 1 == at) item_max(); // (nb >= 2 whenever nl >= 2)
                if (have && own_max && i == at) {
                    amax_bits = (uint32_t)__builtin_amdgcn_readfirstlane((int)lred[12]);
                    split();
                }
                __syncthreads();
            }
            if (have && !own_max) {
                amax_bits = (uint32_t)__builtin_amdgcn_readfirstlane((int)a.amax_in[chain_item(a, nxt).grp]);
                split();
            }
            __syncthreads(); // F: every read of the image is done
            if (nxt < n) dump();
        }
        return;
    }

    // ---- compute waves.  Operand i = (tap t = i / 2, channel half s = i & 1); A fragment: W[row l31][k = 16 s + 8 h + j][t], hi and
    // (unscaled) lo parts in registers for one layer; the next layer's are requested as soon as the layer's last matrix instruction is out.
    // All three products of a group (wl xh, wh xl, wh xh) go into ONE sum -- the residuals are unscaled here, unlike
    // encoder_x2.hip's, because a chain's scales are per work item (no outlier image can push another one's values down f16's
    // range) -- which leaves room to keep the PREVIOUS tile's raw results: a tile's epilogue (un-scale, ReLU, maximum, split,
    // in-place store) is spread over the gaps of the NEXT tile's 54 matrix instructions.
    f16x8 wh[18], wl[18];
    auto load_weights = [&](const vu4 *wf) {
#pragma unroll
        for (int i = 0; i < 18; ++i) {
            wh[i] = __builtin_bit_cast(f16x8, wf[i * 64 + lane]);
            wl[i] = __builtin_bit_cast(f16x8, wf[(18 + i) * 64 + lane]);
        }
    };
    uint32_t nx_kw = 0u, nx_l1 = 0u; // the coming layer's weight exponent / largest absolute row sum (float bits) ...
    float nx_bv = 0.0f;              // ... and bias (lane c < 32: channel c)
    auto load_consts = [&](const vu4 *wf, const float *bsrc) {
        const uint32_t *tail = reinterpret_cast<const uint32_t *>(wf + X2_FRAG_VU4);
        nx_kw = tail[0];
        nx_l1 = tail[1];
        nx_bv = (bsrc && lane < 32) ? bsrc[lane] : 0.0f;
    };
    float *lbias = reinterpret_cast<float *>(ldsc) + 32 * cw; // this wave's copy of the layer's bias seeds (accumulator scale)
    char *dump_slot = ldsc + 576 + 160 * cw;                  // 144 bytes nobody reads: where a lane without a pixel "stores"
    const __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc(a.out, 0, (int)((unsigned)(a.B * 32 * Hf * Wf) * 4u), 0x00020000);
    constexpr unsigned kNoStore = 0xfffffff0u;
    const int chw = Hf * Wf;
    float vmax_out[2] = {0.0f, 0.0f}; // max |final output| per weight set (amax_out)
#ifdef AUGCL_TUNING
    unsigned long long c_mma = 0, c_wait = 0, c_epi = 0, c_layer = 0, c_item = 0, c_rounds = 0;
    const unsigned long long t_kernel = __builtin_amdgcn_s_memtime(), r_kernel = __builtin_amdgcn_s_memrealtime();
    unsigned long long t_mark = t_kernel;
#endif
    if (first < n) {
        const int g0 = chain_item(a, first).grp;
        load_weights(a.wfrag[g0][0]);
        load_consts(a.wfrag[g0][0], a.bias[g0][0]);
    }
    uint32_t *lkeep = reinterpret_cast<uint32_t *>(ldsc + 1216); // KEEP: [j] = max |output of layer j| over this workgroup's items (true units)
    if (KEEP && tid < CH_MAXL) lkeep[tid] = 0u;
    __syncthreads(); // P0
    __syncthreads(); // P1 (the loaders agree on the first item's maximum between the two)
    typedef uint32_t vu2 __attribute__((ext_vector_type(2)));

    for (int item = first; item < n; item += G) {
        const ChainItem it = chain_item(a, item);
        __syncthreads(); // A
#ifdef AUGCL_TUNING
        { const unsigned long long t = __builtin_amdgcn_s_memtime(); c_item += t - t_mark; t_mark = t; }
#endif
        const uint32_t ain = (uint32_t)__builtin_amdgcn_readfirstlane((int)lred[13]); // left by the loaders with the image
        int k_in = scale_exp(ain);
        float amax_true = __uint_as_float(ain); // max |x| of this item (one-layer chains: of the tensor)
        // one layer of the chain; `last` (compile time: the epilogue slices sit between matrix instructions, where a scalar branch
        // per slice would cost more than the slice) = the chain's last layer, which stores float32 to `out` instead of in place
        auto layer = [&](int j, auto last_tag) {
            constexpr bool last = decltype(last_tag)::value;
            const int Wout = Wi - 2 * (j + 1), nrow = it.ny + 2 * (nl - 1 - j);
            const int npix = nrow * Wout, ntiles = (npix + 31) >> 5, rounds = (ntiles + 3) >> 2;
            const uint32_t magic = a.magic_w[j];
            // the layer's constants (weights' exponent, their largest row sum, the bias) were requested together with its weight
            // fragments, a layer ago: nothing global is waited for between two layers
            const int k_w = __builtin_amdgcn_readfirstlane((int)nx_kw);
            const float l1 = __uint_as_float(__builtin_amdgcn_readfirstlane((int)nx_l1));
            // bias seeds and their maximum (this wave's own LDS copy: no barrier between the layers' bias traffic)
            const float bv = nx_bv;
            const float bmax = wave_max(fabsf(bv));
            if (lane < 32) lbias[lane] = bv * pow2f(k_in) * pow2f(k_w);
            // the scale of what this layer stores: 0 for the chain's output (float32), else from the a-priori bound of the item
            int k_out = 0;
            if (!last) k_out = scale_exp(__float_as_uint((l1 * amax_true + bmax) * 1.001f));
            const int relu_floor = __builtin_amdgcn_readfirstlane((!last || a.relu_last) ? 0 : (int)0x80000000);
            float vmax = 0.0f;
            const int grp_next = !last ? it.grp : (item + G < n ? chain_item(a, item + G).grp : it.grp);
            const int j_next = !last ? j + 1 : (item + G < n ? 0 : j);
            const vu4 *wf_next = a.wfrag[grp_next][j_next];
            const float *b_next = a.bias[grp_next][j_next];
            // KEEP: where this layer's output goes for the backward pass.  Rows [y0, y0 + ny) of every layer belong to this part,
            // the rows behind them to the next one (the last part: all of them); a lane that owns no pixel names an offset beyond
            // the buffer and its store is dropped.  The chain's last layer already stores float32 (`out`): only its gate words are new.
            const int Hj = a.Hi - 2 * (j + 1), chwj = Hj * Wout;
            float *keep = (KEEP && !last) ? a.acts[j] : nullptr;
            uint32_t *kbits = KEEP ? a.bits[j] : nullptr;
            const __amdgpu_buffer_rsrc_t rs_keep = __builtin_amdgcn_make_buffer_rsrc(
                keep ? keep : a.out, 0, keep ? (int)((unsigned)(a.B * 32 * chwj) * 4u) : 0, 0x00020000);
            const __amdgpu_buffer_rsrc_t rs_bits = __builtin_amdgcn_make_buffer_rsrc(
                kbits ? kbits : reinterpret_cast<uint32_t *>(a.out), 0, kbits ? (int)((unsigned)(a.B * chwj) * 4u) : 0, 0x00020000);
            const bool last_part = it.y0 + it.ny >= Hf;
            unsigned pkoff = kNoStore, pgoff = kNoStore;
            uint32_t gb = 0u;
            float uv = 0.0f;
            // byte offset of accumulator row q's channel plane.  KEEP: computed where it is used (one s_mul per store) -- hoisted out
            // of the loops, the 16 offsets per plane size sat in scalar registers the kernel does not have (spills into vector lanes)
            auto plane_off = [&](int q, int plane) -> unsigned {
                const int rowc = (q & 3) + 8 * (q >> 2);
                if (KEEP) {
                    unsigned c4 = (unsigned)plane * 4u;
                    asm volatile("" : "+s"(c4));
                    return (unsigned)rowc * c4;
                }
                return (unsigned)(rowc * plane) * 4u;
            };

            auto locate = [&](int tl, bool &valid_, int &off_, int &yy_, int &xx_) {
                int m_ = tl * 32 + l31;
                valid_ = m_ < npix;
                if (!valid_) m_ = npix - 1; // read a real pixel, discard the result
                yy_ = (int)__umulhi((uint32_t)m_, magic);
                xx_ = m_ - yy_ * Wout;
                off_ = (yy_ * Wi + xx_) * X2_SLOT + h * 16;
            };
            struct Ops {
                vu4 xh, xl;
            };
            auto rd1 = [&](int off_, int i, Ops &o, int part) { // one of the operand's two reads: row base + immediate
                const int t = i >> 1;
                const char *p = img + off_ + (t / 3) * rowb + ((t % 3) * X2_SLOT + (i & 1) * 32);
                if (part == 0) o.xh = *reinterpret_cast<const vu4 *>(p);
                else o.xl = *reinterpret_cast<const vu4 *>(p + 64);
            };
            // the previous tile of this wave: raw accumulators + where its results go (an LDS slot, or a byte offset of `out`)
            f32x16 prev;
#pragma unroll
            for (int q = 0; q < 16; ++q) prev[q] = 0.0f;
            char *pdst = dump_slot;
            unsigned pboff = kNoStore;
            bool pending = false;
            // The previous tile's epilogue as a SOFTWARE PIPELINE over the 54 gaps between the current tile's matrix instructions
            // (slot g = 3 i + {0, 1, 2} behind the three instructions of operand group i); every slot holds instructions whose inputs
            // are at least one slot old, so none waits for the one just before it.  (Measured: the loop costs ~2350 cycles with the
            // epilogue in its gaps against ~1930 bare + 1070 for the epilogue behind it; dependent or independent slices, one or two
            // accumulation chains made no difference to the 2350 -- DESIGN.md 4d.)
            //   3q      s_q = prev_q 2^kk                      (v_ldexp_f32: the un-scale and the next layer's scale in one, any kk)
            //   3q + 1  t_q = max(s_q, 0 | -inf)               (ReLU as an integer maximum)           [chain's last layer: 3q + 2 store]
            //   3q + 2  (q odd)  hi = f16x2(t_{q-1}, t_q),  vmax = max(vmax, t_{q-1}, t_q)
            //   3q + 3  f = f32(hi)   3q + 4  r = t - f   3q + 5  lo = f16x2(r)   3q + 6 (q % 4 == 3)  the two 8-byte LDS stores
            float sv[16], tvv[16];
            uint32_t hi2[8], lo2[8];
            float f0[8], f1[8];
            const int kk = k_out - k_in - k_w; // |kk| <= 194: ldexp takes it in one step, two float factors could not
            auto epi_slot = [&](int g) {
                const int q = g / 3, ph = g % 3;
                if (ph == 0 && q < 16) sv[q] = __builtin_amdgcn_ldexpf(prev[q], kk);
                if (ph == 1 && q < 16) tvv[q] = __int_as_float(max(__float_as_int(sv[q]), relu_floor));
                if (KEEP) {
                    // gate bit of accumulator row q (channel (q & 3) + 8 (q >> 2) + 4 h): post-ReLU values are >= +0, positive <=>
                    // bit pattern >= 1 (a last layer without ReLU has no gate words: its store is dropped)
                    if (ph == 2 && q < 16) {
                        if (q == 0) gb = 0u;
                        gb = (min(__float_as_uint(tvv[q]), 1u) << ((q & 3) + 8 * (q >> 2))) | gb;
                    }
                    if (g == 49) { // the two half-waves hold channels {0-3, 8-11, ..} and {4-7, 12-15, ..} of the same 32 pixels
                        const uint32_t mine = gb << (4 * h);
                        const auto sw = __builtin_amdgcn_permlane32_swap(mine, mine, false, false);
                        gb = mine | (uint32_t)sw[1];
                    }
                    if (g == 51) __builtin_amdgcn_raw_buffer_store_b32(gb, rs_bits, pgoff, 0, 0);
                    if (!last) { // the float32 copy: the stored value without the next layer's scale (a power of two: exact)
                        if (ph == 0 && q >= 1 && q <= 16)
                            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, uv), rs_keep, pkoff, plane_off(q - 1, chwj), 0);
                        if (ph == 2 && q < 16) uv = __builtin_amdgcn_ldexpf(tvv[q], -k_out);
                    }
                }
                if (last) {
                    if (ph == 2 && q < 16) {
                        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, tvv[q]), rs_out, pboff, plane_off(q, chw), 0);
                        if (q & 1) vmax = fmaxf(vmax, fmaxf(fabsf(tvv[q - 1]), fabsf(tvv[q])));
                    }
                    return;
                }
                // pair p = (2p, 2p + 1) completes at slot 3 (2p + 1) + 1
                if (ph == 2 && (q & 1) && q < 16) {
                    const int p_ = q >> 1;
                    hi2[p_] = __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2{tvv[q - 1], tvv[q]}, f16x2));
                    vmax = fmaxf(vmax, fmaxf(tvv[q - 1], tvv[q])); // post-ReLU: non-negative
                }
                if (ph == 0 && q >= 2 && !(q & 1) && q <= 16) { // 3 (q-1) + 3
                    const int p_ = (q - 2) >> 1;
                    const f16x2 hv = __builtin_bit_cast(f16x2, hi2[p_]);
                    f0[p_] = (float)hv[0];
                    f1[p_] = (float)hv[1];
                }
                if (ph == 1 && q >= 2 && !(q & 1) && q <= 16) {
                    const int p_ = (q - 2) >> 1;
                    f0[p_] = tvv[2 * p_] - f0[p_];
                    f1[p_] = tvv[2 * p_ + 1] - f1[p_];
                }
                if (ph == 2 && q >= 2 && !(q & 1) && q <= 16) {
                    const int p_ = (q - 2) >> 1;
                    lo2[p_] = __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2{f0[p_], f1[p_]}, f16x2));
                }
                if (ph == 0 && q >= 5 && (q & 3) == 1 && q <= 17) { // group g4 = (q - 5) / 4 of four channels: pairs 2 g4, 2 g4 + 1 are split
                    const int g4 = (q - 5) >> 2;
                    *reinterpret_cast<vu2 *>(pdst + 16 * g4) = vu2{hi2[2 * g4], hi2[2 * g4 + 1]};
                    *reinterpret_cast<vu2 *>(pdst + 16 * g4 + 64) = vu2{lo2[2 * g4], lo2[2 * g4 + 1]};
                }
            };
            auto epilogue_now = [&]() { // the same, not hidden (a layer's last tile; a wave that has run out of tiles)
#pragma unroll
                for (int g = 0; g < 54; ++g) epi_slot(g);
            };
            bool valid = false;
            int off = 0, yy = 0, xx = 0;
            Ops o[3];
            if (cw < ntiles) {
                locate(cw, valid, off, yy, xx);
                rd1(off, 0, o[0], 0); rd1(off, 0, o[0], 1);
                rd1(off, 1, o[1], 0); rd1(off, 1, o[1], 1);
            }
            // One round = the wave's tile 4 r + cw: operand reads + matrix instructions (+ the previous tile's epilogue in their gaps),
            // then the round's barrier.  The previous tile's in-place stores are behind the barrier of ITS round, as they must be, and
            // in front of every later read (a tile reads at and behind its own pixels).  The layer's LAST round is peeled (FINAL):
            // behind its matrix instructions the next layer's (or the next item's first layer's) weight fragments are requested
            // into the registers the layer just finished with -- unconditionally, at one place in the program (requested inside
            // the loop under a condition they cost 80 spilled registers); they land during the last epilogue and the layer barrier.
            auto round = [&](int r, auto final_round) {
                constexpr bool FINAL = decltype(final_round)::value;
                const int tl = 4 * r + cw;
                const bool active = tl < ntiles; // wave-uniform
#ifdef AUGCL_TUNING
                { const unsigned long long t = __builtin_amdgcn_s_memtime(); c_layer += t - t_mark; t_mark = t; ++c_rounds; }
#endif
                if (active) {
                    bool valid2;
                    int off2, yy2, xx2;
                    locate(tl + 4 < ntiles ? tl + 4 : tl, valid2, off2, yy2, xx2); // no next tile: re-read this one
                    f32x16 acc, acc2;
#pragma unroll
                    for (int g4 = 0; g4 < 4; ++g4) { // accumulator seed = bias of rows 8 g4 + 4 h + (0..3)
                        const vf4 b4 = *reinterpret_cast<const vf4 *>(lbias + 8 * g4 + 4 * h);
                        acc[4 * g4] = b4[0]; acc[4 * g4 + 1] = b4[1]; acc[4 * g4 + 2] = b4[2]; acc[4 * g4 + 3] = b4[3];
                    }
                    // TWO accumulators taken in turn (their sum is the tile): one chain of 54 dependent matrix instructions ran at ~44
                    // cycles per instruction, two interleaved chains issue at the pipe's 32-36
#pragma unroll
                    for (int q = 0; q < 16; ++q) acc2[q] = 0.0f;
#define CH_MM(A_, B_) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(A_, __builtin_bit_cast(f16x8, B_), acc, 0, 0, 0)
// (KEEP: ONE accumulator -- measured flat in the no-grad kernel, 2382 against 2348 cycles per tile loop -- which frees the sixteen
// registers the kept copy's offsets and gate word need; with two, the kernel spilled into its matrix loop)
#define CH_MM2(A_, B_)                                                                                              \
    do {                                                                                                            \
        if (KEEP) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(A_, __builtin_bit_cast(f16x8, B_), acc, 0, 0, 0);    \
        else acc2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(A_, __builtin_bit_cast(f16x8, B_), acc2, 0, 0, 0);       \
    } while (0)
#pragma unroll
                    for (int i = 0; i < 18; ++i) {
                        Ops &cur = o[i % 3];
                        Ops &nn = o[(i + 2) % 3];
                        const int noff = i < 16 ? off : off2, ni = i < 16 ? i + 2 : i - 16; // the operand two groups on (across the tile boundary)
                        // smallest products first: wl xh, wh xl (cross), wh xh (main); between them one LDS read and a slice of the
                        // previous tile's epilogue
                        if (i & 1) CH_MM2(wl[i], cur.xh); else CH_MM(wl[i], cur.xh);
                        __builtin_amdgcn_sched_barrier(0);
                        rd1(noff, ni, nn, 0);
                        epi_slot(3 * i);
                        __builtin_amdgcn_sched_barrier(0);
                        if (i & 1) CH_MM(wh[i], cur.xl); else CH_MM2(wh[i], cur.xl);
                        __builtin_amdgcn_sched_barrier(0);
                        rd1(noff, ni, nn, 1);
                        epi_slot(3 * i + 1);
                        __builtin_amdgcn_sched_barrier(0);
                        if (i & 1) CH_MM2(wh[i], cur.xh); else CH_MM(wh[i], cur.xh);
                        __builtin_amdgcn_sched_barrier(0);
                        epi_slot(3 * i + 2);
                        __builtin_amdgcn_sched_barrier(0);
                    }
#undef CH_MM
#undef CH_MM2
                    if (KEEP) prev = acc;
                    else prev = acc + acc2;
                    pdst = valid ? img + (yy * Wi + xx) * X2_SLOT + 8 * h : dump_slot;
                    pboff = valid ? (unsigned)((it.b * 32 * Hf + (it.y0 + yy)) * Wf + xx + 4 * h * chw) * 4u : kNoStore;
                    if (KEEP) {
                        const bool mine_ = valid && (last || yy < it.ny || last_part);
                        pkoff = mine_ ? (unsigned)((it.b * 32 * Hj + (it.y0 + yy)) * Wout + xx + 4 * h * chwj) * 4u : kNoStore;
                        pgoff = (mine_ && h == 0) ? (unsigned)((it.b * Hj + (it.y0 + yy)) * Wout + xx) * 4u : kNoStore;
                    }
                    pending = true;
                    valid = valid2; off = off2; yy = yy2; xx = xx2;
                } else if (pending) { // out of tiles: finish the last one (its round's barrier is behind us)
                    epilogue_now();
                    pending = false;
                }
                if (FINAL) {
#ifndef AUGCL_ABL_NO_WEIGHT_FETCH // (tuning builds: AUGCL_ABL=-DAUGCL_ABL_NO_WEIGHT_FETCH tools/build_tuning.sh -- timing only, wrong results)
                    load_weights(wf_next);
#endif
                    load_consts(wf_next, b_next);
                }
#ifdef AUGCL_TUNING
                { const unsigned long long t = __builtin_amdgcn_s_memtime(); c_mma += t - t_mark; t_mark = t; }
#endif
                if (!last) __syncthreads(); // R: the round's operand reads are done
#ifdef AUGCL_TUNING
                { const unsigned long long t = __builtin_amdgcn_s_memtime(); c_wait += t - t_mark; t_mark = t; }
#endif
            };
            for (int r = 0; r + 1 < rounds; ++r) round(r, std::false_type{});
            round(rounds - 1, std::true_type{});
            if (pending) epilogue_now(); // the wave's last tile of the layer (behind its round's barrier)
#ifdef AUGCL_TUNING
            { const unsigned long long t = __builtin_amdgcn_s_memtime(); c_epi += t - t_mark; t_mark = t; }
#endif
            if (!last) {
                // the item's maximum of this layer's output (scaled units): one LDS atomic per wave, read by all after the barrier
                vmax = wave_max(vmax);
                if (lane == 0) atomicMax(lred + j, __float_as_uint(vmax));
                __syncthreads(); // L: the layer's output is in place
                amax_true = __uint_as_float(__builtin_amdgcn_readfirstlane((int)lred[j])) * pow2f(-k_out);
                k_in = k_out;
                if (KEEP && cw == 0 && lane == 0) atomicMax(lkeep + j, __float_as_uint(amax_true)); // (halo rows are other parts' rows: same tensor)
            } else {
                vmax_out[it.grp] = fmaxf(vmax_out[it.grp], vmax);
            }
        };
        for (int j = 0; j + 1 < nl; ++j) layer(j, std::false_type{});
        layer(nl - 1, std::true_type{});
        __syncthreads(); // F
    }
#ifdef AUGCL_TUNING
    if (blockIdx.x == 0 && wave == 0 && lane == 0) {
        g_chain_clk[0] = c_mma; g_chain_clk[1] = c_wait; g_chain_clk[2] = c_epi; g_chain_clk[3] = c_layer; g_chain_clk[4] = c_item;
        g_chain_clk[5] = c_rounds;
        g_chain_clk[6] = __builtin_amdgcn_s_memtime() - t_kernel;
        g_chain_clk[7] = __builtin_amdgcn_s_memrealtime() - r_kernel;
    }
#endif
    if (a.amax_out != nullptr) {
        // ONE global atomic per workgroup and weight set: every compute wave adds its maximum in LDS and takes a ticket; the last
        // one sends them on.  (The loaders are gone; lred[8..11] were zeroed by their last dump... they are not: zero them here.)
        const float m0 = wave_max(vmax_out[0]), m1 = wave_max(vmax_out[1]);
        if (lane == 0) {
            atomicMax(lred + 8, __float_as_uint(m0));
            atomicMax(lred + 9, __float_as_uint(m1));
            if (atomicAdd(lred + 10, 1u) == 3u) {
                atomicMax(a.amax_out, lred[8]);
                if (a.dual) atomicMax(a.amax_out + 1, lred[9]);
            }
        }
    }
    if (KEEP) { // the kept layers' maxima: wave 0 collected them in LDS (its own earlier instructions), one global atomic per layer
        if (cw == 0 && lane == 0)
            for (int j = 0; j + 1 < nl; ++j)
                if (a.amax_acts[j] != nullptr) atomicMax(a.amax_acts[j], lkeep[j]);
    }
}


// =========================================================================================================================
// The same chain for the bf16 PERFORMANCE MODE (BASELINE config 2 "bf16"; csrc/encoder_bf16.hip): activations bf16 (f32's
// exponent range: no scales, no maxima), one v_mfma_f32_32x32x16_bf16 per (tap, channel half) = 18 per tile, f32 accumulation,
// every layer's output rounded to bf16 (RNE) exactly as the per-layer kernels store it.  A pixel slot is 64 + 16 bytes, so
// a WHOLE 41 x 41 image (1681 slots = 131 KiB) is one work item: the encoder's ten 32 -> 32 layers run as ONE launch with no
// halo, no recomputation, and nothing but the first layer's output and the last layer's written or read in HBM -- where
// the per-layer bf16 kernels are HBM-paced (27 us of a launch for 6 us of matrix work).  The weights of a layer are 72
// registers: the NEXT layer's are requested when a layer starts and are there when it ends.
constexpr int CB_MAXL = 12;   // layers per launch
constexpr int CB_NU = 28;     // loader rounds per item: 256 lanes x CB_NU chunks of 16 B = 1792 pixels
constexpr int CB_SLOT = 80;   // LDS bytes per pixel: 32 bf16 + 16 B pad (20 p mod 64 dwords: conflict-free ds_read_b128 over 16 pixels)
constexpr int CB_HDR = 1024;  // 4 x 128 B bias seeds (one copy per compute wave), 4 dump slots
constexpr int CB_FRAG_BYTES = 18 * 64 * 16; // prepared bf16 fragments of one layer

typedef __bf16 bf16x8c __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2c __attribute__((ext_vector_type(2)));
typedef short s16x2c __attribute__((ext_vector_type(2)));

struct ChainBfArgs {
    const vu4 *in;                 // bf16 NHWC (B,Hi,Wi,32) as 16-byte chunks (4 per pixel)
    void *out;                     // f32 NCHW (B,32,Hf,Wf) when out_f32, else bf16 NHWC (B,Hf,Wf,32)
    const float *bias[2][CB_MAXL]; // [weight set][layer] (32) or null
    const vu4 *wfrag[2][CB_MAXL];  // k_bf16_prep: [18 operands][64 lanes] x 16 B
    void *acts[CB_MAXL];           // gradient-carrying forward: acts[j] != null receives layer j's output (bf16 NHWC, every row
                                   // once: a part stores the rows it owns) for the backward pass; j < nl - 1
    int nl, B, Hi, Wi, parts, rows_out, relu_last, dual, n_items, out_f32;
    uint32_t magic_w[CB_MAXL];     // floor(2^32 / width of layer j's output) + 1
    uint32_t magic_parts;
};

__device__ __forceinline__ ChainItem chain_bf_item(const ChainBfArgs &a, int item)
{
    ChainItem it;
    it.b = a.parts == 1 ? item : (int)__umulhi((uint32_t)item, a.magic_parts);
    const int p = item - it.b * a.parts;
    it.y0 = p * a.rows_out;
    it.ny = min(a.rows_out, a.Hi - 2 * a.nl - it.y0);
    it.grp = (a.dual && it.b >= (a.B >> 1)) ? 1 : 0;
    return it;
}

// bf16 fragments of n 32 -> 32 layers: lane l of operand i = (tap t = i / 2, channel half s = i & 1) holds W[co = l & 31][ci = 16 s + 8 (l >> 5) + j][t]
struct PrepBfArgs {
    const float *w[32];
    char *out;     // [n] blocks of `stride` bytes
    int stride;
    int mode;      // 0: forward W[co][ci][t];  1: backward-data W'[ci][co][t] = W[co][ci][8 - t]
};

__global__ __launch_bounds__(256) void k_bf16_prep(PrepBfArgs a)
{
    const int lane = threadIdx.x & 63, l31 = lane & 31, h = lane >> 5;
    const int i = 4 * blockIdx.x + (threadIdx.x >> 6), t = i >> 1, s = i & 1;
    if (i >= 18) return;
    const float *w = a.w[blockIdx.y];
    bf16x8c f;
#pragma unroll
    for (int j = 0; j < 8; ++j)
        f[j] = a.mode == 0 ? (__bf16)w[l31 * 288 + (16 * s + 8 * h + j) * 9 + t] : (__bf16)w[(16 * s + 8 * h + j) * 288 + l31 * 9 + (8 - t)];
    reinterpret_cast<vu4 *>(a.out + (size_t)blockIdx.y * a.stride)[i * 64 + lane] = __builtin_bit_cast(vu4, f);
}

template <bool OUT_F32>
__global__ __launch_bounds__(512, 2) void k_conv32_chain_bf16(ChainBfArgs a)
{
    extern __shared__ __attribute__((aligned(16))) char ldsb[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool loader = wave >= 4;
    const int cw = wave & 3;
    const int l31 = lane & 31, h = lane >> 5;
    const int Wi = a.Wi, nl = a.nl;
    const int Hf = a.Hi - 2 * nl, Wf = Wi - 2 * nl;
    char *img = ldsb + CB_HDR;
    const int G = (int)gridDim.x, n = a.n_items, first = xcd_slot((int)blockIdx.x, G);
    const int rowb = Wi * CB_SLOT;
    auto rounds_of = [&](int j, int ny) {
        const int npix = (ny + 2 * (nl - 1 - j)) * (Wi - 2 * (j + 1));
        return (((npix + 31) >> 5) + 3) >> 2;
    };

    if (loader) {
        // the NEXT item's input (its rows of every pixel's 64 bytes: one contiguous run) is fetched into registers while the compute
        // waves work and dumped into the image between the item's last and the next item's first barrier
        const int lt = cw * 64 + lane;
        vu4 d[CB_NU];
        int nchunk = 0;
        auto issue = [&](int item) {
            const ChainItem it = chain_bf_item(a, item);
            nchunk = (it.ny + 2 * nl) * Wi * 4;
            const vu4 *src = a.in + ((size_t)(it.b * a.Hi + it.y0) * Wi) * 4;
#pragma unroll
            for (int u = 0; u < CB_NU; ++u)
                if (256 * u < nchunk) d[u] = src[min(u * 256 + lt, nchunk - 1)]; // (wave-uniform test; idempotent tail)
        };
        auto dump = [&]() {
#pragma unroll
            for (int u = 0; u < CB_NU; ++u)
                if (256 * u < nchunk) {
                    const int q = min(u * 256 + lt, nchunk - 1);
                    *reinterpret_cast<vu4 *>(img + (q >> 2) * CB_SLOT + (q & 3) * 16) = d[u];
                }
        };
        if (first < n) {
            issue(first);
            dump();
        }
        for (int item = first; item < n; item += G) {
            __syncthreads(); // A: the image is ready
            const int nxt = item + G;
            if (nxt < n) issue(nxt);
            const int ny = chain_bf_item(a, item).ny;
            int nb = 0;
            for (int j = 0; j + 1 < nl; ++j) nb += rounds_of(j, ny) + 1;
            for (int i = 0; i < nb; ++i) __syncthreads();
            __syncthreads(); // F: every read of the image is done
            if (nxt < n) dump();
        }
        return;
    }

    // ---- compute waves
    bf16x8c wc[18], wn[18]; // this layer's fragments / the next layer's (in flight while this one computes)
    auto load_next = [&](const vu4 *wf) {
#pragma unroll
        for (int i = 0; i < 18; ++i) wn[i] = __builtin_bit_cast(bf16x8c, wf[i * 64 + lane]);
    };
    float nx_bv = 0.0f;
    float *lbias = reinterpret_cast<float *>(ldsb) + 32 * cw;
    char *dump_slot = ldsb + 512 + 96 * cw;
    const __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc(
        a.out, 0, (int)((unsigned)(a.B * 32 * Hf * Wf) * (OUT_F32 ? 4u : 2u)), 0x00020000);
    constexpr unsigned kNoStore = 0xfffffff0u;
    const int chw = Hf * Wf;
    typedef uint32_t vu2 __attribute__((ext_vector_type(2)));
    if (first < n) {
        const int g0 = chain_bf_item(a, first).grp;
        load_next(a.wfrag[g0][0]);
        nx_bv = (a.bias[g0][0] && lane < 32) ? a.bias[g0][0][lane] : 0.0f;
    }

    for (int item = first; item < n; item += G) {
        const ChainItem it = chain_bf_item(a, item);
        __syncthreads(); // A
        auto layer = [&](int j, auto last_tag) {
            constexpr bool last = decltype(last_tag)::value;
            const int Wout = Wi - 2 * (j + 1), nrow = it.ny + 2 * (nl - 1 - j);
            const int npix = nrow * Wout, ntiles = (npix + 31) >> 5, rounds = (ntiles + 3) >> 2;
            const uint32_t magic = a.magic_w[j];
#pragma unroll
            for (int i = 0; i < 18; ++i) wc[i] = wn[i];
            if (lane < 32) lbias[lane] = nx_bv;
            { // request the next layer's (the next item's first layer's) fragments and bias: a whole layer to arrive in
                const int gn = !last ? it.grp : (item + G < n ? chain_bf_item(a, item + G).grp : it.grp);
                const int jn = !last ? j + 1 : 0;
                load_next(a.wfrag[gn][jn]);
                nx_bv = (a.bias[gn][jn] && lane < 32) ? a.bias[gn][jn][lane] : 0.0f;
            }
            const bool relu = !last || a.relu_last;
            const int relu_floor = __builtin_amdgcn_readfirstlane(relu ? 0 : (int)0x80000000);
            const short relu16 = (short)__builtin_amdgcn_readfirstlane(relu ? 0 : -32768);
            // a copy of the layer's output for the backward pass (bounded by its own size: a lane that owns no row names an offset
            // beyond it and the store is dropped)
            const int Hj = a.Hi - 2 * (j + 1);
            void *keep = last ? nullptr : a.acts[j];
            const __amdgpu_buffer_rsrc_t rs_keep = __builtin_amdgcn_make_buffer_rsrc(
                keep ? keep : a.out, 0, keep ? (int)((unsigned)(a.B * Hj * Wout) * 64u) : 0, 0x00020000);
            const bool last_part = it.y0 + it.ny >= Hf;
            unsigned pkoff = kNoStore;

            auto locate = [&](int tl, bool &valid_, int &off_, int &yy_, int &xx_) {
                int m_ = tl * 32 + l31;
                valid_ = m_ < npix;
                if (!valid_) m_ = npix - 1;
                yy_ = (int)__umulhi((uint32_t)m_, magic);
                xx_ = m_ - yy_ * Wout;
                off_ = (yy_ * Wi + xx_) * CB_SLOT + h * 16;
            };
            auto rd = [&](int off_, int i) { // operand i = (tap i / 2, channel half i & 1): row base + immediate
                const int t = i >> 1;
                return *reinterpret_cast<const vu4 *>(img + off_ + (t / 3) * rowb + ((t % 3) * CB_SLOT + (i & 1) * 32));
            };
            f32x16 prev;
#pragma unroll
            for (int q = 0; q < 16; ++q) prev[q] = 0.0f;
            char *pdst = dump_slot;
            unsigned pboff = kNoStore;
            bool pending = false;
            uint32_t pk[8];
            // the previous tile's epilogue, one slice per matrix-instruction gap (18 of them): pair p = channels (2p, 2p + 1) of
            // the lane's 16 is converted in gap 2p (f32 pair -> packed bf16, RNE), gets its ReLU in gap 2p + 1 (packed signed
            // 16-bit maximum with 0) and is stored with its neighbour pair (8 bytes) two gaps later
            auto epi_slot = [&](int g) {
                if (last && OUT_F32) {
                    if (g < 16) {
                        const float t = __int_as_float(max(__float_as_int(prev[g]), relu_floor));
                        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, t), rs_out, pboff,
                                                              (unsigned)(((g & 3) + 8 * (g >> 2)) * chw) * 4u, 0);
                    }
                    return;
                }
                if (g < 16 && !(g & 1)) {
                    typedef float f32x2c __attribute__((ext_vector_type(2)));
                    pk[g >> 1] = __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2c{prev[g], prev[g + 1]}, bf16x2c));
                }
                if (g < 16 && (g & 1))
                    pk[g >> 1] = __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(s16x2c, pk[g >> 1]), s16x2c{relu16, relu16}));
                if (g >= 5 && (g & 3) == 1 && g <= 17) { // channels 8 g4 + 4 h + (0..3): pairs 2 g4, 2 g4 + 1
                    const int g4 = (g - 5) >> 2;
                    if (last) // bf16 NHWC out: 8 bytes at pixel * 64 + 16 g4 + 8 h
                        __builtin_amdgcn_raw_buffer_store_b64(vu2{pk[2 * g4], pk[2 * g4 + 1]}, rs_out, pboff, (unsigned)(16 * g4), 0);
                    else {
                        *reinterpret_cast<vu2 *>(pdst + 16 * g4) = vu2{pk[2 * g4], pk[2 * g4 + 1]};
                        __builtin_amdgcn_raw_buffer_store_b64(vu2{pk[2 * g4], pk[2 * g4 + 1]}, rs_keep, pkoff, (unsigned)(16 * g4), 0);
                    }
                }
            };
            auto epilogue_now = [&]() {
#pragma unroll
                for (int g = 0; g < 18; ++g) epi_slot(g);
            };
            bool valid = false;
            int off = 0, yy = 0, xx = 0;
            constexpr int AH = 6; // operand reads run six matrix instructions ahead of their use, across tile boundaries
            vu4 raw[AH];
            if (cw < ntiles) {
                locate(cw, valid, off, yy, xx);
#pragma unroll
                for (int i = 0; i < AH; ++i) raw[i] = rd(off, i);
            }
            for (int r = 0; r < rounds; ++r) {
                const int tl = 4 * r + cw;
                if (tl < ntiles) { // wave-uniform
                    bool valid2;
                    int off2, yy2, xx2;
                    locate(tl + 4 < ntiles ? tl + 4 : tl, valid2, off2, yy2, xx2); // no next tile: re-read this one
                    f32x16 acc;
#pragma unroll
                    for (int g4 = 0; g4 < 4; ++g4) {
                        const vf4 b4 = *reinterpret_cast<const vf4 *>(lbias + 8 * g4 + 4 * h);
                        acc[4 * g4] = b4[0]; acc[4 * g4 + 1] = b4[1]; acc[4 * g4 + 2] = b4[2]; acc[4 * g4 + 3] = b4[3];
                    }
#pragma unroll
                    for (int i = 0; i < 18; ++i) {
                        const vu4 cur = raw[i % AH];
                        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wc[i], __builtin_bit_cast(bf16x8c, cur), acc, 0, 0, 0);
                        __builtin_amdgcn_sched_barrier(0);
                        raw[i % AH] = i + AH < 18 ? rd(off, i + AH) : rd(off2, i + AH - 18);
                        epi_slot(i);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    prev = acc;
                    if (last) {
                        pdst = dump_slot;
                        pboff = !valid ? kNoStore
                                : OUT_F32 ? (unsigned)((it.b * 32 * Hf + (it.y0 + yy)) * Wf + xx + 4 * h * chw) * 4u
                                          : (unsigned)(((it.b * Hf + it.y0 + yy) * Wf + xx) * 64 + 8 * h);
                    } else {
                        pdst = valid ? img + (yy * Wi + xx) * CB_SLOT + 8 * h : dump_slot;
                        // rows [y0, y0 + ny) of every layer belong to this part, the rows behind them to the next one (the last part: all)
                        pkoff = (valid && (yy < it.ny || last_part)) ? (unsigned)(((it.b * Hj + it.y0 + yy) * Wout + xx) * 64 + 8 * h) : kNoStore;
                    }
                    pending = true;
                    valid = valid2; off = off2; yy = yy2; xx = xx2;
                } else if (pending) { // out of tiles: finish the last one (its round's barrier is behind us)
                    epilogue_now();
                    pending = false;
                }
                if (!last) __syncthreads(); // R: the round's operand reads are done
            }
            if (pending) epilogue_now();
            if (!last) __syncthreads(); // L: the layer's output is in place
        };
        for (int j = 0; j + 1 < nl; ++j) layer(j, std::false_type{});
        layer(nl - 1, std::true_type{});
        __syncthreads(); // F
    }
}


// =========================================================================================================================
// bf16 mode, BACKWARD-DATA of the whole stack in one launch.  dZ of layer i - 1's output = full correlation of dZ_i with the
// flipped, transposed weights, gated by the ReLU of layer i's input -- a map that GROWS by two pixels per step.  On a zeroed
// canvas of (Ht + 2 nl + 2)^2 pixel slots with the incoming gradient placed at origin (2 nl, 2 nl), step s's output pixel at
// canvas (r, c) is sum_k W'[k] canvas(r + ky, c + kx): the forward chain's access pattern (a pixel reads at and behind itself),
// so every step runs IN PLACE exactly like a forward layer, its map's origin moving two pixels up and left; what lies behind
// and below a map is zero because nothing has been written there yet.  Every step's output is also stored (bf16 NHWC) for the
// backward-weights launches that follow; nothing is read back.  One work item = one image (the 43 x 43 canvas of the training
// stack is 145 KiB).  Same products, order and roundings as k_conv32_bf16<1>: bit-identical gradients.
constexpr int CW_NU = 8; // loader rounds for the incoming map: 256 lanes x 8 chunks of 16 B = 512 pixels

struct ChainBwArgs {
    const vu4 *dz_top;            // bf16 NHWC (B,Ht,Wt,32)
    const vu4 *wfrag[CB_MAXL];    // step s: k_bf16_prep mode-1 fragments of layer n - 1 - s
    const void *mask[CB_MAXL];    // step s: that layer's INPUT (bf16 NHWC (B,Ht+2s+2,Wt+2s+2,32)): the gate of the step's output; or null
    void *dz_out[CB_MAXL];        // step s: receives the step's output, bf16 NHWC (B,Ht+2s+2,Wt+2s+2,32)
    int nl, B, Ht, Wt;
    uint32_t magic_w[CB_MAXL];    // floor(2^32 / (Wt + 2 s + 2)) + 1
    uint32_t magic_p;             // floor(2^32 / canvas pitch) + 1
};

__global__ __launch_bounds__(512, 2) void k_bwd32_chain_bf16(ChainBwArgs a)
{
    extern __shared__ __attribute__((aligned(16))) char ldsw[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool loader = wave >= 4;
    const int cw = wave & 3;
    const int l31 = lane & 31, h = lane >> 5;
    const int nl = a.nl, Ht = a.Ht, Wt = a.Wt;
    const int P = Wt + 2 * nl + 2, R = Ht + 2 * nl + 2; // canvas pitch / rows (two spare zero columns / rows at the far edges)
    char *img = ldsw + CB_HDR;
    const int G = (int)gridDim.x, n = a.B, first = xcd_slot((int)blockIdx.x, G);
    const int rowb = P * CB_SLOT;
    auto rounds_of = [&](int s_) {
        const int npix = (Ht + 2 * s_ + 2) * (Wt + 2 * s_ + 2);
        return (((npix + 31) >> 5) + 3) >> 2;
    };

    if (loader) {
        const int lt = cw * 64 + lane;
        vu4 d[CW_NU];
        const int nchunk = Ht * Wt * 4;
        auto issue = [&](int b) {
            const vu4 *src = a.dz_top + (size_t)b * nchunk;
#pragma unroll
            for (int u = 0; u < CW_NU; ++u)
                if (256 * u < nchunk) d[u] = src[min(u * 256 + lt, nchunk - 1)];
        };
        auto dump = [&](bool first_item) {
            // zero what the previous image left (its last map covers rows / columns [0, 2 nl + Ht)); the first time: everything.
            // The incoming map's own pixels are skipped: they are written below, by other lanes, in no particular order
            const int zr = first_item ? R : Ht + 2 * nl;
            const vu4 zero = {0u, 0u, 0u, 0u};
            for (int id = lt; id < zr * P * 4; id += 256) {
                const int px = id >> 2;
                const int r = (int)__umulhi((uint32_t)px, a.magic_p), c = px - r * P;
                const bool inside = r >= 2 * nl && r < 2 * nl + Ht && c >= 2 * nl && c < 2 * nl + Wt;
                if (!inside) *reinterpret_cast<vu4 *>(img + px * CB_SLOT + (id & 3) * 16) = zero;
            }
#pragma unroll
            for (int u = 0; u < CW_NU; ++u)
                if (256 * u < nchunk) {
                    const int q = min(u * 256 + lt, nchunk - 1), p_ = q >> 2;
                    const int y = p_ / Wt, x = p_ - y * Wt;
                    *reinterpret_cast<vu4 *>(img + ((2 * nl + y) * P + 2 * nl + x) * CB_SLOT + (q & 3) * 16) = d[u];
                }
        };
        if (first < n) {
            issue(first);
            dump(true);
        }
        int nb = 0;
        for (int s_ = 0; s_ + 1 < nl; ++s_) nb += rounds_of(s_) + 1;
        for (int item = first; item < n; item += G) {
            __syncthreads(); // A
            const int nxt = item + G;
            if (nxt < n) issue(nxt);
            for (int i = 0; i < nb; ++i) __syncthreads();
            __syncthreads(); // F
            if (nxt < n) dump(false);
        }
        return;
    }

    bf16x8c wc[18], wn[18];
    auto load_next = [&](const vu4 *wf) {
#pragma unroll
        for (int i = 0; i < 18; ++i) wn[i] = __builtin_bit_cast(bf16x8c, wf[i * 64 + lane]);
    };
    char *dump_slot = ldsw + 512 + 96 * cw;
    constexpr unsigned kNoStore = 0xfffffff0u;
    typedef uint32_t vu2 __attribute__((ext_vector_type(2)));
    typedef unsigned short u16x2c __attribute__((ext_vector_type(2)));
    if (first < n) load_next(a.wfrag[0]);

    for (int item = first; item < n; item += G) {
        __syncthreads(); // A
        auto step = [&](int s_, auto last_tag) {
            constexpr bool last = decltype(last_tag)::value;
            const int Ho = Ht + 2 * s_ + 2, Wo = Wt + 2 * s_ + 2, oo = 2 * (nl - s_) - 2; // output size, its origin on the canvas
            const int npix = Ho * Wo, ntiles = (npix + 31) >> 5, rounds = (ntiles + 3) >> 2;
            const uint32_t magic = a.magic_w[s_];
#pragma unroll
            for (int i = 0; i < 18; ++i) wc[i] = wn[i];
            load_next(a.wfrag[last ? 0 : s_ + 1]); // (the next image starts with step 0's again)
            const unsigned osize = (unsigned)(a.B * Ho * Wo) * 64u;
            const __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc(a.dz_out[s_], 0, (int)osize, 0x00020000);
            // no gate: the loads read the output buffer itself (valid memory) and `gate_and` lets everything pass
            const bool gated = a.mask[s_] != nullptr;
            const __amdgpu_buffer_rsrc_t rs_mask = __builtin_amdgcn_make_buffer_rsrc(
                gated ? const_cast<void *>(a.mask[s_]) : a.dz_out[s_], 0, (int)osize, 0x00020000);
            const uint32_t gate_or = (uint32_t)__builtin_amdgcn_readfirstlane(gated ? 0 : (int)0x00010001);

            auto locate = [&](int tl, bool &valid_, int &off_, unsigned &goff_) {
                int m_ = tl * 32 + l31;
                valid_ = m_ < npix;
                if (!valid_) m_ = npix - 1;
                const int yy_ = (int)__umulhi((uint32_t)m_, magic), xx_ = m_ - yy_ * Wo;
                off_ = ((oo + yy_) * P + oo + xx_) * CB_SLOT + h * 16;
                goff_ = (unsigned)(((item * Ho + yy_) * Wo + xx_) * 64 + 8 * h);
            };
            auto rd = [&](int off_, int i) {
                const int t = i >> 1;
                return *reinterpret_cast<const vu4 *>(img + off_ + (t / 3) * rowb + ((t % 3) * CB_SLOT + (i & 1) * 32));
            };
            f32x16 prev;
#pragma unroll
            for (int q = 0; q < 16; ++q) prev[q] = 0.0f;
            char *pdst = dump_slot;
            unsigned pgoff = kNoStore;
            vu2 pgate[4] = {vu2{0u, 0u}, vu2{0u, 0u}, vu2{0u, 0u}, vu2{0u, 0u}};
            bool pending = false;
            uint32_t pk[8];
            // the previous tile's epilogue in the gaps: pair p = channels (2p, 2p + 1) of the lane's 16 -> packed bf16 (gap 2p), times the
            // gate (a_prev != 0, as min(|bits|, 1) per 16-bit half: gap 2p + 1), stored with its neighbour pair two gaps later
            auto epi_slot = [&](int g) {
                if (g < 16 && !(g & 1)) {
                    typedef float f32x2c __attribute__((ext_vector_type(2)));
                    pk[g >> 1] = __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2c{prev[g], prev[g + 1]}, bf16x2c));
                }
                if (g < 16 && (g & 1)) {
                    const int p_ = g >> 1;
                    const uint32_t gb = (pgate[p_ >> 1][p_ & 1] & 0x7fff7fffu) | gate_or; // -0 counts as 0; no gate: every half is 1
                    const u16x2c keep = __builtin_elementwise_min(__builtin_bit_cast(u16x2c, gb), u16x2c{1, 1});
                    pk[p_] = __builtin_bit_cast(uint32_t, (u16x2c)(__builtin_bit_cast(u16x2c, pk[p_]) * keep));
                }
                if (g >= 5 && (g & 3) == 1 && g <= 17) {
                    const int g4 = (g - 5) >> 2;
                    if (!last) *reinterpret_cast<vu2 *>(pdst + 16 * g4) = vu2{pk[2 * g4], pk[2 * g4 + 1]};
                    __builtin_amdgcn_raw_buffer_store_b64(vu2{pk[2 * g4], pk[2 * g4 + 1]}, rs_out, pgoff, (unsigned)(16 * g4), 0);
                }
            };
            auto epilogue_now = [&]() {
#pragma unroll
                for (int g = 0; g < 18; ++g) epi_slot(g);
            };
            bool valid = false;
            int off = 0;
            unsigned goff = 0u;
            constexpr int AH = 6;
            vu4 raw[AH];
            if (cw < ntiles) {
                locate(cw, valid, off, goff);
#pragma unroll
                for (int i = 0; i < AH; ++i) raw[i] = rd(off, i);
            }
            for (int r = 0; r < rounds; ++r) {
                const int tl = 4 * r + cw;
                if (tl < ntiles) {
                    bool valid2;
                    int off2;
                    unsigned goff2;
                    locate(tl + 4 < ntiles ? tl + 4 : tl, valid2, off2, goff2);
                    vu2 gcur[4]; // this tile's gates: requested now, used in its epilogue (one tile later)
#pragma unroll
                    for (int g4 = 0; g4 < 4; ++g4)
                        gcur[g4] = __builtin_bit_cast(vu2, __builtin_amdgcn_raw_buffer_load_b64(rs_mask, valid ? goff : 0u, (unsigned)(16 * g4), 0));
                    f32x16 acc;
#pragma unroll
                    for (int q = 0; q < 16; ++q) acc[q] = 0.0f;
#pragma unroll
                    for (int i = 0; i < 18; ++i) {
                        const vu4 cur = raw[i % AH];
                        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wc[i], __builtin_bit_cast(bf16x8c, cur), acc, 0, 0, 0);
                        __builtin_amdgcn_sched_barrier(0);
                        raw[i % AH] = i + AH < 18 ? rd(off, i + AH) : rd(off2, i + AH - 18);
                        epi_slot(i);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    prev = acc;
#pragma unroll
                    for (int g4 = 0; g4 < 4; ++g4) pgate[g4] = gcur[g4];
                    pdst = (valid && !last) ? img + off - h * 16 + 8 * h : dump_slot;
                    pgoff = valid ? goff : kNoStore;
                    pending = true;
                    valid = valid2; off = off2; goff = goff2;
                } else if (pending) {
                    epilogue_now();
                    pending = false;
                }
                if (!last) __syncthreads(); // R
            }
            if (pending) epilogue_now();
            if (!last) __syncthreads(); // L
        };
        for (int s_ = 0; s_ + 1 < nl; ++s_) step(s_, std::false_type{});
        step(nl - 1, std::true_type{});
        __syncthreads(); // F
    }
}

} // namespace augcl

using namespace augcl;

namespace {

int cu_count_chain()
{
    int dev = 0, v = 256;
    if (hipGetDevice(&dev) == hipSuccess) {
        int n = 0;
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0) v = n;
    }
    return v;
}

constexpr int kLdsBytes = 160 * 1024;

} // namespace

extern "C" {

#ifdef AUGCL_TUNING
int augcl_tuning_chain_read_clk(unsigned long long *out8)
{
    return hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_chain_clk), 64) == hipSuccess ? AUGCL_OK : AUGCL_E_LAUNCH;
}
#endif

// Plan of one launch: `nl` layers starting at a (Hi x Wi) input, final rows split in `parts` per image.
// Returns the number of tile rounds of the longest item (the launch's critical path in units of ~1 us), or -1 if it does not fit.
int augcl_internal_chain_rounds(int Hi, int Wi, int nl, int parts)
{
    if (nl < 1 || nl > CH_MAXL || parts < 1) return -1;
    const int Hf = Hi - 2 * nl, Wf = Wi - 2 * nl;
    if (Hf < 1 || Wf < 1 || parts > Hf) return -1;
    const int rows_out = (Hf + parts - 1) / parts;
    if ((parts - 1) * rows_out >= Hf) return -1; // an empty last part
    const long npx = (long)(rows_out + 2 * nl) * Wi;
    if (npx > 256L * CH_NU || npx < 4 || CH_HDR + npx * X2_SLOT > kLdsBytes) return -1;
    int rounds = 0;
    for (int j = 0; j < nl; ++j) {
        const int npix = (rows_out + 2 * (nl - 1 - j)) * (Wi - 2 * (j + 1));
        rounds += (((npix + 31) >> 5) + 3) >> 2;
    }
    return rounds;
}

// One chain launch.  wfrag[set][j] / bias[set][j]: the prepared fragments (k_x2_prep, mode 0) and biases of its layers; sets = 2:
// images [B/2, B) use set 1.  +1: the shape is not covered.
static int chain_forward_launch(const char *name, const float *in, float *out, const void *const *wfragA, const float *const *biasA,
                                const void *const *wfragB, const float *const *biasB, int nl, int B, int Hi, int Wi, int parts,
                                int relu_last, const float *amax_in, float *amax_out, bool keep, void *const *acts_out,
                                void *const *bits_out, float *amax_acts, augcl_stream_t stream)
{
    AUGCL_REQUIRE(in && out && wfragA && biasA && amax_in, "%s: null tensor", name);
    if (B == 0) return AUGCL_OK;
    if (augcl_internal_chain_rounds(Hi, Wi, nl, parts) < 0) return 1;
    if ((size_t)B * 32 * Hi * Wi >= ((size_t)1 << 30) - 8) return 1;
    AUGCL_REQUIRE(wfragB == nullptr || (B & 1) == 0, "%s: a two-weight-set batch must be even", name);
    AUGCL_REQUIRE(aligned4(in) && aligned4(out), "%s: tensors must be 4-byte aligned", name);
    ChainArgs a{};
    a.in = in; a.out = out;
    for (int j = 0; j < nl; ++j) {
        a.wfrag[0][j] = reinterpret_cast<const vu4 *>(wfragA[j]);
        a.bias[0][j] = biasA[j];
        a.wfrag[1][j] = reinterpret_cast<const vu4 *>(wfragB ? wfragB[j] : wfragA[j]);
        a.bias[1][j] = wfragB ? biasB[j] : biasA[j];
        a.magic_w[j] = (uint32_t)(0x100000000ull / (unsigned)(Wi - 2 * (j + 1))) + 1u;
    }
    a.nl = nl; a.B = B; a.Hi = Hi; a.Wi = Wi; a.parts = parts;
    const int Hf = Hi - 2 * nl;
    a.rows_out = (Hf + parts - 1) / parts;
    a.relu_last = relu_last;
    a.dual = wfragB != nullptr;
    a.n_items = B * parts;
    a.magic_parts = (uint32_t)(0x100000000ull / (unsigned)parts) + 1u;
    a.amax_in = reinterpret_cast<const uint32_t *>(amax_in);
    a.amax_out = reinterpret_cast<uint32_t *>(amax_out);
    if (keep) {
        AUGCL_REQUIRE(wfragB == nullptr && acts_out && amax_acts, "%s: the kept forward takes one weight set, the activation buffers and the maxima", name);
        for (int j = 0; j < nl; ++j) {
            if (j + 1 < nl) {
                AUGCL_REQUIRE(acts_out[j] != nullptr && aligned4(acts_out[j]), "%s: null / misaligned activation buffer", name);
                a.acts[j] = reinterpret_cast<float *>(acts_out[j]);
                a.amax_acts[j] = reinterpret_cast<uint32_t *>(amax_acts + j);
            }
            a.bits[j] = bits_out ? reinterpret_cast<uint32_t *>(bits_out[j]) : nullptr;
        }
        if (!relu_last) a.bits[nl - 1] = nullptr; // gate words are those of a ReLU layer
    }
    const size_t lds = (size_t)CH_HDR + (size_t)(a.rows_out + 2 * nl) * Wi * X2_SLOT;
    const void *fn = keep ? reinterpret_cast<const void *>(&k_conv32_chain<true>) : reinterpret_cast<const void *>(&k_conv32_chain<false>);
    hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return augcl::fail(AUGCL_E_LAUNCH, "%s: cannot set dynamic LDS %zu: %s", name, lds, hipGetErrorString(e));
    const int cus = cu_count_chain();
    const int grid = a.n_items < cus ? a.n_items : cus;
    if (keep) hipLaunchKernelGGL(k_conv32_chain<true>, dim3(grid), dim3(512), lds, as_stream(stream), a);
    else hipLaunchKernelGGL(k_conv32_chain<false>, dim3(grid), dim3(512), lds, as_stream(stream), a);
    return augcl::check_launch(name);
}

int augcl_internal_chain_forward(const float *in, float *out, const void *const *wfragA, const float *const *biasA,
                                 const void *const *wfragB, const float *const *biasB, int nl, int B, int Hi, int Wi, int parts,
                                 int relu_last, const float *amax_in, float *amax_out, augcl_stream_t stream)
{
    return chain_forward_launch("augcl_enc_stack_infer", in, out, wfragA, biasA, wfragB, biasB, nl, B, Hi, Wi, parts, relu_last, amax_in,
                                amax_out, false, nullptr, nullptr, nullptr, stream);
}

// The same launch for the GRADIENT-CARRYING forward: acts_out[j] (j < nl - 1) receives layer j's float32 output, bits_out[j] (j < nl,
// entries may be null) its gate words, amax_acts[j] (j < nl - 1; zeroed by the caller) its maximum; the last layer's output, gate
// words and maximum go to out / bits_out[nl - 1] / amax_out as in the no-grad launch.
int augcl_internal_chain_forward_keep(const float *in, float *out, const void *const *wfrag, const float *const *bias, int nl, int B,
                                      int Hi, int Wi, int parts, int relu_last, const float *amax_in, float *amax_out,
                                      void *const *acts_out, void *const *bits_out, float *amax_acts, augcl_stream_t stream)
{
    return chain_forward_launch("augcl_enc_stack_forward_chained", in, out, wfrag, bias, nullptr, nullptr, nl, B, Hi, Wi, parts, relu_last,
                                amax_in, amax_out, true, acts_out, bits_out, amax_acts, stream);
}

// ---- bf16 performance mode (k_conv32_chain_bf16) ----
int augcl_internal_chain_bf16_rounds(int Hi, int Wi, int nl, int parts)
{
    if (nl < 1 || nl > CB_MAXL || parts < 1) return -1;
    const int Hf = Hi - 2 * nl, Wf = Wi - 2 * nl;
    if (Hf < 1 || Wf < 1 || parts > Hf) return -1;
    const int rows_out = (Hf + parts - 1) / parts;
    if ((parts - 1) * rows_out >= Hf) return -1;
    const long npx = (long)(rows_out + 2 * nl) * Wi;
    if (npx > 64L * CB_NU || npx < 1 || CB_HDR + npx * CB_SLOT > kLdsBytes) return -1;
    int rounds = 0;
    for (int j = 0; j < nl; ++j) {
        const int npix = (rows_out + 2 * (nl - 1 - j)) * (Wi - 2 * (j + 1));
        rounds += (((npix + 31) >> 5) + 3) >> 2;
    }
    return rounds;
}

// bf16 fragments of n (<= 32) 32 -> 32 layers into out, `stride` bytes apart (>= 18 KiB each)
int augcl_internal_bf16_prep_weights_mode(const float *const *w, int n, void *out, int stride, int mode, augcl_stream_t stream);
int augcl_internal_bf16_prep_weights(const float *const *w, int n, void *out, int stride, augcl_stream_t stream)
{
    return augcl_internal_bf16_prep_weights_mode(w, n, out, stride, 0, stream);
}

int augcl_internal_bf16_prep_weights_mode(const float *const *w, int n, void *out, int stride, int mode, augcl_stream_t stream)
{
    const char *name = "augcl_internal_bf16_prep_weights";
    AUGCL_REQUIRE(w && out && n >= 1 && n <= 32 && stride >= CB_FRAG_BYTES && (stride & 15) == 0 && aligned16(out), "%s: bad argument", name);
    PrepBfArgs a{};
    for (int i = 0; i < n; ++i) {
        AUGCL_REQUIRE(w[i] != nullptr, "%s: null weight tensor", name);
        a.w[i] = w[i];
    }
    a.out = reinterpret_cast<char *>(out);
    a.stride = stride;
    a.mode = mode;
    hipLaunchKernelGGL(k_bf16_prep, dim3(5, n), dim3(256), 0, as_stream(stream), a);
    return augcl::check_launch(name);
}

int augcl_internal_chain_bf16_forward(const void *in_bf16_nhwc, void *out, const void *const *wfragA, const float *const *biasA,
                                      const void *const *wfragB, const float *const *biasB, int nl, int B, int Hi, int Wi, int parts,
                                      int relu_last, int out_f32_nchw, void *const *acts_out /* nl - 1 buffers or null */,
                                      augcl_stream_t stream)
{
    const char *name = "augcl_enc_stack_infer";
    AUGCL_REQUIRE(in_bf16_nhwc && out && wfragA && biasA, "%s: null tensor", name);
    if (B == 0) return AUGCL_OK;
    if (augcl_internal_chain_bf16_rounds(Hi, Wi, nl, parts) < 0) return 1;
    if ((size_t)B * 32 * Hi * Wi >= ((size_t)1 << 30) - 8) return 1;
    AUGCL_REQUIRE(wfragB == nullptr || (B & 1) == 0, "%s: a two-weight-set batch must be even", name);
    AUGCL_REQUIRE(aligned16(in_bf16_nhwc) && aligned4(out), "%s: misaligned tensor", name);
    ChainBfArgs a{};
    a.in = reinterpret_cast<const vu4 *>(in_bf16_nhwc);
    a.out = out;
    for (int j = 0; j < nl; ++j) {
        a.wfrag[0][j] = reinterpret_cast<const vu4 *>(wfragA[j]);
        a.bias[0][j] = biasA[j];
        a.wfrag[1][j] = reinterpret_cast<const vu4 *>(wfragB ? wfragB[j] : wfragA[j]);
        a.bias[1][j] = wfragB ? biasB[j] : biasA[j];
        a.magic_w[j] = (uint32_t)(0x100000000ull / (unsigned)(Wi - 2 * (j + 1))) + 1u;
        a.acts[j] = (acts_out && j + 1 < nl) ? acts_out[j] : nullptr;
    }
    a.nl = nl; a.B = B; a.Hi = Hi; a.Wi = Wi; a.parts = parts;
    const int Hf = Hi - 2 * nl;
    a.rows_out = (Hf + parts - 1) / parts;
    a.relu_last = relu_last;
    a.out_f32 = out_f32_nchw;
    a.dual = wfragB != nullptr;
    a.n_items = B * parts;
    a.magic_parts = (uint32_t)(0x100000000ull / (unsigned)parts) + 1u;
    const size_t lds = (size_t)CB_HDR + (size_t)(a.rows_out + 2 * nl) * Wi * CB_SLOT;
    const int cus = cu_count_chain();
    const int grid = a.n_items < cus ? a.n_items : cus;
    hipError_t e;
    if (out_f32_nchw) {
        e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_conv32_chain_bf16<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return augcl::fail(AUGCL_E_LAUNCH, "%s: cannot set dynamic LDS %zu: %s", name, lds, hipGetErrorString(e));
        hipLaunchKernelGGL(k_conv32_chain_bf16<true>, dim3(grid), dim3(512), lds, as_stream(stream), a);
    } else {
        e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_conv32_chain_bf16<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return augcl::fail(AUGCL_E_LAUNCH, "%s: cannot set dynamic LDS %zu: %s", name, lds, hipGetErrorString(e));
        hipLaunchKernelGGL(k_conv32_chain_bf16<false>, dim3(grid), dim3(512), lds, as_stream(stream), a);
    }
    return augcl::check_launch(name);
}

// Backward-data of nl (<= 12) consecutive 32 -> 32 layers in one launch (bf16 mode).  Step s handles layer n - 1 - s: wfrag[s] = its
// mode-1 fragments, mask[s] = its input (the ReLU gate, or null), dz_out[s] receives the gradient w.r.t. that input's
// producer.  Returns +1 when the canvas does not fit LDS (the caller runs the per-layer kernels).
int augcl_internal_chain_bf16_backward(const void *dz_top, const void *const *wfrag, const void *const *mask, void *const *dz_out,
                                       int nl, int B, int Ht, int Wt, augcl_stream_t stream)
{
    const char *name = "augcl_enc_stack_backward";
    AUGCL_REQUIRE(dz_top && wfrag && mask && dz_out && B >= 1, "%s: null tensor", name);
    if (nl < 1 || nl > CB_MAXL || Ht < 1 || Wt < 1) return 1;
    const int P = Wt + 2 * nl + 2, R = Ht + 2 * nl + 2;
    if ((size_t)CB_HDR + (size_t)R * P * CB_SLOT > (size_t)kLdsBytes || Ht * Wt * 4 > 256 * CW_NU) return 1;
    if ((size_t)B * (Ht + 2 * nl) * (Wt + 2 * nl) * 64 >= ((size_t)1 << 31)) return 1;
    ChainBwArgs a{};
    a.dz_top = reinterpret_cast<const vu4 *>(dz_top);
    for (int s_ = 0; s_ < nl; ++s_) {
        AUGCL_REQUIRE(wfrag[s_] && dz_out[s_] && aligned16(dz_out[s_]), "%s: null / misaligned step buffer", name);
        a.wfrag[s_] = reinterpret_cast<const vu4 *>(wfrag[s_]);
        a.mask[s_] = mask[s_];
        a.dz_out[s_] = dz_out[s_];
        a.magic_w[s_] = (uint32_t)(0x100000000ull / (unsigned)(Wt + 2 * s_ + 2)) + 1u;
    }
    a.nl = nl; a.B = B; a.Ht = Ht; a.Wt = Wt;
    a.magic_p = (uint32_t)(0x100000000ull / (unsigned)P) + 1u;
    const size_t lds = (size_t)CB_HDR + (size_t)R * P * CB_SLOT;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_bwd32_chain_bf16), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return augcl::fail(AUGCL_E_LAUNCH, "%s: cannot set dynamic LDS %zu: %s", name, lds, hipGetErrorString(e));
    const int cus = cu_count_chain();
    const int grid = B < cus ? B : cus;
    hipLaunchKernelGGL(k_bwd32_chain_bf16, dim3(grid), dim3(512), lds, as_stream(stream), a);
    return augcl::check_launch(name);
}

} // extern "C"
