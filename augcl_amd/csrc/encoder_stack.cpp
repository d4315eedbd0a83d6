// encoder_stack.cpp -- whole-stack entry points: one C call runs every layer of the conv encoder forward
// (or backward), so the host pays one FFI crossing per pass instead of one per layer kernel (the bf16 mode's
// ~5 ms update had become CPU-launch-bound: ~90 ctypes calls + ~60 tensor allocations per step).
#include "augcl_common.h"

extern "C" {

static inline int out_hw(int h, int stride) { return (h - 3) / stride + 1; }

int augcl_internal_x3_conv32_forward(const float *in, const float *w, const float *bias, const float *w2, const float *bias2,
                                     float *out, int B, int Hi, int Wi, int relu, const void *wfrag, const void *wfrag2,
                                     augcl_stream_t stream);
int augcl_internal_x3_conv32_backward_data(const float *dz, const float *w, const float *act_prev, float *dact, int B, int Ho,
                                           int Wo, const void *wfrag, augcl_stream_t stream);
int augcl_internal_x3_prep_weights(const float *const *w, int n, int mode, void *out, augcl_stream_t stream);
// encoder_x2.hip (f16-pair kernels, precision 3); the layer calls return +1 for a shape they do not cover
int augcl_internal_x2_prep_weights(const float *const *w, int n, int mode, void *out, float *zero, int n_zero,
                                   augcl_stream_t stream);
int augcl_internal_amax_f32(const float *x, size_t n, float *out, augcl_stream_t stream);
int augcl_internal_gate_bits(const float *x, void *bits, int B, int H, int W, augcl_stream_t stream);
int augcl_internal_x2_conv32_forward(const float *in, const float *bias, const float *bias2, float *out, int B, int Hi, int Wi,
                                     int relu, const void *wfrag, const void *wfrag2, const float *amax_in, float *amax_out,
                                     void *gate_bits_out, augcl_stream_t stream);
int augcl_internal_x2_conv32_backward_data(const float *dz, const float *act_prev, float *dact, int B, int Ho, int Wo,
                                           const void *wfrag, const float *amax_in, float *amax_out, const void *gate_bits,
                                           augcl_stream_t stream);
int augcl_internal_x2_conv32_backward_weights(const float *act_prev, const float *dz, float *dw, float *db, float *workspace,
                                              size_t workspace_bytes, int B, int Hi, int Wi, const float *amax_act,
                                              const float *amax_dz, augcl_stream_t stream);

// precision 2: the weight fragments of every 32 -> 32 layer are prepared by ONE small launch per pass into frag_ws (>=
// augcl_enc_frag_workspace_bytes(n_layers - 1, sets) bytes; null / too small: every layer launch prepares its own in its
// prologue -- ~6 us of a 20-90 us launch).  Returns the per-layer block size in bytes, or 0 when not used.
static size_t x3_prepare(const float *const *w, int n_layers, int mode, char *ws, size_t ws_bytes, size_t at, augcl_stream_t stream,
                         int *rc, bool f16_pairs = false, float *zero = nullptr, int n_zero = 0)
{
    *rc = AUGCL_OK;
    const size_t block = augcl_enc_frag_workspace_bytes(1, 1), need = at + (size_t)(n_layers - 1) * block;
    if (ws == nullptr || n_layers < 2 || n_layers > 33 || ws_bytes < need) return 0;
    *rc = f16_pairs ? augcl_internal_x2_prep_weights(w + 1, n_layers - 1, mode, ws + at, zero, n_zero, stream)
                    : augcl_internal_x3_prep_weights(w + 1, n_layers - 1, mode, ws + at, stream);
    return *rc == AUGCL_OK ? block : 0;
}

// precision 3 needs both of its side buffers: the prepared fragments and the per-tensor maxima
static int x2_require(const char *name, int n_layers, const void *frag_ws, size_t frag_ws_bytes, int sets, const float *amax)
{
    if (n_layers < 2) return AUGCL_OK;
    AUGCL_REQUIRE(n_layers <= 33, "%s: precision 3 covers at most 33 layers", name);
    AUGCL_REQUIRE(frag_ws != nullptr && frag_ws_bytes >= augcl_enc_frag_workspace_bytes(n_layers - 1, sets),
                  "%s: precision 3 needs frag_ws of augcl_enc_frag_workspace_bytes(n_layers - 1, %d) bytes", name, sets);
    AUGCL_REQUIRE(amax != nullptr, "%s: precision 3 needs the amax buffer", name);
    return AUGCL_OK;
}

// ---- single 32 -> 32 layers on the f16-pair kernels (public: include/augcl.h).  A lone layer has no producer that left its
// input's maximum, so the caller passes it (augcl_enc_amax_f32 computes one); the weight fragments are prepared per call.
int augcl_enc_amax_f32(const float *x, size_t n, float *amax_out, augcl_stream_t stream)
{
    const char *name = "augcl_enc_amax_f32";
    AUGCL_REQUIRE(x && amax_out, "%s: null tensor", name);
    hipError_t e = hipMemsetAsync(amax_out, 0, sizeof(float), (hipStream_t)stream);
    if (e != hipSuccess) return augcl::fail(AUGCL_E_LAUNCH, "%s: memset: %s", name, hipGetErrorString(e));
    return n ? augcl_internal_amax_f32(x, n, amax_out, stream) : AUGCL_OK;
}

static int x2_layer_prep(const char *name, const float *w, int mode, void *frag_ws, size_t frag_ws_bytes, float *amax_out,
                         augcl_stream_t stream)
{
    AUGCL_REQUIRE(frag_ws != nullptr && frag_ws_bytes >= augcl_enc_frag_workspace_bytes(1, 1),
                  "%s: frag_ws must hold augcl_enc_frag_workspace_bytes(1, 1) bytes", name);
    if (amax_out != nullptr) {
        hipError_t e = hipMemsetAsync(amax_out, 0, sizeof(float), (hipStream_t)stream);
        if (e != hipSuccess) return augcl::fail(AUGCL_E_LAUNCH, "%s: memset: %s", name, hipGetErrorString(e));
    }
    const float *wl[1] = {w};
    return augcl_internal_x2_prep_weights(wl, 1, mode, frag_ws, nullptr, 0, stream);
}

int augcl_enc_x2_conv32_forward(const float *in, const float *w, const float *bias, float *out, int B, int Hi, int Wi, int relu,
                                void *frag_ws, size_t frag_ws_bytes, const float *amax_in, float *amax_out,
                                augcl_stream_t stream)
{
    const char *name = "augcl_enc_x2_conv32_forward";
    AUGCL_REQUIRE(in && w && out && amax_in, "%s: null tensor", name);
    if (B == 0) return AUGCL_OK;
    int rc = x2_layer_prep(name, w, 0, frag_ws, frag_ws_bytes, amax_out, stream);
    if (rc != AUGCL_OK) return rc;
    rc = augcl_internal_x2_conv32_forward(in, bias, nullptr, out, B, Hi, Wi, relu, frag_ws, nullptr, amax_in, amax_out, nullptr, stream);
    if (rc <= 0) return rc;
    rc = augcl_enc_x3_conv32_forward(in, w, bias, out, B, Hi, Wi, relu, stream); // shape not covered: the split-bf16 kernel
    if (rc == AUGCL_OK && amax_out != nullptr) rc = augcl_internal_amax_f32(out, (size_t)B * 32 * (Hi - 2) * (Wi - 2), amax_out, stream);
    return rc;
}

int augcl_enc_x2_conv32_backward_data(const float *dz, const float *w, const float *act_prev, float *dact, int B, int Ho, int Wo,
                                      void *frag_ws, size_t frag_ws_bytes, const float *amax_dz, float *amax_out,
                                      augcl_stream_t stream)
{
    const char *name = "augcl_enc_x2_conv32_backward_data";
    AUGCL_REQUIRE(dz && w && dact && amax_dz, "%s: null tensor", name);
    if (B == 0) return AUGCL_OK;
    int rc = x2_layer_prep(name, w, 1, frag_ws, frag_ws_bytes, amax_out, stream);
    if (rc != AUGCL_OK) return rc;
    rc = augcl_internal_x2_conv32_backward_data(dz, act_prev, dact, B, Ho, Wo, frag_ws, amax_dz, amax_out, nullptr, stream);
    if (rc <= 0) return rc;
    rc = augcl_enc_x3_conv32_backward_data(dz, w, act_prev, dact, B, Ho, Wo, stream);
    if (rc == AUGCL_OK && amax_out != nullptr) rc = augcl_internal_amax_f32(dact, (size_t)B * 32 * (Ho + 2) * (Wo + 2), amax_out, stream);
    return rc;
}

int augcl_enc_x2_conv32_backward_weights(const float *act_prev, const float *dz, float *dw, float *db, float *workspace,
                                         size_t workspace_bytes, int B, int Hi, int Wi, const float *amax_act,
                                         const float *amax_dz, augcl_stream_t stream)
{
    const char *name = "augcl_enc_x2_conv32_backward_weights";
    AUGCL_REQUIRE(act_prev && dz && dw && db && workspace && amax_act && amax_dz, "%s: null tensor", name);
    const int rc = augcl_internal_x2_conv32_backward_weights(act_prev, dz, dw, db, workspace, workspace_bytes, B, Hi, Wi, amax_act,
                                                             amax_dz, stream);
    if (rc <= 0) return rc;
    return augcl_enc_x3_conv32_backward_weights(act_prev, dz, dw, db, workspace, workspace_bytes, B, Hi, Wi, stream);
}

// bf16 mode's gradient-carrying forward with the 32 -> 32 layers as chains (encoder_chain.hip): AUGCL_OK when it ran, +1 when the
// shape / workspace is not covered (the caller runs the per-layer kernels), negative on error.  Defined below the chain planner.
static int bf16_forward_chained(const float *x, const float *const *w, const float *const *b, void *const *acts_out, int n_layers,
                                int B, int C, int H, int W, int first_stride, float scale, void *frag_ws, size_t frag_ws_bytes,
                                augcl_stream_t stream);

// precision: 0 = f32 NCHW activations, f32 MFMA; 1 = bf16 NHWC activations (performance mode); 2 = f32 NCHW activations
// with the 32 -> 32 layers on the split-bf16 kernels (float32-accurate products on the bf16 pipe); 3 = the same tensors with
// the 32 -> 32 layers on the f16-pair kernels (float32-accurate, half the matrix instructions of 2: the parity path).
// acts_out[i] = output buffer of layer i; the last layer always writes f32 NCHW.  amax (precision 3): n_layers floats,
// amax[i] = max |output of layer i| on return (the backward pass reads them).
int augcl_enc_stack_forward(const float *x, const float *const *w, const float *const *b, void *const *acts_out,
                            int n_layers, int B, int C, int H, int W, int first_stride, float scale, int precision,
                            void *frag_ws, size_t frag_ws_bytes, float *amax, void *const *gate_bits, augcl_stream_t stream)
{
    const char *name = "augcl_enc_stack_forward";
    AUGCL_REQUIRE(x && w && b && acts_out && n_layers >= 1, "%s: bad argument", name);
    AUGCL_REQUIRE(precision >= 0 && precision <= 3, "%s: precision must be 0, 1, 2 or 3", name);
    AUGCL_REQUIRE(precision != 1 || n_layers >= 2, "%s: bf16 mode needs >= 2 layers", name);
    int h = H, ww = W, rc;
    if (precision == 1) { // the 32 -> 32 layers as LDS-resident chains that also store every layer's output for the backward pass
        rc = bf16_forward_chained(x, w, b, acts_out, n_layers, B, C, H, W, first_stride, scale, frag_ws, frag_ws_bytes, stream);
        if (rc <= 0) return rc;
    }
    const void *cur = x;
    size_t fblock = 0;
    const bool x2 = precision == 3 && n_layers >= 2;
    if (precision == 3 && (rc = x2_require(name, n_layers, frag_ws, frag_ws_bytes, 1, amax)) != AUGCL_OK) return rc;
    if (precision == 2 || x2) {
        fblock = x3_prepare(w, n_layers, 0, (char *)frag_ws, frag_ws_bytes, 0, stream, &rc, x2, x2 ? amax : nullptr, n_layers);
        if (rc != AUGCL_OK) return rc; // (precision 3: the same launch zeroes the maxima)
    }
    for (int i = 0; i < n_layers; ++i) {
        const int relu = i < n_layers - 1;
        const bool last = i == n_layers - 1;
        if (i == 0) {
            if (x2) { // the pipelined first-layer kernel leaves max |a_0| itself; any other shape: generic kernel + one reduction
                void *gb = gate_bits && relu ? gate_bits[0] : nullptr;
                rc = augcl::conv_first9_try(x, w[0], b[0], acts_out[0], B, C, h, ww, first_stride, scale, relu, false,
                                            (hipStream_t)stream, name, nullptr, nullptr, amax, gb);
                if (rc > 0) {
                    const int ho = out_hw(h, first_stride), wo = out_hw(ww, first_stride);
                    rc = augcl_enc_conv_first_forward(x, w[0], b[0], (float *)acts_out[0], B, C, h, ww, first_stride, scale, relu, stream);
                    if (rc == AUGCL_OK) rc = augcl_internal_amax_f32((const float *)acts_out[0], (size_t)B * 32 * ho * wo, amax, stream);
                    if (rc == AUGCL_OK && gb) rc = augcl_internal_gate_bits((const float *)acts_out[0], gb, B, ho, wo, stream);
                }
            } else if (precision != 1)
                rc = augcl_enc_conv_first_forward(x, w[0], b[0], (float *)acts_out[0], B, C, h, ww, first_stride, scale, relu,
                                                  stream);
            else
                rc = augcl_enc_bf16_conv_first_forward(x, w[0], b[0], acts_out[0], B, C, h, ww, first_stride, scale, relu,
                                                       stream);
            h = out_hw(h, first_stride);
            ww = out_hw(ww, first_stride);
        } else {
            if (precision == 0)
                rc = augcl_enc_conv32_forward((const float *)cur, w[i], b[i], (float *)acts_out[i], B, h, ww, relu, stream);
            else if (precision == 2)
                rc = augcl_internal_x3_conv32_forward((const float *)cur, w[i], b[i], nullptr, nullptr, (float *)acts_out[i], B, h, ww,
                                                      relu, fblock ? (char *)frag_ws + (size_t)(i - 1) * fblock : nullptr, nullptr, stream);
            else if (precision == 3) {
                void *gb = gate_bits && relu ? gate_bits[i] : nullptr;
                rc = augcl_internal_x2_conv32_forward((const float *)cur, b[i], nullptr, (float *)acts_out[i], B, h, ww, relu,
                                                      (char *)frag_ws + (size_t)(i - 1) * fblock, nullptr, amax + i - 1,
                                                      last ? nullptr : amax + i, gb, stream);
                if (rc > 0) { // a shape the f16-pair kernel does not cover: the split-bf16 kernel (or its own fallback) + reductions
                    rc = augcl_enc_x3_conv32_forward((const float *)cur, w[i], b[i], (float *)acts_out[i], B, h, ww, relu, stream);
                    if (rc == AUGCL_OK && !last)
                        rc = augcl_internal_amax_f32((const float *)acts_out[i], (size_t)B * 32 * (h - 2) * (ww - 2), amax + i, stream);
                    if (rc == AUGCL_OK && gb) rc = augcl_internal_gate_bits((const float *)acts_out[i], gb, B, h - 2, ww - 2, stream);
                }
            } else
                rc = augcl_enc_bf16_conv32_forward(cur, w[i], b[i], acts_out[i], B, h, ww, relu, last ? 1 : 0, stream);
            h -= 2;
            ww -= 2;
        }
        if (rc != AUGCL_OK) return rc;
        cur = acts_out[i];
    }
    return AUGCL_OK;
}

int augcl_internal_conv32_forward_dual(const float *in, const float *w, const float *bias, const float *w2,
                                       const float *bias2, float *out, int B2, int Hi, int Wi, int relu,
                                       augcl_stream_t stream);
int augcl_internal_bf16_conv32_forward_dual(const void *in_bf16_nhwc, const float *w, const float *bias, const float *w2,
                                            const float *bias2, void *out, int B2, int Hi, int Wi, int relu,
                                            int out_f32_nchw, augcl_stream_t stream);

// Two encoders with identical shapes over the SAME input batch, one launch per layer (no-grad passes: the actor's CNN and
// the target critic's CNN over next_obs in SAC.calculate_target_Q).  acts_out[i] holds 2B images: [0,B) from set A,
// [B,2B) from set B.  Even workgroups serve set A, odd ones set B (a workgroup keeps one weight set in registers).
// Returns AUGCL_E_UNSUPPORTED for shapes without a pipelined first-layer kernel: the caller then runs two passes.
int augcl_enc_stack_forward_dual(const float *x, const float *const *wA, const float *const *bA, const float *const *wB,
                                 const float *const *bB, void *const *acts_out, int n_layers, int B, int C, int H, int W,
                                 int first_stride, float scale, int precision, void *frag_ws, size_t frag_ws_bytes,
                                 float *amax /* precision 3: 2 x n_layers floats of scratch */, augcl_stream_t stream)
{
    const char *name = "augcl_enc_stack_forward_dual";
    AUGCL_REQUIRE(x && wA && bA && wB && bB && acts_out && n_layers >= 2 && B >= 1, "%s: bad argument", name);
    AUGCL_REQUIRE(precision >= 0 && precision <= 3, "%s: precision must be 0, 1, 2 or 3", name);
    const int B2 = 2 * B;
    int rc;
    const bool x2 = precision == 3;
    const size_t setB = (size_t)(n_layers - 1) * augcl_enc_frag_workspace_bytes(1, 1); // set B's blocks follow set A's
    size_t fblock = 0;
    if (x2) { // amax[2 i + set] = max |output of layer i| of that weight set's half; zeroed by the first preparation launch
        if ((rc = x2_require(name, n_layers, frag_ws, frag_ws_bytes, 2, amax)) != AUGCL_OK) return rc;
        fblock = x3_prepare(wA, n_layers, 0, (char *)frag_ws, frag_ws_bytes, 0, stream, &rc, true, amax, 2 * n_layers);
        if (rc != AUGCL_OK) return rc;
        x3_prepare(wB, n_layers, 0, (char *)frag_ws, frag_ws_bytes, setB, stream, &rc, true);
        if (rc != AUGCL_OK) return rc;
    }
    rc = augcl::conv_first9_try(x, wA[0], bA[0], acts_out[0], B2, C, H, W, first_stride, scale, 1, precision == 1,
                                (hipStream_t)stream, name, wB[0], bB[0], x2 ? amax : nullptr);
    if (rc > 0) return augcl::fail(AUGCL_E_UNSUPPORTED, "%s: first-layer shape (%d,%d,%d) stride %d not covered", name, C, H, W, first_stride);
    if (rc != AUGCL_OK) return rc;
    int h = out_hw(H, first_stride), ww = out_hw(W, first_stride);
    if (precision == 2 && frag_ws_bytes >= 2 * setB) {
        fblock = x3_prepare(wA, n_layers, 0, (char *)frag_ws, frag_ws_bytes, 0, stream, &rc);
        if (rc != AUGCL_OK) return rc;
        if (fblock) {
            x3_prepare(wB, n_layers, 0, (char *)frag_ws, frag_ws_bytes, setB, stream, &rc);
            if (rc != AUGCL_OK) return rc;
        }
    }
    for (int i = 1; i < n_layers; ++i) {
        const int relu = i < n_layers - 1;
        if (precision == 0)
            rc = augcl_internal_conv32_forward_dual((const float *)acts_out[i - 1], wA[i], bA[i], wB[i], bB[i],
                                                    (float *)acts_out[i], B2, h, ww, relu, stream);
        else if (precision == 2)
            rc = augcl_internal_x3_conv32_forward((const float *)acts_out[i - 1], wA[i], bA[i], wB[i], bB[i], (float *)acts_out[i], B2,
                                                  h, ww, relu, fblock ? (char *)frag_ws + (size_t)(i - 1) * fblock : nullptr,
                                                  fblock ? (char *)frag_ws + setB + (size_t)(i - 1) * fblock : nullptr, stream);
        else if (x2) {
            const bool last = i == n_layers - 1;
            rc = augcl_internal_x2_conv32_forward((const float *)acts_out[i - 1], bA[i], bB[i], (float *)acts_out[i], B2, h, ww, relu,
                                                  (char *)frag_ws + (size_t)(i - 1) * fblock,
                                                  (char *)frag_ws + setB + (size_t)(i - 1) * fblock, amax + 2 * (i - 1),
                                                  last ? nullptr : amax + 2 * i, nullptr, stream);
            if (rc > 0) { // shape not covered: the split-bf16 kernel, then the two halves' maxima
                rc = augcl_internal_x3_conv32_forward((const float *)acts_out[i - 1], wA[i], bA[i], wB[i], bB[i], (float *)acts_out[i],
                                                      B2, h, ww, relu, nullptr, nullptr, stream);
                const size_t half = (size_t)B * 32 * (h - 2) * (ww - 2);
                if (rc == AUGCL_OK && !last) rc = augcl_internal_amax_f32((const float *)acts_out[i], half, amax + 2 * i, stream);
                if (rc == AUGCL_OK && !last)
                    rc = augcl_internal_amax_f32((const float *)acts_out[i] + half, half, amax + 2 * i + 1, stream);
            }
        } else
            rc = augcl_internal_bf16_conv32_forward_dual(acts_out[i - 1], wA[i], bA[i], wB[i], bB[i], acts_out[i], B2, h, ww,
                                                         relu, i == n_layers - 1 ? 1 : 0, stream);
        if (rc != AUGCL_OK) return rc;
        h -= 2;
        ww -= 2;
    }
    return AUGCL_OK;
}

// ---- no-grad forward with the 32 -> 32 layers in CHAINS (encoder_chain.hip): the first layer's kernel, one weight-fragment
// launch per weight set, then one launch per chain of up to six layers whose intermediates stay in LDS.
int augcl_internal_chain_rounds(int Hi, int Wi, int nl, int parts);
int augcl_internal_chain_bf16_rounds(int Hi, int Wi, int nl, int parts);
int augcl_internal_bf16_prep_weights(const float *const *w, int n, void *out, int stride, augcl_stream_t stream);
int augcl_internal_bf16_prep_weights_mode(const float *const *w, int n, void *out, int stride, int mode, augcl_stream_t stream);
int augcl_internal_chain_bf16_backward(const void *dz_top, const void *const *wfrag, const void *const *mask, void *const *dz_out,
                                       int nl, int B, int Ht, int Wt, augcl_stream_t stream);
int augcl_internal_chain_bf16_forward(const void *in_bf16_nhwc, void *out, const void *const *wfragA, const float *const *biasA,
                                      const void *const *wfragB, const float *const *biasB, int nl, int B, int Hi, int Wi, int parts,
                                      int relu_last, int out_f32_nchw, void *const *acts_out, augcl_stream_t stream);
int augcl_internal_chain_forward(const float *in, float *out, const void *const *wfragA, const float *const *biasA,
                                 const void *const *wfragB, const float *const *biasB, int nl, int B, int Hi, int Wi, int parts,
                                 int relu_last, const float *amax_in, float *amax_out, augcl_stream_t stream);

namespace {

struct ChainPlan {
    int n = 0;
    int nl[32], parts[32];
};

// Fewest estimated microseconds over all ways to cut n32 layers (input h x w, `images` of them, `cus` workgroups at a time) into
// chains: a launch costs ceil(items / cus) x (the longest item's tile rounds + ~2 for its staging) + ~8 for ramp and drain.
// Small batches come out with many parts per image (more workgroups busy), large ones with the fewest that fit LDS.
bool plan_chains(int h, int w, int n32, long images, int cus, ChainPlan &plan, int (*rounds_fn)(int, int, int, int), int max_l)
{
    if (n32 < 1 || n32 > 32) return false;
    double best[33];
    int bl[33], bp[33];
    best[n32] = 0.0;
    for (int s = n32 - 1; s >= 0; --s) {
        best[s] = 1e30;
        bl[s] = bp[s] = 0;
        const int hs = h - 2 * s, ws = w - 2 * s;
        for (int L = 1; L <= max_l && s + L <= n32; ++L) {
            const int hf = hs - 2 * L;
            for (int P = 1; P <= hf; ++P) {
                const int r = rounds_fn(hs, ws, L, P);
                if (r < 0) continue;
                const double cost = (double)((images * P + cus - 1) / cus) * (r + 2) + 8.0 + best[s + L];
                if (cost < best[s]) { best[s] = cost; bl[s] = L; bp[s] = P; }
            }
        }
        if (bl[s] == 0) return false;
    }
    plan.n = 0;
    for (int s = 0; s < n32; s += bl[s]) {
        plan.nl[plan.n] = bl[s];
        plan.parts[plan.n] = bp[s];
        ++plan.n;
    }
    return true;
}

int cu_count_stack()
{
    int dev = 0, v = 256;
    if (hipGetDevice(&dev) == hipSuccess) {
        int n = 0;
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0) v = n;
    }
    return v;
}

} // namespace

static int bf16_forward_chained(const float *x, const float *const *w, const float *const *b, void *const *acts_out, int n_layers,
                                int B, int C, int H, int W, int first_stride, float scale, void *frag_ws, size_t frag_ws_bytes,
                                augcl_stream_t stream)
{
    const char *name = "augcl_enc_stack_forward";
    if (n_layers < 2 || n_layers > 33 || frag_ws == nullptr || frag_ws_bytes < augcl_enc_frag_workspace_bytes(n_layers - 1, 1)) return 1;
    const int h1 = out_hw(H, first_stride), w1 = out_hw(W, first_stride);
    ChainPlan plan;
    if (!plan_chains(h1, w1, n_layers - 1, B, cu_count_stack(), plan, augcl_internal_chain_bf16_rounds, 12)) return 1;
    const size_t block = augcl_enc_frag_workspace_bytes(1, 1);
    int rc = augcl_internal_bf16_prep_weights(w + 1, n_layers - 1, frag_ws, (int)block, stream);
    if (rc != AUGCL_OK) return rc;
    rc = augcl_enc_bf16_conv_first_forward(x, w[0], b[0], acts_out[0], B, C, H, W, first_stride, scale, 1, stream);
    if (rc != AUGCL_OK) return rc;
    int layer = 1, h = h1, ww = w1;
    for (int c = 0; c < plan.n; ++c) {
        const int nl = plan.nl[c];
        const bool final_chain = c == plan.n - 1;
        const void *fa[12];
        for (int j = 0; j < nl; ++j) fa[j] = (char *)frag_ws + (size_t)(layer + j - 1) * block;
        // acts_out[layer + j] = output of the chain's layer j; its last layer's goes through the kernel's `out`
        rc = augcl_internal_chain_bf16_forward(acts_out[layer - 1], acts_out[layer + nl - 1], fa, b + layer, nullptr, nullptr, nl, B, h, ww,
                                               plan.parts[c], final_chain ? 0 : 1, final_chain ? 1 : 0, acts_out + layer, stream);
        if (rc > 0) return augcl::fail(AUGCL_E_UNSUPPORTED, "%s: chain of %d layers at %d x %d not covered", name, nl, h, ww);
        if (rc != AUGCL_OK) return rc;
        layer += nl;
        h -= 2 * nl;
        ww -= 2 * nl;
    }
    return AUGCL_OK;
}

int augcl_internal_chain_forward_keep(const float *in, float *out, const void *const *wfrag, const float *const *bias, int nl, int B,
                                      int Hi, int Wi, int parts, int relu_last, const float *amax_in, float *amax_out,
                                      void *const *acts_out, void *const *bits_out, float *amax_acts, augcl_stream_t stream);

// The GRADIENT-CARRYING forward of precision 3 with the 32 -> 32 layers as chains (encoder_chain.hip, k_conv32_chain<true>): the
// critic's pass over the (augmented) batch in SAC.update_critic (src/algorithms/sac.py:103-116 -> modules.py:145-165).  Same
// tensors on return as augcl_enc_stack_forward(precision 3) -- every layer's float32 output in acts_out[i], amax[i], the gate
// words -- but a layer reads its input from LDS, where the previous layer left it as f16 pairs, instead of from HBM: the
// activations are WRITTEN once and not read back (the per-layer kernels read each one once), 4 launches instead of 12.  Values:
// the chains' (per-item scales; the accuracy class of the per-layer kernels, not their bits).  AUGCL_E_UNSUPPORTED: a shape the
// chains do not cover -- the caller runs augcl_enc_stack_forward.
int augcl_enc_stack_forward_chained(const float *x, const float *const *w, const float *const *b, void *const *acts_out,
                                    int n_layers, int B, int C, int H, int W, int first_stride, float scale, void *frag_ws,
                                    size_t frag_ws_bytes, float *amax, void *const *gate_bits, augcl_stream_t stream)
{
    const char *name = "augcl_enc_stack_forward_chained";
    AUGCL_REQUIRE(x && w && b && acts_out && n_layers >= 1 && B >= 0, "%s: bad argument", name);
    if (n_layers < 2 || n_layers > 33) return augcl::fail(AUGCL_E_UNSUPPORTED, "%s: 2..33 layers", name);
    if (B == 0) return AUGCL_OK;
    int rc;
    if ((rc = x2_require(name, n_layers, frag_ws, frag_ws_bytes, 1, amax)) != AUGCL_OK) return rc;
    const int h1 = out_hw(H, first_stride), w1 = out_hw(W, first_stride);
    AUGCL_REQUIRE(h1 >= 1 && w1 >= 1 && h1 - 2 * (n_layers - 1) >= 1 && w1 - 2 * (n_layers - 1) >= 1, "%s: image too small", name);
    ChainPlan plan;
    if (!plan_chains(h1, w1, n_layers - 1, B, cu_count_stack(), plan, augcl_internal_chain_rounds, 6))
        return augcl::fail(AUGCL_E_UNSUPPORTED, "%s: no chain plan for a %d x %d input", name, h1, w1);
    // fragments with unscaled residuals (mode 2); the same launch zeroes the maxima
    rc = augcl_internal_x2_prep_weights(w + 1, n_layers - 1, 2, frag_ws, amax, n_layers, stream);
    if (rc != AUGCL_OK) return rc;
    void *gb0 = gate_bits ? gate_bits[0] : nullptr;
    rc = augcl::conv_first9_try(x, w[0], b[0], acts_out[0], B, C, H, W, first_stride, scale, 1, false, (hipStream_t)stream, name, nullptr,
                                nullptr, amax, gb0);
    if (rc > 0) {
        rc = augcl_enc_conv_first_forward(x, w[0], b[0], (float *)acts_out[0], B, C, H, W, first_stride, scale, 1, stream);
        if (rc == AUGCL_OK) rc = augcl_internal_amax_f32((const float *)acts_out[0], (size_t)B * 32 * h1 * w1, amax, stream);
        if (rc == AUGCL_OK && gb0) rc = augcl_internal_gate_bits((const float *)acts_out[0], gb0, B, h1, w1, stream);
    }
    if (rc != AUGCL_OK) return rc;
    const size_t block = augcl_enc_frag_workspace_bytes(1, 1);
    int layer = 1, h = h1, ww = w1;
    for (int c = 0; c < plan.n; ++c) {
        const int nl = plan.nl[c];
        const bool final_chain = c == plan.n - 1;
        const void *fa[6];
        for (int j = 0; j < nl; ++j) fa[j] = (char *)frag_ws + (size_t)(layer + j - 1) * block;
        // stack layer (layer + j) = the chain's layer j; its last layer's output goes through the kernel's `out`
        rc = augcl_internal_chain_forward_keep((const float *)acts_out[layer - 1], (float *)acts_out[layer + nl - 1], fa, b + layer, nl, B, h,
                                               ww, plan.parts[c], final_chain ? 0 : 1, amax + layer - 1,
                                               final_chain ? nullptr : amax + layer + nl - 1, acts_out + layer,
                                               gate_bits ? gate_bits + layer : nullptr, amax + layer, stream);
        if (rc > 0) return augcl::fail(AUGCL_E_UNSUPPORTED, "%s: chain of %d layers at %d x %d not covered", name, nl, h, ww);
        if (rc != AUGCL_OK) return rc;
        layer += nl;
        h -= 2 * nl;
        ww -= 2 * nl;
    }
    return AUGCL_OK;
}

size_t augcl_enc_stack_backward_chain_bytes(int n_layers, int B, int C, int H, int W, int first_stride)
{
    (void)C;
    if (n_layers < 2 || B < 1 || H < 3 || W < 3 || first_stride < 1) return 0;
    size_t tot = 0;
    int h = out_hw(H, first_stride), w = out_hw(W, first_stride); // output of layer 0 = input of layer 1
    for (int i = 1; i <= n_layers; ++i) { // inputs of layers 1 .. n - 1 (the steps' outputs) and the last layer's output (the top gradient)
        if (h < 1 || w < 1) return 0;
        tot += (((size_t)B * h * w * 64) + 255) & ~(size_t)255;
        h -= 2;
        w -= 2;
    }
    return tot;
}

size_t augcl_enc_stack_infer_scratch_bytes(int n_layers, int B_total, int C, int H, int W, int first_stride)
{
    (void)n_layers; (void)C;
    if (B_total < 1 || H < 3 || W < 3 || first_stride < 1) return 0;
    const size_t a0 = (size_t)B_total * 32 * out_hw(H, first_stride) * out_hw(W, first_stride) * sizeof(float);
    return 2 * ((a0 + 255) & ~(size_t)255); // two buffers of the largest activation: the chains ping-pong between them
}

// x (B,C,H,W) raw f32; out ((wB ? 2 : 1) x B, 32, h, w) f32 NCHW: rows [0,B) from weight set A, [B,2B) from set B over the SAME
// input.  amax: 2 x (n_layers + 1) floats of device scratch.  AUGCL_E_UNSUPPORTED: a shape the chains (or, with two weight
// sets, the pipelined first-layer kernel) do not cover -- the caller runs augcl_enc_stack_forward[_dual].
static int stack_infer_impl(const char *name, const float *x, const uint8_t *x8, const float *const *wA, const float *const *bA,
                            const float *const *wB, const float *const *bB, float *out, void *scratch, size_t scratch_bytes,
                            int n_layers, int B, int C, int H, int W, int first_stride, float scale, int precision, void *frag_ws,
                            size_t frag_ws_bytes, float *amax, augcl_stream_t stream)
{
    AUGCL_REQUIRE((x || x8) && wA && bA && out && n_layers >= 1 && B >= 1, "%s: bad argument", name);
    AUGCL_REQUIRE(precision == 3 || precision == 1, "%s: precision must be 3 (f16 pairs) or 1 (bf16 mode)", name);
    if (x8 && (precision != 3 || n_layers < 2))
        return augcl::fail(AUGCL_E_UNSUPPORTED, "%s: uint8 observations are built for precision 3 stacks of >= 2 layers", name);
    if (precision == 1) {
        // ---- bf16 performance mode: first layer (bf16 NHWC out), one fragment launch per weight set, the 32 -> 32 layers as chains
        AUGCL_REQUIRE((wB == nullptr) == (bB == nullptr), "%s: weight set B needs both arrays", name);
        AUGCL_REQUIRE(n_layers >= 2 && n_layers <= 33, "%s: bf16 mode needs 2..33 layers", name);
        const int sets = wB ? 2 : 1, Bt = sets * B;
        const int h1 = out_hw(H, first_stride), w1 = out_hw(W, first_stride);
        AUGCL_REQUIRE(h1 >= 1 && w1 >= 1 && h1 - 2 * (n_layers - 1) >= 1 && w1 - 2 * (n_layers - 1) >= 1, "%s: image too small", name);
        ChainPlan plan;
        if (!plan_chains(h1, w1, n_layers - 1, Bt, cu_count_stack(), plan, augcl_internal_chain_bf16_rounds, 12))
            return augcl::fail(AUGCL_E_UNSUPPORTED, "%s: no chain plan for a %d x %d input", name, h1, w1);
        const size_t block = augcl_enc_frag_workspace_bytes(1, 1), setB = (size_t)(n_layers - 1) * block;
        AUGCL_REQUIRE(frag_ws != nullptr && frag_ws_bytes >= augcl_enc_frag_workspace_bytes(n_layers - 1, sets),
                      "%s: frag_ws of augcl_enc_frag_workspace_bytes(n_layers - 1, %d) bytes needed", name, sets);
        AUGCL_REQUIRE(scratch != nullptr && scratch_bytes >= augcl_enc_stack_infer_scratch_bytes(n_layers, Bt, C, H, W, first_stride),
                      "%s: scratch smaller than augcl_enc_stack_infer_scratch_bytes", name);
        const size_t half = augcl_enc_stack_infer_scratch_bytes(n_layers, Bt, C, H, W, first_stride) / 2;
        void *buf[2] = {scratch, reinterpret_cast<char *>(scratch) + half};
        int rc;
        if (wB && 2 * (n_layers - 1) <= 32) { // both sets in ONE launch: set B's blocks follow set A's in frag_ws
            const float *wl[32];
            for (int i = 1; i < n_layers; ++i) {
                wl[i - 1] = wA[i];
                wl[n_layers - 1 + i - 1] = wB[i];
            }
            rc = augcl_internal_bf16_prep_weights(wl, 2 * (n_layers - 1), frag_ws, (int)block, stream);
        } else {
            rc = augcl_internal_bf16_prep_weights(wA + 1, n_layers - 1, frag_ws, (int)block, stream);
            if (rc == AUGCL_OK && wB) rc = augcl_internal_bf16_prep_weights(wB + 1, n_layers - 1, (char *)frag_ws + setB, (int)block, stream);
        }
        if (rc != AUGCL_OK) return rc;
        rc = augcl::conv_first9_try(x, wA[0], bA[0], buf[0], Bt, C, H, W, first_stride, scale, 1, true, (hipStream_t)stream, name,
                                    wB ? wB[0] : nullptr, wB ? bB[0] : nullptr);
        if (rc > 0) {
            if (wB) return augcl::fail(AUGCL_E_UNSUPPORTED, "%s: first-layer shape (%d,%d,%d) stride %d not covered", name, C, H, W, first_stride);
            rc = augcl_enc_bf16_conv_first_forward(x, wA[0], bA[0], buf[0], B, C, H, W, first_stride, scale, 1, stream);
        }
        if (rc != AUGCL_OK) return rc;
        int layer = 1, h = h1, w = w1, cur = 0;
        for (int c = 0; c < plan.n; ++c) {
            const int nl = plan.nl[c];
            const bool final_chain = c == plan.n - 1;
            const void *fa[12], *fb[12];
            for (int j = 0; j < nl; ++j) {
                fa[j] = (char *)frag_ws + (size_t)(layer + j - 1) * block;
                fb[j] = (char *)frag_ws + setB + (size_t)(layer + j - 1) * block;
            }
            rc = augcl_internal_chain_bf16_forward(buf[cur], final_chain ? (void *)out : buf[cur ^ 1], fa, bA + layer, wB ? fb : nullptr,
                                                   wB ? bB + layer : nullptr, nl, Bt, h, w, plan.parts[c], final_chain ? 0 : 1,
                                                   final_chain ? 1 : 0, nullptr, stream);
            if (rc > 0) return augcl::fail(AUGCL_E_UNSUPPORTED, "%s: chain of %d layers at %d x %d not covered", name, nl, h, w);
            if (rc != AUGCL_OK) return rc;
            layer += nl;
            h -= 2 * nl;
            w -= 2 * nl;
            cur ^= 1;
        }
        return AUGCL_OK;
    }
    AUGCL_REQUIRE(amax != nullptr, "%s: precision 3 needs the amax scratch", name);
    AUGCL_REQUIRE((wB == nullptr) == (bB == nullptr), "%s: weight set B needs both arrays", name);
    const int sets = wB ? 2 : 1, Bt = sets * B;
    const int h1 = out_hw(H, first_stride), w1 = out_hw(W, first_stride);
    AUGCL_REQUIRE(h1 >= 1 && w1 >= 1 && h1 - 2 * (n_layers - 1) >= 1 && w1 - 2 * (n_layers - 1) >= 1, "%s: image too small", name);
    int rc;
    ChainPlan plan;
    if (n_layers >= 2) {
        AUGCL_REQUIRE(n_layers <= 33, "%s: at most 33 layers", name);
        if (!plan_chains(h1, w1, n_layers - 1, Bt, cu_count_stack(), plan, augcl_internal_chain_rounds, 6))
            return augcl::fail(AUGCL_E_UNSUPPORTED, "%s: no chain plan for a %d x %d input", name, h1, w1);
        if ((rc = x2_require(name, n_layers, frag_ws, frag_ws_bytes, sets, amax)) != AUGCL_OK) return rc;
        AUGCL_REQUIRE(scratch != nullptr && scratch_bytes >= augcl_enc_stack_infer_scratch_bytes(n_layers, Bt, C, H, W, first_stride),
                      "%s: scratch smaller than augcl_enc_stack_infer_scratch_bytes", name);
    }
    const size_t half = n_layers >= 2 ? augcl_enc_stack_infer_scratch_bytes(n_layers, Bt, C, H, W, first_stride) / 2 : 0;
    float *buf[2] = {reinterpret_cast<float *>(scratch), reinterpret_cast<float *>(reinterpret_cast<char *>(scratch) + half)};
    // fragments of both sets (the first launch also zeroes the maxima: amax[2 c + set] = max |input of chain c| of that half)
    const size_t block = augcl_enc_frag_workspace_bytes(1, 1), setB = (size_t)(n_layers - 1) * block;
    if (n_layers >= 2) {
        if (wB && 2 * (n_layers - 1) <= 32) { // both sets in ONE launch: set B's blocks follow set A's in frag_ws
            const float *wl[32];
            for (int i = 1; i < n_layers; ++i) {
                wl[i - 1] = wA[i];
                wl[n_layers - 1 + i - 1] = wB[i];
            }
            rc = augcl_internal_x2_prep_weights(wl, 2 * (n_layers - 1), 2, frag_ws, amax, 2 * (n_layers + 1), stream);
            if (rc != AUGCL_OK) return rc;
        } else {
            x3_prepare(wA, n_layers, 2, (char *)frag_ws, frag_ws_bytes, 0, stream, &rc, true, amax, 2 * (n_layers + 1));
            if (rc != AUGCL_OK) return rc;
            if (wB) {
                x3_prepare(wB, n_layers, 2, (char *)frag_ws, frag_ws_bytes, setB, stream, &rc, true);
                if (rc != AUGCL_OK) return rc;
            }
        }
    }
    // first layer (f32 MFMA, stride 2, x / scale fused): leaves max |a_0| per half
    float *a0 = n_layers >= 2 ? buf[0] : out;
    const int relu0 = n_layers >= 2;
    if (x8) { // uint8 observations: a quarter of the input bytes, exact bf16 operands (encoder_first_u8.hip)
        rc = augcl::conv_first9_u8_try(x8, wA[0], bA[0], a0, B, C, H, W, first_stride, scale, relu0, (hipStream_t)stream, name,
                                       wB ? wB[0] : nullptr, wB ? bB[0] : nullptr, amax);
        if (rc > 0) return augcl::fail(AUGCL_E_UNSUPPORTED, "%s: uint8 first layer covers (9,84,84) stride 2, got (%d,%d,%d) stride %d", name, C, H, W, first_stride);
    } else
        rc = augcl::conv_first9_try(x, wA[0], bA[0], a0, Bt, C, H, W, first_stride, scale, relu0, false, (hipStream_t)stream, name,
                                    wB ? wB[0] : nullptr, wB ? bB[0] : nullptr, n_layers >= 2 ? amax : nullptr);
    if (rc > 0) {
        if (wB) return augcl::fail(AUGCL_E_UNSUPPORTED, "%s: first-layer shape (%d,%d,%d) stride %d not covered", name, C, H, W, first_stride);
        rc = augcl_enc_conv_first_forward(x, wA[0], bA[0], a0, B, C, H, W, first_stride, scale, relu0, stream);
        if (rc == AUGCL_OK && n_layers >= 2) rc = augcl_internal_amax_f32(a0, (size_t)B * 32 * h1 * w1, amax, stream);
    }
    if (rc != AUGCL_OK || n_layers < 2) return rc;
    int layer = 1, h = h1, w = w1, cur = 0;
    for (int c = 0; c < plan.n; ++c) {
        const int nl = plan.nl[c];
        const bool final_chain = c == plan.n - 1;
        const void *fa[6], *fb[6];
        for (int j = 0; j < nl; ++j) {
            fa[j] = (char *)frag_ws + (size_t)(layer + j - 1) * block;
            fb[j] = (char *)frag_ws + setB + (size_t)(layer + j - 1) * block;
        }
        float *dst = final_chain ? out : buf[cur ^ 1];
        rc = augcl_internal_chain_forward(buf[cur], dst, fa, bA + layer, wB ? fb : nullptr, wB ? bB + layer : nullptr, nl, Bt, h, w,
                                          plan.parts[c], final_chain ? 0 : 1, amax + 2 * c, final_chain ? nullptr : amax + 2 * (c + 1),
                                          stream);
        if (rc > 0) return augcl::fail(AUGCL_E_UNSUPPORTED, "%s: chain of %d layers at %d x %d not covered", name, nl, h, w);
        if (rc != AUGCL_OK) return rc;
        layer += nl;
        h -= 2 * nl;
        w -= 2 * nl;
        cur ^= 1;
    }
    return AUGCL_OK;
}

int augcl_enc_stack_infer(const float *x, const float *const *wA, const float *const *bA, const float *const *wB,
                          const float *const *bB, float *out, void *scratch, size_t scratch_bytes, int n_layers, int B, int C,
                          int H, int W, int first_stride, float scale, int precision, void *frag_ws, size_t frag_ws_bytes,
                          float *amax, augcl_stream_t stream)
{
    AUGCL_REQUIRE(x != nullptr, "augcl_enc_stack_infer: bad argument");
    return stack_infer_impl("augcl_enc_stack_infer", x, nullptr, wA, bA, wB, bB, out, scratch, scratch_bytes, n_layers, B, C, H, W,
                            first_stride, scale, precision, frag_ws, frag_ws_bytes, amax, stream);
}

int augcl_enc_stack_infer_u8(const uint8_t *x, const float *const *wA, const float *const *bA, const float *const *wB,
                             const float *const *bB, float *out, void *scratch, size_t scratch_bytes, int n_layers, int B, int C,
                             int H, int W, int first_stride, float scale, int precision, void *frag_ws, size_t frag_ws_bytes,
                             float *amax, augcl_stream_t stream)
{
    AUGCL_REQUIRE(x != nullptr, "augcl_enc_stack_infer_u8: bad argument");
    return stack_infer_impl("augcl_enc_stack_infer_u8", nullptr, x, wA, bA, wB, bB, out, scratch, scratch_bytes, n_layers, B, C, H, W,
                            first_stride, scale, precision, frag_ws, frag_ws_bytes, amax, stream);
}

// acts_in[i] = input of layer i (acts_in[0] = x raw f32; i >= 1: post-ReLU activation, f32 NCHW or bf16 NHWC);
// grad_out = f32 NCHW gradient of the last layer's output; dz_a / dz_b = two scratch buffers, each large enough for
// the largest activation (layer 0's output) in the mode's element type; dw[i] / db[i] receive the gradients.
int augcl_enc_stack_backward(const void *const *acts_in, const float *const *w, const float *grad_out, float *const *dw,
                             float *const *db, void *dz_a, void *dz_b, float *ws32, size_t ws32_bytes, float *ws_first,
                             size_t ws_first_bytes, int n_layers, int B, int C, int H, int W, int first_stride,
                             float scale, int precision, int bf16_wgrad_mfma, void *frag_ws, size_t frag_ws_bytes,
                             const float *amax_acts /* precision 3: the forward pass's amax */,
                             float *amax_dz /* precision 3: n_layers floats of scratch */,
                             const void *const *gate_bits /* precision 3: the forward pass's gate bits, or null */,
                             void *dz_chain /* precision 1: augcl_enc_stack_backward_chain_bytes(...) of scratch, or null */,
                             size_t dz_chain_bytes, augcl_stream_t stream)
{
    const char *name = "augcl_enc_stack_backward";
    AUGCL_REQUIRE(acts_in && w && grad_out && dw && db && dz_a && dz_b && n_layers >= 1, "augcl_enc_stack_backward: bad argument");
    AUGCL_REQUIRE(precision >= 0 && precision <= 3, "%s: precision must be 0, 1, 2 or 3", name);
    int hs[64], wsz[64]; // input size of every layer
    AUGCL_REQUIRE(n_layers <= 63, "augcl_enc_stack_backward: too many layers");
    hs[0] = H; wsz[0] = W;
    hs[1] = out_hw(H, first_stride); wsz[1] = out_hw(W, first_stride);
    for (int i = 2; i <= n_layers; ++i) { hs[i] = hs[i - 1] - 2; wsz[i] = wsz[i - 1] - 2; }
    int rc;
    // With a workspace slice per 32-channel layer (ws32_bytes >= (n_layers-1) x the per-layer size) every layer keeps
    // its slabs until the end and ALL of them are reduced by one batched pair of launches (2 instead of 2 per layer).
    const size_t per_layer = augcl_enc_wgrad_workspace_bytes(32);
    const bool batched = n_layers >= 2 && ws32_bytes >= per_layer * (size_t)(n_layers - 1);
    struct Scope { // the deferral must not outlive this call, whatever path returns
        bool on;
        ~Scope() { if (on) augcl::reduce_defer_abort(); }
    } scope{batched};
    if (batched) augcl::reduce_defer_begin();
    size_t fblock = 0;
    const bool x2 = precision == 3 && n_layers >= 2;
    if (x2) {
        if ((rc = x2_require(name, n_layers, frag_ws, frag_ws_bytes, 1, amax_acts)) != AUGCL_OK) return rc;
        AUGCL_REQUIRE(amax_dz != nullptr, "%s: precision 3 needs the amax_dz scratch", name);
    }
    if (precision == 2 || x2) { // backward-data fragments (transposed weights, flipped taps) of every 32 -> 32 layer: one launch
        fblock = x3_prepare(w, n_layers, 1, (char *)frag_ws, frag_ws_bytes, 0, stream, &rc, x2, x2 ? amax_dz : nullptr, n_layers);
        if (rc != AUGCL_OK) return rc;
    }
    if (x2) { // amax_dz[i] = max |gradient w.r.t. the output of layer i| (zeroed by the preparation launch): the top one by a
              // reduction, the others by the kernels
        rc = augcl_internal_amax_f32(grad_out, (size_t)B * 32 * hs[n_layers] * wsz[n_layers], amax_dz + n_layers - 1, stream);
        if (rc != AUGCL_OK) return rc;
    }
    const void *dz = grad_out; // f32 NCHW (B,32,hs[n],ws[n])
    if (precision == 1 && n_layers >= 2 && n_layers <= 13 && dz_chain != nullptr && frag_ws != nullptr &&
        frag_ws_bytes >= augcl_enc_frag_workspace_bytes(n_layers - 1, 1) &&
        dz_chain_bytes >= augcl_enc_stack_backward_chain_bytes(n_layers, B, C, H, W, first_stride)) {
        // bf16 mode: backward-data of ALL 32 -> 32 layers as one launch on an LDS-resident canvas (encoder_chain.hip), every
        // layer's gradient stored once; then the backward-weights launches read them.  dz_chain: [dZ of the last layer's output |
        // step 0's output | step 1's | ...], each bf16 NHWC, 256-byte aligned.
        const size_t block = augcl_enc_frag_workspace_bytes(1, 1);
        const int nl = n_layers - 1;
        const float *wrev[32];
        for (int s_ = 0; s_ < nl; ++s_) wrev[s_] = w[n_layers - 1 - s_];
        rc = augcl_internal_bf16_prep_weights_mode(wrev, nl, frag_ws, (int)block, 1, stream);
        if (rc != AUGCL_OK) return rc;
        char *p = reinterpret_cast<char *>(dz_chain);
        auto take = [&](int h_, int w_) {
            char *q = p;
            p += (((size_t)B * h_ * w_ * 64) + 255) & ~(size_t)255;
            return (void *)q;
        };
        void *dz_top = take(hs[n_layers], wsz[n_layers]);
        rc = augcl_enc_nchw_f32_to_nhwc_bf16(grad_out, dz_top, B, hs[n_layers], wsz[n_layers], stream);
        if (rc != AUGCL_OK) return rc;
        const void *fr[12], *mk[12];
        void *dzo[12];
        for (int s_ = 0; s_ < nl; ++s_) {
            const int i = n_layers - 1 - s_; // the layer this step differentiates: its input acts_in[i] is hs[i] x wsz[i]
            fr[s_] = (char *)frag_ws + (size_t)s_ * block;
            mk[s_] = acts_in[i];
            dzo[s_] = take(hs[i], wsz[i]);
        }
        rc = augcl_internal_chain_bf16_backward(dz_top, fr, mk, dzo, nl, B, hs[n_layers], wsz[n_layers], stream);
        if (rc < 0) return rc;
        if (rc == 0) {
            for (int i = n_layers - 1; i >= 1; --i) { // gradient w.r.t. the output of layer i: the top one, else step (n - 2 - i)'s output
                const void *dzi = i == n_layers - 1 ? dz_top : dzo[n_layers - 2 - i];
                float *ws_i = batched ? ws32 + (size_t)(i - 1) * (per_layer / sizeof(float)) : ws32;
                const size_t ws_i_bytes = batched ? per_layer : ws32_bytes;
                rc = bf16_wgrad_mfma ? augcl_enc_bf16_conv32_backward_weights_mfma(acts_in[i], dzi, dw[i], db[i], ws_i, ws_i_bytes, B, hs[i], wsz[i], stream)
                                     : augcl_enc_bf16_conv32_backward_weights(acts_in[i], dzi, dw[i], db[i], ws_i, ws_i_bytes, B, hs[i], wsz[i], stream);
                if (rc != AUGCL_OK) return rc;
            }
            rc = augcl_enc_bf16_conv_first_backward_weights((const float *)acts_in[0], dzo[nl - 1], dw[0], db[0], ws_first, ws_first_bytes, B,
                                                            C, H, W, first_stride, scale, stream);
            if (batched) {
                scope.on = false;
                const int rc2 = augcl::reduce_defer_flush((hipStream_t)stream, name);
                if (rc == AUGCL_OK) rc = rc2;
            }
            return rc;
        } // (+1: the canvas does not fit LDS -- the per-layer kernels below)
    }
    if (precision == 1) {
        rc = augcl_enc_nchw_f32_to_nhwc_bf16(grad_out, dz_a, B, hs[n_layers], wsz[n_layers], stream);
        if (rc != AUGCL_OK) return rc;
        dz = dz_a;
    }
    void *nxt = (dz == dz_a) ? dz_b : dz_a;
    for (int i = n_layers - 1; i >= 1; --i) {
        float *ws_i = batched ? ws32 + (size_t)(i - 1) * (per_layer / sizeof(float)) : ws32;
        const size_t ws_i_bytes = batched ? per_layer : ws32_bytes;
        if (precision == 0)
            rc = augcl_enc_conv32_backward_weights((const float *)acts_in[i], (const float *)dz, dw[i], db[i], ws_i, ws_i_bytes, B,
                                                   hs[i], wsz[i], stream);
        else if (precision == 2)
            rc = augcl_enc_x3_conv32_backward_weights((const float *)acts_in[i], (const float *)dz, dw[i], db[i], ws_i, ws_i_bytes,
                                                      B, hs[i], wsz[i], stream);
        else if (precision == 3) {
            rc = augcl_internal_x2_conv32_backward_weights((const float *)acts_in[i], (const float *)dz, dw[i], db[i], ws_i,
                                                           ws_i_bytes, B, hs[i], wsz[i], amax_acts + i - 1, amax_dz + i, stream);
            if (rc > 0)
                rc = augcl_enc_x3_conv32_backward_weights((const float *)acts_in[i], (const float *)dz, dw[i], db[i], ws_i,
                                                          ws_i_bytes, B, hs[i], wsz[i], stream);
        } else if (bf16_wgrad_mfma)
            rc = augcl_enc_bf16_conv32_backward_weights_mfma(acts_in[i], dz, dw[i], db[i], ws_i, ws_i_bytes, B, hs[i], wsz[i],
                                                             stream);
        else
            rc = augcl_enc_bf16_conv32_backward_weights(acts_in[i], dz, dw[i], db[i], ws_i, ws_i_bytes, B, hs[i], wsz[i], stream);
        if (rc != AUGCL_OK) return rc;
        if (precision == 0)
            rc = augcl_enc_conv32_backward_data((const float *)dz, w[i], (const float *)acts_in[i], (float *)nxt, B, hs[i + 1],
                                                wsz[i + 1], stream);
        else if (precision == 2)
            rc = augcl_internal_x3_conv32_backward_data((const float *)dz, w[i], (const float *)acts_in[i], (float *)nxt, B, hs[i + 1],
                                                        wsz[i + 1], fblock ? (char *)frag_ws + (size_t)(i - 1) * fblock : nullptr, stream);
        else if (precision == 3) {
            // (the gradient that reaches the first layer's backward-weights needs no maximum: that kernel reads float32)
            // gate: the bit map the forward pass left for acts_in[i] (= the output of layer i - 1), else acts_in[i] itself
            rc = augcl_internal_x2_conv32_backward_data((const float *)dz, (const float *)acts_in[i], (float *)nxt, B, hs[i + 1],
                                                        wsz[i + 1], (char *)frag_ws + (size_t)(i - 1) * fblock, amax_dz + i,
                                                        i > 1 ? amax_dz + i - 1 : nullptr, gate_bits ? gate_bits[i - 1] : nullptr,
                                                        stream);
            if (rc > 0) {
                rc = augcl_enc_x3_conv32_backward_data((const float *)dz, w[i], (const float *)acts_in[i], (float *)nxt, B, hs[i + 1],
                                                       wsz[i + 1], stream);
                if (rc == AUGCL_OK && i > 1)
                    rc = augcl_internal_amax_f32((const float *)nxt, (size_t)B * 32 * hs[i] * wsz[i], amax_dz + i - 1, stream);
            }
        } else
            rc = augcl_enc_bf16_conv32_backward_data(dz, w[i], acts_in[i], nxt, B, hs[i + 1], wsz[i + 1], stream);
        if (rc != AUGCL_OK) return rc;
        dz = nxt;
        nxt = (nxt == dz_a) ? dz_b : dz_a;
    }
    if (precision != 1)
        rc = augcl_enc_conv_first_backward_weights((const float *)acts_in[0], (const float *)dz, dw[0], db[0], ws_first,
                                                   ws_first_bytes, B, C, H, W, first_stride, scale, stream);
    else
        rc = augcl_enc_bf16_conv_first_backward_weights((const float *)acts_in[0], dz, dw[0], db[0], ws_first, ws_first_bytes, B,
                                                        C, H, W, first_stride, scale, stream);
    if (batched) {
        scope.on = false;
        const int rc2 = augcl::reduce_defer_flush((hipStream_t)stream, "augcl_enc_stack_backward");
        if (rc == AUGCL_OK) rc = rc2;
    }
    return rc;
}

} // extern "C"
