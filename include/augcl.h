/*
 * include/augcl.h -- C ABI of libaugcl_hip.so, the MI355X (gfx950) implementation of
 * AugCL's data-parallel hot path: pixel augmentations on stacked 84x84 frames, the
 * shared conv encoder, and the fused optimiser / target-network kernels of the
 * SAC / SVEA / DrQ / AugCL update().
 *
 * The reference (GT-RIPL/AugCL) has no FFI layer: its boundary is the Python call
 * surface `augmentations.<fn>(x, **params)` / `algorithms.<algo>.update(...)`.
 * Each entry point below names the reference function (file:line under
 * /root/reference) whose device work it replaces; augcl_amd/ (Python) is the only
 * caller and keeps the reference's names and argument meaning.
 *
 * Conventions
 *  - every pointer is a DEVICE pointer borrowed from the caller (torch tensors);
 *    the library never allocates, frees or retains user buffers;
 *  - every call is asynchronous on `stream` (a hipStream_t passed as void*;
 *    NULL = the null stream) and safe to capture in a hipGraph: no hipMalloc,
 *    no synchronisation, no host<->device copies inside;
 *  - return value: AUGCL_OK or a negative AUGCL_E_* code; augcl_last_error()
 *    returns a thread-local message for the last failure on this thread;
 *  - no global state; re-entrant.
 *  - tensors are contiguous NCHW; "frames" = C/3 RGB frames per observation.
 *  - random parameters come in two forms: EXPLICIT (device arrays; what the parity
 *    tests use) or RNG-INTERNAL (pass NULL for the array: the kernel derives the
 *    parameters from Philox4x32-10 keyed by (seed, global sample index), so results
 *    do not depend on how a batch is sharded over ranks).  augcl_philox_* export
 *    the same streams so a checker can recover the parameters that were used.
 */
#ifndef AUGCL_H
#define AUGCL_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define AUGCL_ABI_VERSION 6 /* 2: the whole-stack entry points take a weight-fragment workspace; 3: ... and the per-tensor maxima of precision 3;
                             * 4: augcl_enc_stack_infer (no-grad forward with LDS-resident layer chains); 5: uint8 observations into the first layer
                             * (augcl_enc_stack_infer_u8, augcl_enc_conv_first_forward_u8); 6: augcl_enc_stack_forward_chained (the
                             * gradient-carrying forward of precision 3 with LDS-resident layer chains) */

#define AUGCL_OK 0
#define AUGCL_E_INVALID (-1)   /* bad argument (null pointer, bad shape, misalignment) */
#define AUGCL_E_LAUNCH (-2)    /* HIP reported a launch/runtime error */
#define AUGCL_E_UNSUPPORTED (-3)

typedef void *augcl_stream_t; /* hipStream_t */

int augcl_abi_version(void);
const char *augcl_last_error(void);
/* number of HIP devices visible, or a negative error: lets a host verify the GPU
 * path exists before relying on it (the Python shim raises if this fails). */
int augcl_device_count(void);

/* ---- RNG parameter streams (Philox4x32-10; counter = (sample, slot, offset_lo, offset_hi), key = seed) */
int augcl_philox_shift_offsets(uint64_t seed, uint64_t offset, int sample0, int B, int pad,
                               int32_t *oy, int32_t *ox, augcl_stream_t stream);
int augcl_philox_conv_weights(uint64_t seed, uint64_t offset, int sample0, int B,
                              float *weights /* B x 81 */, augcl_stream_t stream);
int augcl_philox_normal(uint64_t seed, uint64_t offset, size_t n, float *out, augcl_stream_t stream);
/* replay sampling indices drawn ON the device (ReplayBuffer.sample's np.random.randint(0, n_valid, B), src/utils.py:229-232,
 * without the host draw and its H2D copy): idx[b] = floor(u32 * n_valid / 2^32), u32 from slot 32 of sample (sample0 + b). */
int augcl_philox_replay_indices(uint64_t seed, uint64_t offset, int sample0, int n_valid, int64_t *idx, int B,
                                augcl_stream_t stream);

/* ---- random_shift: src/augmentations.py:378-382 (F.pad replicate + kornia RandomCrop((h,w)))
 * out[b,c,y,x] = in[b,c,clamp(y+oy[b]-pad,0,H-1),clamp(x+ox[b]-pad,0,W-1)], oy/ox in [0,2*pad].
 * oy/ox: device int32[B], or both NULL => Philox (seed, offset, sample0 + b).
 * _u8 / _u8_f32 read the replay buffer's uint8 frames directly (fusing the `.float()` of
 * src/utils.py:208-209 for _u8_f32).  Integer index op: bit-exact. */
int augcl_random_shift_f32(const float *in, float *out, int B, int C, int H, int W, int pad,
                           const int32_t *oy, const int32_t *ox,
                           uint64_t seed, uint64_t offset, int sample0, augcl_stream_t stream);
int augcl_random_shift_u8(const uint8_t *in, uint8_t *out, int B, int C, int H, int W, int pad,
                          const int32_t *oy, const int32_t *ox,
                          uint64_t seed, uint64_t offset, int sample0, augcl_stream_t stream);
int augcl_random_shift_u8_f32(const uint8_t *in, float *out, int B, int C, int H, int W, int pad,
                              const int32_t *oy, const int32_t *ox,
                              uint64_t seed, uint64_t offset, int sample0, augcl_stream_t stream);

/* ---- random_conv: src/augmentations.py:265-274
 * per sample b, W_b = weights[b] laid out [o][i][ky][kx] (3,3,3,3): every RGB frame f of the
 * sample: out = sigmoid(conv2d(replicate_pad1(in/255.0), W_b)) * 255.0.
 * weights: device float[B*81], or NULL => Philox normals (seed, offset, sample0 + b). */
int augcl_random_conv_f32(const float *in, float *out, int B, int frames, int H, int W,
                          const float *weights,
                          uint64_t seed, uint64_t offset, int sample0, augcl_stream_t stream);

/* ---- random_overlay: src/augmentations.py:253-262 (arithmetic at L262; the image batch comes
 * from _get_data_batch L76-86, `.repeat(1, C//3, 1, 1)` L260)
 * out = ((1-alpha)*(in/255.0) + alpha*overlay[b]) * 255.0, overlay (B,3,H,W) in [0,1],
 * the same image for every frame of a sample.  Literal op order: bit-exact vs torch f32. */
int augcl_random_overlay_f32(const float *in, const float *overlay, float *out,
                             int B, int frames, int H, int W, float alpha, augcl_stream_t stream);

/* ---- color_jitter: src/augmentations.py:225-237 (+ get_jitter_model L15-16)
 * In place.  idx: device int32[n_idx] indices into the (B*frames,3,H,W) view (duplicates
 * allowed, as np.random.randint draws them); order[4]: torchvision op ids in application
 * order (0 brightness=disabled, 1 contrast, 2 saturation, 3 hue), HOST array; factors are
 * the python floats torchvision would draw.  bound = clamp ceiling of _blend (1.0 = the
 * literal float path the reference runs on its 0..255 floats); scale: frame is divided by
 * scale before and multiplied after (1 = literal/compat, 255 = documented "sane" mode). */
int augcl_color_jitter_f32(float *inout, int B, int frames, int H, int W,
                           const int32_t *idx, int n_idx, const int32_t *order_host,
                           double contrast, double saturation, double hue,
                           float bound, float scale, augcl_stream_t stream);

/* ---- rest of the augmentation registry (src/augmentations.py:436-449), all index/select operators, bit-exact:
 * random_crop (L385-433: out[b,c,y,x] = in[b,c,w1[b]+y,h1[b]+x] -- the reference's w1 indexes ROWS), random_flip
 * (L139-168), random_rotation (L171-213) share ONE per-sample integer-affine gather:
 *   out[b,c,y,x] = in[b,c, ay*y+by*x+cy, ax*y+bx*x+cx], coef = device int32[B][6] = {ay,by,cy,ax,bx,cx}.
 * random_cutout_color (L327-350): box rows h1..2*h1-1, cols w1..2*w1-1 of every channel <- color[b][c];
 * box = device int32[B][2] = {h1,w1}.   random_grayscale (L89-136), literal: t = uint8(x*255) with C-style wrap,
 * gray = uint8(.2989r+.587g+.114b), out = (mask[b] ? gray : t)/255. */
int augcl_affine_gather_f32(const float *in, float *out, int B, int C, int Hin, int Win, int Hout, int Wout,
                            const int32_t *coef, augcl_stream_t stream);
int augcl_cutout_color_f32(const float *in, float *out, int B, int C, int H, int W, const int32_t *box,
                           const float *color, augcl_stream_t stream);
int augcl_random_grayscale_f32(const float *in, float *out, int B, int frames, int H, int W, const uint8_t *mask,
                               augcl_stream_t stream);
/* splice (cs_mode 0, src/augmentations.py:298-309) / CS_splice (cs_mode 1, L312-324) + create_hsv_mask (L277-285):
 * HSV-keyed replacement of x/255 by an overlay image; frame j of the (B*frames,3,H,W) view pairs with imgs[j % B]
 * (the reference's batch-dim tiling).  HSV per kornia.color.rgb_to_hsv (hue in radians, s = d/(v+eps)).
 * x, out: (B, 3*frames, H, W) f32; imgs: (B,3,H,W) f32 in [0,1]. */
int augcl_splice_f32(const float *x, const float *imgs, float *out, int B, int frames, int H, int W, int cs_mode,
                     float hue_thres, float sat_thres, float val_thres, float eps, augcl_stream_t stream);

/* ---- eval side: do_green_screen (src/env/wrappers.py:209-260), the video_easy / video_hard background keying the
 * reference runs as a per-pixel Python loop: x, bg, out = (N,3,H,W) uint8; a pixel whose HSV (double arithmetic,
 * colorsys formulas) lies in h*360 in [100,185], s*255 in [80,255], v*255 in [70,255] takes the bg pixel. */
int augcl_green_screen_u8(const uint8_t *x, const uint8_t *bg, uint8_t *out, int N, int H, int W, augcl_stream_t stream);
/* VideoWrapper._interpolate_bg (src/env/wrappers.py:336-340): (N,3,Hb,Wb) uint8 video frames -> (N,3,H,W) uint8 through
 * float/255, bilinear (align_corners=False), *255, truncating cast -- ATen's float32 arithmetic operation by operation
 * (uint8-exact vs the reference's function).  _resize_: VideoWrapper._greenscreen (L342-348) = that resize + the keying
 * above in one launch: bg_raw is the raw video frame, only keyed-in pixels are interpolated. */
int augcl_interpolate_bg_u8(const uint8_t *bg, uint8_t *out, int N, int Hb, int Wb, int H, int W, augcl_stream_t stream);
int augcl_green_screen_resize_u8(const uint8_t *x, const uint8_t *bg_raw, uint8_t *out, int N, int H, int W, int Hb, int Wb,
                                 augcl_stream_t stream);

/* ---- fused SVEA/AugCL front end: src/algorithms/svea.py:68 + :49 (random_shift, then
 * apply_aug=random_conv on the shifted clone) in one pass over the uint8/f32 observation:
 * out_shift = random_shift(in); out_aug = random_conv(out_shift).  Either output may be NULL. */
int augcl_shift_conv_u8_f32(const uint8_t *in, float *out_shift, float *out_aug,
                            int B, int frames, int H, int W, int pad,
                            const int32_t *oy, const int32_t *ox, const float *weights,
                            uint64_t seed, uint64_t offset, int sample0, augcl_stream_t stream);
int augcl_shift_conv_f32(const float *in, float *out_shift, float *out_aug,
                         int B, int frames, int H, int W, int pad,
                         const int32_t *oy, const int32_t *ox, const float *weights,
                         uint64_t seed, uint64_t offset, int sample0, augcl_stream_t stream);

/* ---- ReplayBuffer.sample -> random_shift -> random_conv in ONE pass over the replay buffer's uint8 frame ring
 * (src/utils.py:206-214, 234-240 + src/algorithms/svea.py:68 + :49): sample b's `frames` RGB frames are
 * ring[slot_tab[idx[b]][f]] (slot_tab: device int32 [capacity][tab_cols], first `frames` columns = the obs stack;
 * idx: device int64[B], the np.random.randint draw), each 3*H*W bytes.  63.5 KB read per observation instead of
 * 63.5 KB read + 254 KB written + 254 KB read again.  Same outputs / parameters as augcl_shift_conv_u8_f32 on the
 * gathered batch (bit-identical). */
int augcl_replay_shift_conv_u8_f32(const uint8_t *ring, const int32_t *slot_tab, const int64_t *idx, int tab_cols,
                                   float *out_shift, float *out_aug, int B, int frames, int H, int W, int pad,
                                   const int32_t *oy, const int32_t *ox, const float *weights,
                                   uint64_t seed, uint64_t offset, int sample0, augcl_stream_t stream);

/* ---- fused optimiser step over a FLAT parameter group: torch.optim.Adam(betas=(b1,.999), eps=1e-8)
 * as used at src/algorithms/sac.py:48-56 (math of torch/optim/adam.py, single-tensor form):
 *   g *= grad_scale; m = b1*m+(1-b1)*g; v = b2*v+(1-b2)*g*g; p -= (lr/bc1) * m/(sqrt(v)/sqrt(bc2)+eps)
 * bias corrections use `step` (1-based).  grad_scale folds the 1/world_size of a data-parallel
 * SUM all-reduce into the same pass.  All four buffers hold n floats, 16-byte aligned. */
int augcl_adam_f32(float *param, const float *grad, float *exp_avg, float *exp_avg_sq, size_t n,
                   double lr, double beta1, double beta2, double eps, int64_t step, double grad_scale,
                   augcl_stream_t stream);
/* Same step with the step count kept ON THE DEVICE: step_state = 16 bytes {int64 step; float scratch[2]}, 8-byte
 * aligned, zero-initialised by the caller.  advance != 0: the call increments the count in stream order and derives the
 * bias corrections from it (a 1-thread launch in front of the step); advance == 0: augcl_gather_f32(step_state = ...)
 * has already done so for this step.  Either way no host scalar is baked in: capturable in a hipGraph. */
int augcl_adam_f32_devstep(float *param, const float *grad, float *exp_avg, float *exp_avg_sq, size_t n,
                           double lr, double beta1, double beta2, double eps, void *step_state, int advance,
                           double grad_scale, augcl_stream_t stream);
/* Gradient gather in front of the flat Adam step (what loss.backward() + the optimiser's per-parameter loop do to
 * p.grad in the reference, src/algorithms/sac.py:114-116): tensor i (numel[i] contiguous floats at src[i], 4-byte
 * aligned) is copied to dst + dst_offset[i]; one launch per 96 tensors, the table passed by value (host arrays are read
 * during the call only).  step_state != NULL: the launch also advances that optimiser's device-side step (see above). */
int augcl_gather_f32(const void *const *src, const int64_t *dst_offset, const int64_t *numel, int n_tensors, float *dst,
                     void *step_state, double lr, double beta1, double beta2, augcl_stream_t stream);

/* ---- SAC heads: the element-wise tails of the policy / critic heads as single launches.
 * Squashed-Gaussian policy head = Actor.forward after its MLP (src/algorithms/modules.py:225-244) with
 * gaussian_logprob (L19-22) and squash (L25-32): raw (B,2A) = [mu | log_std_raw];
 *   log_std = min + 0.5 (max-min) (tanh(raw_ls) + 1); pi = mu + noise exp(log_std);
 *   log_pi = sum(-0.5 noise^2 - log_std) - 0.5 log(2 pi) A - sum log(relu(1 - tanh(pi)^2) + 1e-6);
 * outputs tanh(mu), tanh(pi), log_pi (B,1), log_std.  noise = NULL: compute_pi = False (mu_t, log_std only).
 * backward: gradients of the four outputs (NULL = none) -> g_raw (B,2A); recomputes from raw / noise.  g_pi_ld = row
 * stride of g_pi_t in floats (A when contiguous; it is usually a column slice of the critic's input gradient). */
int augcl_gauss_head_forward(const float *raw, const float *noise, float *mu_t, float *pi_t, float *log_pi,
                             float *log_std, int B, int A, double log_std_min, double log_std_max,
                             augcl_stream_t stream);
int augcl_gauss_head_backward(const float *raw, const float *noise, const float *g_mu_t, const float *g_pi_t,
                              int g_pi_ld, const float *g_log_pi, const float *g_log_std, float *g_raw, int B, int A,
                              double log_std_min, double log_std_max, augcl_stream_t stream);
/* SAC.calculate_target_Q (src/algorithms/sac.py:94-101):
 *   out = reward + (not_done * discount) * (min(tq1,tq2) - alpha * log_pi); alpha read from a device double. */
int augcl_td_target_f32(const float *tq1, const float *tq2, const float *log_pi, const float *reward,
                        const float *not_done, const double *alpha_dev /* float64 scalar on the device */,
                        int alpha_is_log /* 1: alpha_dev holds log_alpha (SAC.alpha = log_alpha.exp()) */,
                        double discount, float *out, int B, augcl_stream_t stream);
/* actor loss (src/algorithms/sac.py:119-131): out[0] = mean_i(alpha * log_pi_i - min(q1_i, q2_i)), alpha =
 * float(exp(*log_alpha)); backward: d_log_pi = g alpha / B, -g/B to the smaller of (q1_i, q2_i) (ties: half each). */
int augcl_actor_loss_forward(const float *log_pi, const float *q1, const float *q2, const double *log_alpha, float *out,
                             int B, augcl_stream_t stream);
int augcl_actor_loss_backward(const float *q1, const float *q2, const double *log_alpha, const float *g, float *d_log_pi,
                              float *dq1, float *dq2, int B, augcl_stream_t stream);
/* temperature update (sac.py:137-146) in one launch: alpha_loss = mean(alpha * (-log_pi - target_entropy)), its gradient
 * with respect to the float64 log_alpha and torch.optim.Adam's step on that scalar, all on the device.
 * state = double[4] {exp_avg, exp_avg_sq, step, grad}; mode bit 0 = loss + gradient into state[3], bit 1 = Adam step
 * from state[3] (data parallel: average state[3] over ranks in between); out (may be NULL) = {alpha_loss, alpha}. */
int augcl_alpha_step_f64(double *log_alpha, const float *log_pi, int B, double target_entropy, double lr, double beta1,
                         double beta2, double eps, double *state, float *out, int mode, augcl_stream_t stream);
/* The 3-layer MLP trunks -- QFunction.trunk (src/algorithms/modules.py:247-261; Critic's Q1 / Q2, :264-276, run as T = 2
 * trunks side by side) and Actor.mlp (:200-215) -- around their library GEMMs.  Twin strides are in floats (the twin's
 * tensors sit at fixed distances inside the flat parameter buffer).
 * forward: z (T,B,H) = a1 W2^T as the batched GEMM left it (no bias); in place a2 = relu(z + b2[t]);
 *   q[t*B + b] = a2[t][b] . w3[t] + b3[t]  (fixed-order sum per row).
 * backward: g0 / g1 (B) = dL/dq of trunk 0 / 1; dz2[t][b][c] = a2 > 0 ? g w3[t][c] : 0; optional (NULL = skip)
 *   dw3 (T,H) = sum_b g a2, db3 (T) = sum_b g, db2 (T,H) = sum_b dz2, all fixed-order. */
int augcl_mlp_out_forward(float *z, const float *b2, int64_t b2_twin_stride, const float *w3, int64_t w3_twin_stride,
                          const float *b3, int64_t b3_twin_stride, float *q, int T, int B, int H, augcl_stream_t stream);
int augcl_mlp_out_backward(const float *g0, const float *g1, const float *a2, const float *w3, int64_t w3_twin_stride,
                           float *dz2, float *dw3, float *db3, float *db2, int T, int B, int H, augcl_stream_t stream);
/* ReLU gate + bias gradient of a hidden Linear in one pass (torch's threshold_backward + sum(0)):
 *   dz (B,N) <- a > 0 ? dz : 0 in place; db (N, may be NULL) = column sums of the gated dz; extra (B,E) with E > 0:
 *   dextra (E) = its plain column sums (the output layer's bias gradient rides along). */
int augcl_mask_colsum(float *dz, const float *a, float *db, const float *extra, float *dextra, int B, int N, int E,
                      augcl_stream_t stream);
/* Twin-Q loss (sac.py:108-110; svea.py:37-56 with S = 2 segments of one 2B batch against the same M targets):
 *   seg[s] = mse(q1[sM:(s+1)M], y) + mse(q2[...], y); out = {sum_s w[s] seg[s], seg[0..S)}; S <= 4; w on the host.
 * One workgroup, fixed-order tree sum (deterministic).  backward: dq = g_total * w[s] * 2 (q - y) / M. */
int augcl_q_loss_forward(const float *q1, const float *q2, const float *y, int M, int S, const double *w, float *out,
                         augcl_stream_t stream);
int augcl_q_loss_backward(const float *q1, const float *q2, const float *y, const float *g_total, int M, int S,
                          const double *w, float *dq1, float *dq2, augcl_stream_t stream);

/* RLProjection's tail (src/algorithms/modules.py:101-111): y = tanh(LayerNorm(x; eps) * gamma + beta) on (B,D) rows,
 * D <= 256, and its backward (dx, dgamma, dbeta; workspace = 2*B*D floats; fixed-order sums); gy_ld = row stride of
 * gy in floats (D when contiguous). */
int augcl_ln_tanh_forward(const float *x, const float *gamma, const float *beta, float *y, int B, int D, double eps,
                          augcl_stream_t stream);
int augcl_ln_tanh_backward(const float *x, const float *gamma, const float *y, const float *gy, int gy_ld, float *dx,
                           float *dgamma, float *dbeta,
                           float *dbias /* may be NULL: column sums of dx = the bias gradient of the Linear in front */,
                           float *workspace /* 2*B*D floats */, int B, int D, double eps, augcl_stream_t stream);
/* RLProjection forward as a whole (src/algorithms/modules.py:101-111): y = x W^T + bias on the f32 MFMA pipe (split-K, every
 * partial sum added in a fixed order: bitwise reproducible), z = tanh(LayerNorm(y) * gamma + beta).  x (M,K), w (N,K)
 * row-major, N <= 128, K a multiple of 32 with K/32 divisible by one of {63,49,32,21,16,9,8,7} (14112 = 32*21*21 is).
 * workspace: slabs(M,K,N) * M * N floats; augcl_proj_ln_tanh_workspace_slabs returns 0 for an uncovered shape. */
int augcl_proj_ln_tanh_workspace_slabs(int M, int K, int N);
int augcl_proj_ln_tanh_forward(const float *x, const float *w, const float *bias, const float *gamma, const float *beta,
                               float *y, float *z, float *workspace, size_t workspace_bytes, int M, int K, int N, double eps,
                               augcl_stream_t stream);
/* The same with the critic's input row built on the way out: xcat (M, N + A) = [z | tail row (m % tail_rows)], what
 * QFunction.forward's torch.cat([obs, action], dim=1) produces (src/algorithms/modules.py:258-261; SVEA's 2B-row critic
 * pass repeats the B actions, svea.py:44-45).  N + A <= 128, M a multiple of tail_rows. */
int augcl_proj_ln_tanh_cat_forward(const float *x, const float *w, const float *bias, const float *gamma, const float *beta,
                                   float *y, float *z, float *xcat, const float *tail, int A, int tail_rows, float *workspace,
                                   size_t workspace_bytes, int M, int K, int N, double eps, augcl_stream_t stream);
/* RLProjection backward, input side: dx (M,K) = dy (M,N) w (N,K), the gradient that enters the conv stack (what autograd's
 * Linear backward computes for src/algorithms/modules.py:104); N a multiple of 4, <= 128; dy 16-byte aligned. */
int augcl_proj_dx_f32(const float *dy, const float *w, float *dx, int M, int N, int K, augcl_stream_t stream);

/* ---- target-network update: utils.soft_update_params (src/utils.py:76-78), called from
 * SAC.soft_update_critic_target (src/algorithms/sac.py:148-153):
 *   target[i] = tau*src[i] + (1-tau)*target[i]   (separate IEEE ops, bit-exact vs torch f32) */
int augcl_ema_f32(float *target, const float *src, size_t n, double tau, augcl_stream_t stream);
/* The two segments SAC.soft_update_critic_target updates (Q1 + Q2 with critic_tau, the encoder with encoder_tau) in one
 * launch; each segment exactly as augcl_ema_f32 (n0 or n1 may be 0). */
int augcl_ema2_f32(float *target0, const float *src0, size_t n0, double tau0, float *target1, const float *src1, size_t n1,
                   double tau1, augcl_stream_t stream);

/* ---- shared conv encoder: SharedCNN / HeadCNN of src/algorithms/modules.py:145-181, i.e.
 * NormalizeImg (x/255) -> Conv2d(C,32,3,stride=2) -> [ReLU -> Conv2d(32,32,3)]*, and its backward.
 * All tensors f32 NCHW.  Forward kernels fuse the bias and (relu != 0) the ReLU that FOLLOWS the
 * conv, so stored activations are post-ReLU; backward-data fuses the ReLU mask of the layer below
 * (act_prev > 0; NULL = no mask).  f32 MFMA (v_mfma_f32_32x32x2_f32): exact f32 FMA chains.
 * backward-weights needs a workspace of augcl_enc_wgrad_workspace_bytes(C_in) bytes; its reduction
 * order is fixed (no atomics), so gradients are bitwise reproducible. */
int augcl_enc_conv_first_forward(const float *in /* (B,C,Hi,Wi) */, const float *w /* (32,C,3,3) */,
                                 const float *bias, float *out /* (B,32,Ho,Wo) */, int B, int C, int Hi, int Wi,
                                 int stride, float scale /* input divided by this: 255 */, int relu,
                                 augcl_stream_t stream);
int augcl_enc_conv32_forward(const float *in /* (B,32,Hi,Wi) */, const float *w /* (32,32,3,3) */,
                             const float *bias, float *out /* (B,32,Hi-2,Wi-2) */, int B, int Hi, int Wi, int relu,
                             augcl_stream_t stream);
int augcl_enc_conv32_backward_data(const float *dz /* (B,32,Ho,Wo) */, const float *w, const float *act_prev,
                                   float *dact /* (B,32,Ho+2,Wo+2) */, int B, int Ho, int Wo, augcl_stream_t stream);
size_t augcl_enc_wgrad_workspace_bytes(int C_in);
int augcl_enc_conv32_backward_weights(const float *act_prev /* (B,32,Hi,Wi) */, const float *dz /* (B,32,Hi-2,Wi-2) */,
                                      float *dw /* (32,32,3,3) */, float *db /* (32) */, float *workspace,
                                      size_t workspace_bytes, int B, int Hi, int Wi, augcl_stream_t stream);
int augcl_enc_conv_first_backward_weights(const float *x /* (B,C,Hi,Wi) raw */, const float *dz, float *dw /* (32,C,3,3) */,
                                          float *db, float *workspace, size_t workspace_bytes, int B, int C, int Hi,
                                          int Wi, int stride, float scale, augcl_stream_t stream);

/* ---- the same three 32 -> 32 layer operators with FLOAT32-ACCURATE products on the bf16 matrix pipe (the default
 * parity path since round 3; replaces the f32-MFMA form of src/algorithms/modules.py:145-165's Conv2d(32,32,3)).
 * Same tensors, dtypes (f32 NCHW) and semantics as augcl_enc_conv32_*: every f32 operand is split exactly into three
 * bf16 parts (x = hi + mid + lo) while it is staged into LDS, and the six cross products of weight >= 2^-16 run on
 * v_mfma_f32_32x32x16_bf16 with f32 accumulation -- 2.6e-7 relative L2 against float64 at K = 288 (3.0e-7 for the
 * native f32 MFMA chain) at 2.67x its matrix rate.  Not bit-identical to augcl_enc_conv32_* (different summation
 * tree), both within float32 rounding of the exact result.  Inputs with |x| >= 2^127.99 (bf16 rounds them to
 * infinity) give NaN.  Row widths the split tiles do not fit run the augcl_enc_conv32_* kernel. */
int augcl_enc_x3_conv32_forward(const float *in /* (B,32,Hi,Wi) */, const float *w /* (32,32,3,3) */, const float *bias,
                                float *out /* (B,32,Hi-2,Wi-2) */, int B, int Hi, int Wi, int relu, augcl_stream_t stream);
int augcl_enc_x3_conv32_backward_data(const float *dz /* (B,32,Ho,Wo) */, const float *w, const float *act_prev,
                                      float *dact /* (B,32,Ho+2,Wo+2) */, int B, int Ho, int Wo, augcl_stream_t stream);
int augcl_enc_x3_conv32_backward_weights(const float *act_prev /* (B,32,Hi,Wi) */, const float *dz /* (B,32,Hi-2,Wi-2) */,
                                         float *dw /* (32,32,3,3) */, float *db /* (32) */, float *workspace,
                                         size_t workspace_bytes, int B, int Hi, int Wi, augcl_stream_t stream);

/* ---- ... and as F16 PAIRS (the default parity path inside the whole-stack entry points below, precision 3): every operand
 * x s = hi + lo' 2^-11 with hi, lo' in f16 and s a per-tensor power of two that brings max |x| into [2^14, 2^15); the three
 * significant cross products run on v_mfma_f32_32x32x16_f16 with f32 accumulation -- 1.4e-7 relative L2 against float64 at
 * K = 288, half the matrix instructions of augcl_enc_x3_*.  The scale needs max |x| of each input tensor: inside a stack every
 * kernel leaves the maximum of what it stores; a single layer takes them as arguments (device floats, exact maxima or upper
 * bounds within 2x) and can return its output's (amax_out, may be NULL; overwritten, not accumulated).
 * frag_ws: augcl_enc_frag_workspace_bytes(1, 1) bytes of device scratch, 16-byte aligned (the layer's weight fragments).
 * Elements smaller than 2^-26 of their tensor's maximum lose relative accuracy (2^-21 at 2^-29; absolute error < 2^-36 of
 * the maximum below);
 * shapes the kernels do not cover run augcl_enc_x3_*. */
int augcl_enc_amax_f32(const float *x, size_t n, float *amax_out /* device, 1 float */, augcl_stream_t stream);
int augcl_enc_x2_conv32_forward(const float *in, const float *w, const float *bias, float *out, int B, int Hi, int Wi, int relu,
                                void *frag_ws, size_t frag_ws_bytes, const float *amax_in, float *amax_out,
                                augcl_stream_t stream);
int augcl_enc_x2_conv32_backward_data(const float *dz, const float *w, const float *act_prev, float *dact, int B, int Ho, int Wo,
                                      void *frag_ws, size_t frag_ws_bytes, const float *amax_dz, float *amax_out,
                                      augcl_stream_t stream);
int augcl_enc_x2_conv32_backward_weights(const float *act_prev, const float *dz, float *dw, float *db, float *workspace,
                                         size_t workspace_bytes, int B, int Hi, int Wi, const float *amax_act,
                                         const float *amax_dz, augcl_stream_t stream);

/* ---- bf16 PERFORMANCE MODE of the same conv stack (BASELINE config 2 "bf16"; NOT the parity path).
 * Activations are bf16 NHWC (B,H,W,32) in HBM, products on v_mfma_f32_32x32x16_bf16 with f32 accumulation,
 * weights/biases stay f32 masters (rounded to bf16 per launch).  The last layer can emit f32 NCHW for the
 * projection (out_f32_nchw); the incoming gradient is converted with augcl_enc_nchw_f32_to_nhwc_bf16. */
int augcl_enc_bf16_conv_first_forward(const float *in /* (B,C,Hi,Wi) f32 raw */, const float *w, const float *bias,
                                      void *out_bf16_nhwc, int B, int C, int Hi, int Wi, int stride, float scale,
                                      int relu, augcl_stream_t stream);
int augcl_enc_bf16_conv32_forward(const void *in_bf16_nhwc, const float *w, const float *bias, void *out, int B, int Hi,
                                  int Wi, int relu, int out_f32_nchw, augcl_stream_t stream);
int augcl_enc_bf16_conv32_backward_data(const void *dz_bf16_nhwc, const float *w, const void *act_prev_bf16_nhwc,
                                        void *dact_bf16_nhwc, int B, int Ho, int Wo, augcl_stream_t stream);
int augcl_enc_bf16_conv32_backward_weights(const void *act_prev_bf16_nhwc, const void *dz_bf16_nhwc, float *dw, float *db,
                                           float *workspace, size_t workspace_bytes, int B, int Hi, int Wi,
                                           augcl_stream_t stream);
/* same result on the bf16 MFMA pipe (transposed LDS reads supply the pixel-major operands): 8x fewer MFMAs */
int augcl_enc_bf16_conv32_backward_weights_mfma(const void *act_prev_bf16_nhwc, const void *dz_bf16_nhwc, float *dw,
                                                float *db, float *workspace, size_t workspace_bytes, int B, int Hi,
                                                int Wi, augcl_stream_t stream);
int augcl_enc_bf16_conv_first_backward_weights(const float *x /* f32 raw */, const void *dz_bf16_nhwc, float *dw, float *db,
                                               float *workspace, size_t workspace_bytes, int B, int C, int Hi, int Wi,
                                               int stride, float scale, augcl_stream_t stream);
int augcl_enc_nchw_f32_to_nhwc_bf16(const float *in, void *out_bf16_nhwc, int B, int H, int W, augcl_stream_t stream);

/* ---- whole-stack entry points: every layer of the encoder forward / backward behind ONE call (the host pays one
 * FFI crossing per pass).  precision 0 = f32 NCHW on the f32 MFMA pipe, 1 = bf16 NHWC (performance mode), 2 = f32 NCHW
 * with the 32 -> 32 layers on the split-bf16 kernels (augcl_enc_x3_*: float32-accurate, no range limit), 3 = f32 NCHW with
 * the 32 -> 32 layers on the f16-pair kernels (csrc/encoder_x2.hip: every operand = hi + lo' 2^-11 in f16 under a per-tensor
 * power-of-two scale, three v_mfma_f32_32x32x16_f16 per product group -- 1.4e-7 relative L2 against float64 at K = 288, half
 * the matrix instructions of 2; the default parity path).  Precision 3 keeps float32 accuracy (2^-23) for every element
 * within 2^26 of its tensor's largest magnitude, 2^-21 within 2^29; smaller elements keep an absolute error below 2^-36 of
 * that maximum.
 * forward: acts_out[i] = output buffer of layer i (the last layer always writes f32 NCHW).
 * backward: acts_in[i] = input of layer i (acts_in[0] = the raw f32 image batch); dz_a / dz_b = two scratch buffers
 * of the largest activation's size; dw[i] / db[i] receive the weight / bias gradients.
 * frag_ws: device scratch of augcl_enc_frag_workspace_bytes(n_layers - 1, sets) bytes (sets = 1, 2 for the dual forward),
 * 16-byte aligned; the split weight fragments of every 32 -> 32 layer are made by ONE small launch per pass into it.
 * Precision 2: may be NULL (every layer launch then splits its own weights in its prologue, ~6 us per launch more);
 * precision 3: required.  The contents are scratch: valid only inside the call.
 * amax (precision 3; NULL otherwise): n_layers floats (2 n_layers for the dual forward) on the device.  The forward pass
 * leaves amax[i] = max |output of layer i| there (each kernel's epilogue takes the maximum of what it stores); the
 * backward pass of the same activations must be given the same values (amax_acts) plus n_layers floats of scratch
 * (amax_dz) for the gradients' maxima.
 * gate_bits (precision 3; may be NULL, and so may any entry): n_layers device pointers; gate_bits[i] receives one 32-bit word
 * per pixel of layer i's output (B x h_i x w_i words), bit c = (channel c > 0) -- the ReLU gate of the NEXT layer's
 * backward-data, 1/32 of the activation's bytes.  Given the same array, the backward pass reads the bits instead of the
 * activations wherever an entry is non-NULL (an entry the forward pass did not fill must be NULL there). */
size_t augcl_enc_frag_workspace_bytes(int n_layers32, int weight_sets);
int augcl_enc_stack_forward(const float *x, const float *const *w, const float *const *b, void *const *acts_out,
                            int n_layers, int B, int C, int H, int W, int first_stride, float scale, int precision,
                            void *frag_ws, size_t frag_ws_bytes, float *amax, void *const *gate_bits, augcl_stream_t stream);
/* The gradient-carrying forward of precision 3 -- the critic's pass over the augmented batch, src/algorithms/sac.py:103-116 ->
 * src/algorithms/modules.py:145-165 with autograd recording -- with the 32 -> 32 layers as the CHAINS of augcl_enc_stack_infer
 * (csrc/encoder_chain.hip, k_conv32_chain<true>): a layer reads its input from the LDS image the previous layer left, and every
 * layer's output is stored ONCE for the backward pass (float32 NCHW into acts_out[i], each row by the part that owns it), with
 * its gate words and amax[i] exactly as augcl_enc_stack_forward(precision 3) leaves them: augcl_enc_stack_backward takes the
 * result unchanged.  Values are the chains' (per-work-item scales: same accuracy class as the per-layer kernels, not the same
 * bits).  Arguments as augcl_enc_stack_forward with precision 3 (frag_ws and amax required).  AUGCL_E_UNSUPPORTED: a stack the
 * chains do not cover (fewer than 2 layers, an input no plan fits) -- run augcl_enc_stack_forward. */
int augcl_enc_stack_forward_chained(const float *x, const float *const *w, const float *const *b, void *const *acts_out,
                                    int n_layers, int B, int C, int H, int W, int first_stride, float scale, void *frag_ws,
                                    size_t frag_ws_bytes, float *amax, void *const *gate_bits, augcl_stream_t stream);
int augcl_enc_stack_backward(const void *const *acts_in, const float *const *w, const float *grad_out, float *const *dw,
                             float *const *db, void *dz_a, void *dz_b, float *ws32, size_t ws32_bytes, float *ws_first,
                             size_t ws_first_bytes, int n_layers, int B, int C, int H, int W, int first_stride,
                             float scale, int precision, int bf16_wgrad_mfma, void *frag_ws, size_t frag_ws_bytes,
                             const float *amax_acts, float *amax_dz, const void *const *gate_bits, void *dz_chain,
                             size_t dz_chain_bytes, augcl_stream_t stream);
/* dz_chain (precision 1; may be NULL): augcl_enc_stack_backward_chain_bytes(...) bytes of device scratch, 256-byte aligned.  Given
 * it (and frag_ws), the bf16 mode computes the backward-data of ALL 32 -> 32 layers in ONE launch on an LDS-resident canvas
 * (csrc/encoder_chain.hip: the gradient map grows by two pixels per layer in place, every layer's gradient is stored once for
 * the backward-weights launches) -- bit-identical to the per-layer kernels, which run when it is NULL or the canvas does not
 * fit LDS. */
size_t augcl_enc_stack_backward_chain_bytes(int n_layers, int B, int C, int H, int W, int first_stride);

/* Two encoders with identical shapes over the SAME input batch in one launch per layer (no-grad passes: the actor's CNN
 * and the target critic's CNN over next_obs, src/algorithms/sac.py:94-101).  acts_out[i] holds 2B images: [0,B) from
 * weight set A, [B,2B) from set B; the last layer writes f32 NCHW (2B,32,h,w).  AUGCL_E_UNSUPPORTED when the first-layer
 * shape has no pipelined kernel (9 x H x 84, stride 2): run two augcl_enc_stack_forward passes instead. */
int augcl_enc_stack_forward_dual(const float *x, const float *const *wA, const float *const *bA, const float *const *wB,
                                 const float *const *bB, void *const *acts_out, int n_layers, int B, int C, int H, int W,
                                 int first_stride, float scale, int precision, void *frag_ws, size_t frag_ws_bytes,
                                 float *amax, augcl_stream_t stream);

/* The NO-GRAD forward of the whole stack -- SAC.calculate_target_Q's two passes (src/algorithms/sac.py:94-101; wB / bB != NULL:
 * both encoders over the same input, out = (2B,32,h,w) with rows [0,B) from set A and [B,2B) from set B), the actor step's
 * detached pass (sac.py:119-122) and acting (sac.py:73-92; wB = bB = NULL, out = (B,32,h,w)) -- with the f16-pair arithmetic of
 * precision 3 and the 32 -> 32 layers run in CHAINS of up to six layers per launch whose intermediate activations never leave
 * LDS (csrc/encoder_chain.hip): the chain's input (its rows of one image, or of a part of one, plus two halo rows per layer) is
 * staged once into a pixel-major f16 (hi | lo') image and every layer overwrites it in place; an intermediate is stored under a
 * power of two taken from an a-priori bound of that work item's values (the weights' largest absolute row sum x the item's
 * exact input maximum + the largest bias), so no tensor-wide maximum is needed inside a chain and nothing can overflow.  Same
 * accuracy class as augcl_enc_stack_forward(precision 3) -- float32-accurate products, ~1e-7 relative L2 per layer against
 * float64 -- but not bit-identical to it (other scales).  Nothing is kept for a backward pass.
 * precision 3 = that; precision 1 = the bf16 performance mode as chains (bf16 activations need no scales: a whole 41 x 41 image
 * is one work item and the ten 32 -> 32 layers of the training stack are ONE launch; every layer's output is rounded to bf16
 * exactly as augcl_enc_stack_forward(precision 1) stores it).
 * scratch: augcl_enc_stack_infer_scratch_bytes(...) bytes, 256-byte aligned (B_total = B x weight sets); frag_ws as above with
 * the number of weight sets; amax (precision 3; may be NULL otherwise): 2 x (n_layers + 1) floats of device scratch.
 * AUGCL_E_UNSUPPORTED: a shape the chains (or, with two weight sets, the pipelined first-layer kernel) do not cover -- run
 * augcl_enc_stack_forward / augcl_enc_stack_forward_dual instead. */
size_t augcl_enc_stack_infer_scratch_bytes(int n_layers, int B_total, int C, int H, int W, int first_stride);
/* first layer alone from uint8 observations (see augcl_enc_stack_infer_u8): out (B,32,41,41) f32 NCHW */
int augcl_enc_conv_first_forward_u8(const uint8_t *in /* (B,9,84,84) */, const float *w /* (32,9,3,3) */, const float *bias,
                                    float *out, int B, int C, int Hi, int Wi, int stride, float scale, int relu,
                                    augcl_stream_t stream);
/* ..._u8: the same with UINT8 observations (the replay ring's own format, src/utils.py:206-214, wherever the pixels are still
 * integers: next_obs of every algorithm, shifted-only observations): the first layer reads a quarter of the bytes and multiplies
 * exact bf16 pixel operands with three-piece bf16 weights (csrc/encoder_first_u8.hip; conv(x) accumulated exactly in float32,
 * times 1 / scale, plus bias).  precision 3, n_layers >= 2, x (B,9,84,84), first_stride 2; anything else: AUGCL_E_UNSUPPORTED. */
int augcl_enc_stack_infer_u8(const uint8_t *x, const float *const *wA, const float *const *bA, const float *const *wB,
                             const float *const *bB, float *out, void *scratch, size_t scratch_bytes, int n_layers, int B, int C,
                             int H, int W, int first_stride, float scale, int precision, void *frag_ws, size_t frag_ws_bytes,
                             float *amax, augcl_stream_t stream);
int augcl_enc_stack_infer(const float *x, const float *const *wA, const float *const *bA, const float *const *wB,
                          const float *const *bB, float *out, void *scratch, size_t scratch_bytes, int n_layers, int B, int C,
                          int H, int W, int first_stride, float scale, int precision, void *frag_ws, size_t frag_ws_bytes,
                          float *amax, augcl_stream_t stream);

/* ---- device-resident replay sampling: ReplayBuffer.sample -> tensor_buffer_samples / _encode_obses /
 * LazyFrames._force (src/utils.py:206-214, 234-240, 272-280, 293-299).  `frames` is a uint8 ring of RGB frames
 * (frame_bytes = 3*H*W each) in HBM; slots[b*stack + k] names the k-th frame of sample b's stack.
 * out[b, 3k:3k+3, :, :] = frames[slot]; out is uint8 (out_f32 = 0) or float32 (the `.float()` of utils.py:208). */
int augcl_replay_gather(const uint8_t *frames, const int32_t *slots, void *out, int B, int stack, int frame_bytes,
                        int out_f32, augcl_stream_t stream);
/* The whole sample() (src/utils.py:272-280) in one launch from device-resident tables: idx[B] (int64 transition
 * indices, the np.random.randint draw) selects rows of slot_tab[cap][2*stack] (ring slots of the obs | next_obs
 * stacks) and scal[cap][S] (action | reward | not_done); obs / next_obs get the stacked batches (uint8 or float32),
 * scal_out the scalars as three contiguous blocks: actions [B][S-2], rewards [B], not_dones [B]. */
int augcl_replay_sample(const uint8_t *frames, const int32_t *slot_tab, const float *scal, const int64_t *idx, void *obs,
                        void *next_obs, float *scal_out, int B, int stack, int frame_bytes, int S, int out_f32,
                        augcl_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* AUGCL_H */
