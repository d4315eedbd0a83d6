// head_kernels.hip -- the SAC heads' element-wise tails as single launches (forward and backward):
//   * the squashed-Gaussian policy head (src/algorithms/modules.py:19-32 gaussian_logprob / squash and
//     Actor.forward L225-244): ~20 ATen launches forward and ~25 backward in the reference, on (B, A) tensors;
//   * the TD target (src/algorithms/sac.py:94-101) and the twin-Q MSE loss (sac.py:108-110, svea.py:37-56).
// Every formula keeps the reference's operation order; compiled with -ffp-contract=off (the reference's separate
// element-wise kernels never fuse a multiply into an add).  One thread per batch row: A (action dim) is tiny.
#include "augcl_common.h"

namespace augcl {

struct GaussArgs {
    const float *raw;   // (B, 2A): [mu | log_std_raw], the policy MLP's output
    const float *noise; // (B, A) or null (compute_pi = False)
    float *mu_t, *pi_t, *log_pi, *log_std; // outputs; pi_t / log_pi may be null
    const float *g_mu_t, *g_pi_t, *g_log_pi, *g_log_std; // backward: incoming gradients (any may be null)
    float *g_raw;                                        // backward: (B, 2A)
    int B, A;
    int g_pi_ld; // row stride of g_pi_t in floats (>= A): it may be a column slice of a wider matrix
    float lo, k, c; // log_std = lo + k * (tanh(raw) + 1); c = 0.5 * log(2 pi) * A
};

__global__ __launch_bounds__(256) void k_gauss_head_fwd(GaussArgs a)
{
    const int r = blockIdx.x * 256 + threadIdx.x;
    if (r >= a.B) return;
    const float *raw = a.raw + (size_t)r * 2 * a.A;
    float s1 = 0.0f, s2 = 0.0f;
    for (int j = 0; j < a.A; ++j) {
        const float m = raw[j];
        const float t = tanhf(raw[a.A + j]);
        const float ls = a.lo + a.k * (t + 1.0f);
        a.log_std[(size_t)r * a.A + j] = ls;
        a.mu_t[(size_t)r * a.A + j] = tanhf(m);
        if (a.noise) {
            const float n = a.noise[(size_t)r * a.A + j];
            const float p = m + n * expf(ls);
            const float pt = tanhf(p);
            a.pi_t[(size_t)r * a.A + j] = pt;
            s1 += -0.5f * (n * n) - ls;
            s2 += logf(fmaxf(1.0f - pt * pt, 0.0f) + 1e-6f);
        }
    }
    if (a.noise && a.log_pi) a.log_pi[r] = (s1 - a.c) - s2;
}

__global__ __launch_bounds__(256) void k_gauss_head_bwd(GaussArgs a)
{
    const int r = blockIdx.x * 256 + threadIdx.x;
    if (r >= a.B) return;
    const float *raw = a.raw + (size_t)r * 2 * a.A;
    float *g = a.g_raw + (size_t)r * 2 * a.A;
    const float glp = a.g_log_pi ? a.g_log_pi[r] : 0.0f;
    for (int j = 0; j < a.A; ++j) {
        const size_t e = (size_t)r * a.A + j;
        const float m = raw[j];
        const float t = tanhf(raw[a.A + j]);
        const float ls = a.lo + a.k * (t + 1.0f);
        const float mt = tanhf(m);
        float d_m = a.g_mu_t ? a.g_mu_t[e] * (1.0f - mt * mt) : 0.0f;
        float d_ls = a.g_log_std ? a.g_log_std[e] : 0.0f;
        if (a.noise) {
            const float n = a.noise[e];
            const float sd = expf(ls);
            const float pt = tanhf(m + n * sd);
            const float u = 1.0f - pt * pt;
            // d/d pi_t of -sum log(relu(1 - pi_t^2) + 1e-6) = (u > 0) * 2 pi_t / (u + 1e-6)
            float g_pt = a.g_pi_t ? a.g_pi_t[(size_t)r * a.g_pi_ld + j] : 0.0f;
            if (u > 0.0f) g_pt += (glp / (u + 1e-6f)) * (2.0f * pt);
            const float d_p = g_pt * u;
            d_m += d_p;
            d_ls += (d_p * n) * sd - glp;
        }
        g[j] = d_m;
        g[a.A + j] = (d_ls * a.k) * (1.0f - t * t);
    }
}

// target_Q = reward + (not_done * discount) * (min(tq1, tq2) - alpha * log_pi)      (sac.py:94-101)
__global__ __launch_bounds__(256) void k_td_target(const float *__restrict__ tq1, const float *__restrict__ tq2,
                                                   const float *__restrict__ log_pi, const float *__restrict__ reward,
                                                   const float *__restrict__ not_done, const double *__restrict__ alpha,
                                                   int alpha_is_log, float discount, float *__restrict__ out, int B)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= B) return;
    const float al = (float)(alpha_is_log ? exp(alpha[0]) : alpha[0]); // SAC.alpha = log_alpha.exp() in float64
    const float v = fminf(tq1[i], tq2[i]) - al * log_pi[i];
    out[i] = reward[i] + (not_done[i] * discount) * v;
}

// twin-Q loss over S segments of M rows (SVEA: S = 2 halves of one 2B batch against the same M targets):
//   seg[s] = mean_i (q1-y)^2 + mean_i (q2-y)^2 ;  total = sum_s w[s] * seg[s]
// one workgroup, fixed-order tree: deterministic.  out = {total, seg[0..S)}
__global__ __launch_bounds__(256) void k_q_loss_fwd(const float *__restrict__ q1, const float *__restrict__ q2,
                                                    const float *__restrict__ y, int M, int S, float w0, float w1,
                                                    float w2, float w3, float *__restrict__ out)
{
    __shared__ float red[2][256];
    const float w[4] = {w0, w1, w2, w3};
    float total = 0.0f;
    for (int s = 0; s < S; ++s) {
        float a1 = 0.0f, a2 = 0.0f;
        for (int i = threadIdx.x; i < M; i += 256) {
            const float d1 = q1[(size_t)s * M + i] - y[i], d2 = q2[(size_t)s * M + i] - y[i];
            a1 += d1 * d1;
            a2 += d2 * d2;
        }
        red[0][threadIdx.x] = a1;
        red[1][threadIdx.x] = a2;
        __syncthreads();
        for (int k = 128; k > 0; k >>= 1) {
            if ((int)threadIdx.x < k) {
                red[0][threadIdx.x] += red[0][threadIdx.x + k];
                red[1][threadIdx.x] += red[1][threadIdx.x + k];
            }
            __syncthreads();
        }
        const float seg = red[0][0] / (float)M + red[1][0] / (float)M;
        if (threadIdx.x == 0) out[1 + s] = seg;
        total += w[s] * seg;
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = total;
}

// d total / d q1[s*M+i] = g * w[s] * 2 (q1 - y) / M     (mse_loss backward, scaled by the segment weight)
__global__ __launch_bounds__(256) void k_q_loss_bwd(const float *__restrict__ q1, const float *__restrict__ q2,
                                                    const float *__restrict__ y, const float *__restrict__ g, int M,
                                                    int S, float w0, float w1, float w2, float w3,
                                                    float *__restrict__ dq1, float *__restrict__ dq2)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= M * S) return;
    const float w[4] = {w0, w1, w2, w3};
    const int s = i / M;
    const float yy = y[i - s * M];
    const float scale = (g[0] * w[s]) * (2.0f / (float)M);
    dq1[i] = scale * (q1[i] - yy);
    dq2[i] = scale * (q2[i] - yy);
}

// block-wide sum of one float per thread (256 threads), fixed-order tree: deterministic
__device__ __forceinline__ float block_sum_256(float v, float *red)
{
    red[threadIdx.x] = v;
    __syncthreads();
    for (int k = 128; k > 0; k >>= 1) {
        if ((int)threadIdx.x < k) red[threadIdx.x] += red[threadIdx.x + k];
        __syncthreads();
    }
    const float r = red[0];
    __syncthreads();
    return r;
}

// actor loss (src/algorithms/sac.py:119-131): mean_i(alpha * log_pi_i - min(Q1_i, Q2_i)), alpha = float(exp(log_alpha))
__global__ __launch_bounds__(256) void k_actor_loss_fwd(const float *__restrict__ log_pi, const float *__restrict__ q1,
                                                        const float *__restrict__ q2, const double *__restrict__ log_alpha,
                                                        float *__restrict__ out, int B)
{
    __shared__ float red[256];
    const float al = (float)exp(log_alpha[0]);
    float a = 0.0f;
    for (int i = threadIdx.x; i < B; i += 256) a += al * log_pi[i] - fminf(q1[i], q2[i]);
    const float tot = block_sum_256(a, red);
    if (threadIdx.x == 0) out[0] = tot / (float)B;
}

// its backward: d/dlog_pi = g alpha / B; d/dQ of -min(Q1, Q2): -g/B to the smaller one (ties: half each, as ATen's
// minimum backward does)
__global__ __launch_bounds__(256) void k_actor_loss_bwd(const float *__restrict__ q1, const float *__restrict__ q2,
                                                        const double *__restrict__ log_alpha, const float *__restrict__ g,
                                                        float *__restrict__ d_log_pi, float *__restrict__ dq1,
                                                        float *__restrict__ dq2, int B)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= B) return;
    const float gi = g[0] / (float)B;
    d_log_pi[i] = gi * (float)exp(log_alpha[0]);
    const float a = q1[i], b = q2[i];
    dq1[i] = a < b ? -gi : (a == b ? -gi * 0.5f : 0.0f);
    dq2[i] = b < a ? -gi : (a == b ? -gi * 0.5f : 0.0f);
}

// temperature update (sac.py:137-146) in one launch: alpha_loss = mean(alpha * (-log_pi - target_entropy)) with
// alpha = float(exp(log_alpha)) (a 0-dim float64 tensor times a float32 batch stays float32), its gradient with respect
// to the float64 log_alpha, and torch.optim.Adam's step on that scalar in float64.
// state = {exp_avg, exp_avg_sq, step, grad}; mode bit 0: loss + gradient (-> state[3]), bit 1: Adam step from state[3]
// (data parallel: the gradient is averaged over ranks between the two).  out = {alpha_loss, alpha before the step}.
__global__ __launch_bounds__(256) void k_alpha_step(double *__restrict__ log_alpha, const float *__restrict__ log_pi,
                                                    int B, float target_entropy, double lr, double b1, double b2,
                                                    double eps, double *__restrict__ state, float *__restrict__ out,
                                                    int mode)
{
    __shared__ float red[256];
    const double la = log_alpha[0];
    const double alpha = exp(la);
    if (mode & 1) {
        const float al = (float)alpha;
        float a = 0.0f, x = 0.0f;
        for (int i = threadIdx.x; i < B; i += 256) {
            const float xi = -log_pi[i] - target_entropy;
            a += al * xi;
            x += xi * (1.0f / (float)B);
        }
        const float loss = block_sum_256(a, red) / (float)B;
        const float gal = block_sum_256(x, red); // d loss / d alpha (float32 reduction, like ATen's)
        if (threadIdx.x == 0) {
            state[3] = (double)gal * alpha;      // exp backward in float64
            if (out) { out[0] = loss; out[1] = (float)alpha; }
        }
    }
    if ((mode & 2) && threadIdx.x == 0) {
        const double g = state[3];
        const double step = state[2] + 1.0;
        const double m = state[0] + (g - state[0]) * (1.0 - b1); // exp_avg.lerp_(grad, 1 - beta1)
        const double v = state[1] * b2 + (1.0 - b2) * g * g;
        const double bc1 = 1.0 - pow(b1, step), bc2 = 1.0 - pow(b2, step);
        const double denom = sqrt(v) / sqrt(bc2) + eps;
        log_alpha[0] = la - (lr / bc1) * (m / denom);
        state[0] = m; state[1] = v; state[2] = step;
    }
}

// RLProjection's tail (src/algorithms/modules.py:101-111): y = tanh(LayerNorm(x) * gamma + beta), rows of D <= 256.
// One wave per row (lane-strided, wave tree: deterministic).  rstd / mean are recomputed in backward.
__global__ __launch_bounds__(256) void k_ln_tanh_fwd(const float *__restrict__ x, const float *__restrict__ gamma,
                                                     const float *__restrict__ beta, float *__restrict__ y, int B, int D,
                                                     float eps)
{
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= B) return;
    const float *xr = x + (size_t)row * D;
    float v[4], s = 0.0f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int j = lane + 64 * k;
        v[k] = j < D ? xr[j] : 0.0f;
        s += v[k];
    }
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    const float mean = s / (float)D;
    float q = 0.0f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int j = lane + 64 * k;
        const float d = j < D ? v[k] - mean : 0.0f;
        q += d * d;
    }
    for (int o = 32; o > 0; o >>= 1) q += __shfl_xor(q, o, 64);
    const float rstd = 1.0f / sqrtf(q / (float)D + eps);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int j = lane + 64 * k;
        if (j < D) y[(size_t)row * D + j] = tanhf(((v[k] - mean) * rstd) * gamma[j] + beta[j]);
    }
}

// dx of tanh(LN(x)): g1 = gy (1 - y^2); dxhat = g1 gamma; dx = rstd (dxhat - mean(dxhat) - xhat mean(dxhat xhat)).
// Also writes g1 and xhat-weighted g1 for the column sums (dgamma = sum_rows g1 xhat, dbeta = sum_rows g1).
__global__ __launch_bounds__(256) void k_ln_tanh_bwd(const float *__restrict__ x, const float *__restrict__ gamma,
                                                     const float *__restrict__ y, const float *__restrict__ gy,
                                                     float *__restrict__ dx, float *__restrict__ g1_out,
                                                     float *__restrict__ g1x_out, int B, int D, float eps, int gy_ld)
{
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= B) return;
    const float *xr = x + (size_t)row * D;
    float v[4], s = 0.0f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int j = lane + 64 * k;
        v[k] = j < D ? xr[j] : 0.0f;
        s += v[k];
    }
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    const float mean = s / (float)D;
    float q = 0.0f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int j = lane + 64 * k;
        const float d = j < D ? v[k] - mean : 0.0f;
        q += d * d;
    }
    for (int o = 32; o > 0; o >>= 1) q += __shfl_xor(q, o, 64);
    const float rstd = 1.0f / sqrtf(q / (float)D + eps);
    float xh[4], dh[4], a = 0.0f, b = 0.0f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int j = lane + 64 * k;
        xh[k] = 0.0f;
        dh[k] = 0.0f;
        if (j < D) {
            const size_t e = (size_t)row * D + j;
            const float yy = y[e];
            const float g1 = gy[(size_t)row * gy_ld + j] * (1.0f - yy * yy);
            xh[k] = (v[k] - mean) * rstd;
            dh[k] = g1 * gamma[j];
            g1_out[e] = g1;
            g1x_out[e] = g1 * xh[k];
            a += dh[k];
            b += dh[k] * xh[k];
        }
    }
    for (int o = 32; o > 0; o >>= 1) {
        a += __shfl_xor(a, o, 64);
        b += __shfl_xor(b, o, 64);
    }
    a /= (float)D;
    b /= (float)D;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int j = lane + 64 * k;
        if (j < D) dx[(size_t)row * D + j] = rstd * ((dh[k] - a) - xh[k] * b);
    }
}

// column sums over the batch, fixed order: dgamma[j] = sum_rows g1x, dbeta[j] = sum_rows g1 and (optionally) the bias
// gradient db[j] = sum_rows dx of the Linear in front.  One workgroup = 32 columns of one source; 8 row groups of 32
// lanes each sum every 8th row, then the 8 partial sums are added in a fixed order (deterministic).  (The first version
// walked all B rows with one thread per column: 30 us at B = 512 for 3 x 100 sums.)
__global__ __launch_bounds__(256) void k_colsum3(const float *__restrict__ g1, const float *__restrict__ g1x,
                                                 const float *__restrict__ dx, float *__restrict__ dbeta,
                                                 float *__restrict__ dgamma, float *__restrict__ db, int B, int D)
{
    __shared__ float red[8][33];
    const int which = blockIdx.y; // 0: g1x -> dgamma, 1: g1 -> dbeta, 2: dx -> db
    const float *src = which == 0 ? g1x : which == 1 ? g1 : dx;
    float *dst = which == 0 ? dgamma : which == 1 ? dbeta : db;
    const int c = blockIdx.x * 32 + (threadIdx.x & 31), rg = threadIdx.x >> 5;
    float s0 = 0.f, s1 = 0.f;
    if (c < D) {
        int r = rg;
        for (; r + 8 < B; r += 16) {
            s0 += src[(size_t)r * D + c];
            s1 += src[(size_t)(r + 8) * D + c];
        }
        if (r < B) s0 += src[(size_t)r * D + c];
    }
    red[rg][threadIdx.x & 31] = s0 + s1;
    __syncthreads();
    if (rg == 0 && c < D) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) t += red[k][threadIdx.x & 31];
        dst[c] = t;
    }
}

// ---- RLProjection as a whole (src/algorithms/modules.py:101-111): y = x W^T + b (K = 32*21*21 = 14112 -> N = 100), then
// LayerNorm + tanh.  A skinny GEMM: hipBLASLt spends 35-44 us on it (one 16x16 macro-tile kernel); it is 29 MB of
// activations against 1.8 GFLOP on the f32 MFMA pipe (N padded to 128), i.e. ~12 us of matrix-core time.
// k_proj_partial: split-K over S slabs.  A workgroup owns 32 batch rows x all N (<= 128) outputs x one K slab and walks
// the slab in steps of 32: the x tile (32 x 32) and the W tile (128 x 32) are fetched with fully coalesced float4 loads
// (8 threads per 128-B row segment), parked in registers while the previous step's MFMAs run, then written to LDS
// k-major so that a v_mfma_f32_32x32x2_f32 operand is one conflict-free ds_read_b32 per lane: A = W[n][k] (n on lanes),
// B = x[m][k] (batch row on lanes).  Wave t owns output columns 32t..32t+31.  (Feeding the MFMAs straight from global
// memory, each lane reading 16 B of its own row, measured 27 us: every 128-B line was pulled into L1 four times.)
// Each slab's 32 x N partial goes to P[s][m][n]; k_proj_reduce_ln_tanh adds the slabs in slab order, the bias, and
// applies LayerNorm + tanh: every sum in a fixed order -> bitwise reproducible.
typedef float f32x16h __attribute__((ext_vector_type(16)));

struct ProjArgs {
    const float *x, *w;
    float *part; // [S][M][N]
    int M, K, N, S, iters; // iters = 32-wide k steps per slab
};

__global__ __launch_bounds__(256) void k_proj_partial(ProjArgs a)
{
    __shared__ float xs[32][33];   // [k][m]
    __shared__ float wsm[32][129]; // [k][n]; reused as [m][n] for the transposed store at the end
    const int mb = blockIdx.x, s = blockIdx.y, tid = threadIdx.x;
    const int lane = tid & 63, wv = tid >> 6, l31 = lane & 31, h = lane >> 5;
    // loader role: thread -> (row, 16-B segment) of the 32-float step
    const int lr = tid >> 3, lq = tid & 7;
    const size_t kbase = (size_t)s * a.iters * 32;
    const float4 *xg = reinterpret_cast<const float4 *>(a.x + (size_t)min(mb * 32 + lr, a.M - 1) * a.K + kbase) + lq;
    const float4 *wg[4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
        wg[j] = reinterpret_cast<const float4 *>(a.w + (size_t)min(lr + 32 * j, a.N - 1) * a.K + kbase) + lq;
    f32x16h acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
    float4 xr = xg[0], wr[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) wr[j] = wg[j][0];
    for (int it = 0; it < a.iters; ++it) {
        if (it) __syncthreads(); // the previous step's operand reads are done
        {
            const float xe[4] = {xr.x, xr.y, xr.z, xr.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) xs[4 * lq + e][lr] = xe[e];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float we[4] = {wr[j].x, wr[j].y, wr[j].z, wr[j].w};
#pragma unroll
                for (int e = 0; e < 4; ++e) wsm[4 * lq + e][lr + 32 * j] = we[e];
            }
        }
        if (it + 1 < a.iters) { // next step's loads fly under this step's MFMAs
            xr = xg[8 * (it + 1)];
#pragma unroll
            for (int j = 0; j < 4; ++j) wr[j] = wg[j][8 * (it + 1)];
        }
        __syncthreads();
#pragma unroll
        for (int kp = 0; kp < 16; ++kp)
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wsm[2 * kp + h][32 * wv + l31], xs[2 * kp + h][l31], acc, 0, 0, 0);
    }
    __syncthreads();
    float(*out)[129] = reinterpret_cast<float(*)[129]>(&wsm[0][0]); // [m][n], 32 x 129 floats: fits exactly
#pragma unroll
    for (int r = 0; r < 16; ++r) out[l31][32 * wv + (r & 3) + 8 * (r >> 2) + 4 * h] = acc[r];
    __syncthreads();
    const int rows = min(32, a.M - mb * 32);
    float *dst = a.part + ((size_t)s * a.M + (size_t)mb * 32) * a.N;
    for (int i = tid; i < rows * a.N; i += 256) {
        const int mm = i / a.N, n = i - mm * a.N;
        dst[i] = out[mm][n];
    }
}

// y[m][n] = bias[n] + sum_s part[s][m][n] (slab order, four interleaved running sums combined in a fixed order),
// z = tanh(LayerNorm(y) * gamma + beta).  Two rows per workgroup, 128 threads per row, one output column per thread:
// the S loads of a thread are independent (all in flight at once) and coalesced across the row.  N <= 128.
__global__ __launch_bounds__(256) void k_proj_reduce_ln_tanh(const float *__restrict__ part, const float *__restrict__ bias,
                                                             const float *__restrict__ gamma, const float *__restrict__ beta,
                                                             float *__restrict__ y, float *__restrict__ z, int M, int N, int S,
                                                             float eps, float *__restrict__ xcat,
                                                             const float *__restrict__ tail, int A, int tail_rows)
{
    // xcat != null: the critic's input row [z | action] is written as well (row stride N + A <= 128): tail (tail_rows, A),
    // row r takes tail row r % tail_rows -- SVEA's 2B-row pass repeats the B actions -- so no concat launch follows.
    __shared__ float red[2][2][2]; // [row in block][wave of the row][sum | sum of squares]
    const int half = threadIdx.x >> 7, j = threadIdx.x & 127, wv = (threadIdx.x >> 6) & 1;
    const int row = blockIdx.x * 2 + half;
    const bool live = row < M && j < N;
    float t = 0.0f;
    if (live) {
        const float *p = part + (size_t)row * N + j;
        const size_t st = (size_t)M * N;
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
        int k = 0;
        for (; k + 3 < S; k += 4) {
            s0 += p[(size_t)k * st];
            s1 += p[(size_t)(k + 1) * st];
            s2 += p[(size_t)(k + 2) * st];
            s3 += p[(size_t)(k + 3) * st];
        }
        for (; k < S; ++k) s0 += p[(size_t)k * st];
        t = ((s0 + s1) + (s2 + s3)) + bias[j];
        y[(size_t)row * N + j] = t;
    }
    float sum = live ? t : 0.0f;
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
    if ((threadIdx.x & 63) == 0) red[half][wv][0] = sum;
    __syncthreads();
    const float mean = (red[half][0][0] + red[half][1][0]) / (float)N;
    const float d = live ? t - mean : 0.0f;
    float q = d * d;
    for (int o = 32; o > 0; o >>= 1) q += __shfl_xor(q, o, 64);
    if ((threadIdx.x & 63) == 0) red[half][wv][1] = q;
    __syncthreads();
    const float rstd = 1.0f / sqrtf((red[half][0][1] + red[half][1][1]) / (float)N + eps);
    const float zv = live ? tanhf((d * rstd) * gamma[j] + beta[j]) : 0.0f;
    if (live) z[(size_t)row * N + j] = zv;
    if (xcat != nullptr && row < M) {
        if (j < N) xcat[(size_t)row * (N + A) + j] = zv;
        else if (j < N + A) xcat[(size_t)row * (N + A) + j] = tail[(size_t)(row % tail_rows) * A + (j - N)];
    }
}

// ---- the 3-layer MLP trunks (QFunction src/algorithms/modules.py:247-261, Actor.mlp :200-215) around their library
// GEMMs: what ATen spreads over bias-broadcast copies, clamp launches, a (B,H)x(H,1) GEMV, threshold_backward launches and
// one reduce launch per bias gradient.  T trunks side by side (the twin Q functions: T = 2); twin strides in floats.
//
// k_mlp_out_fwd: z (T,B,H) holds a1 W2^T (plain batched GEMM, no bias).  In place a2 = relu(z + b2[t]) and
// q[t][b] = a2[t][b] . w3[t] + b3[t].  One wave per row, lanes stride the row, xor-tree sum: a fixed order.
__global__ __launch_bounds__(256) void k_mlp_out_fwd(float *__restrict__ z, const float *__restrict__ b2, int64_t b2_ts,
                                                     const float *__restrict__ w3, int64_t w3_ts,
                                                     const float *__restrict__ b3, int64_t b3_ts, float *__restrict__ q,
                                                     int T, int B, int H)
{
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6); // = t * B + b
    if (row >= T * B) return;
    const int t = row / B;
    float *zr = z + (size_t)row * H;
    const float *bb = b2 + t * b2_ts, *ww = w3 + t * w3_ts;
    float s = 0.0f;
    for (int c = lane; c < H; c += 64) {
        float v = zr[c] + bb[c];
        v = v < 0.0f ? 0.0f : v; // relu (NaN stays NaN, like clamp_min)
        zr[c] = v;
        s = __builtin_fmaf(v, ww[c], s);
    }
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if (lane == 0) q[row] = s + b3[t * b3_ts];
}

// k_mlp_out_bwd: from g[t][b] = dL/dq: dz2[t][b][c] = (a2 > 0) ? g w3[t][c] : 0, and (when wanted) dw3[t][c] = sum_b g a2,
// db2[t][c] = sum_b dz2, db3[t] = sum_b g.  One workgroup = 32 columns of one trunk x 32 row groups; each row group sums
// every 32nd row, the 32 partial sums are added in row-group order (deterministic).
__global__ __launch_bounds__(1024) void k_mlp_out_bwd(const float *__restrict__ g0, const float *__restrict__ g1,
                                                      const float *__restrict__ a2, const float *__restrict__ w3,
                                                      int64_t w3_ts, float *__restrict__ dz2, float *__restrict__ dw3,
                                                      float *__restrict__ db3, float *__restrict__ db2, int B, int H)
{
    __shared__ float red[3][32][33];
    const int t = blockIdx.y, cl = threadIdx.x & 31, rg = threadIdx.x >> 5;
    const int c = blockIdx.x * 32 + cl;
    const float *g = t ? g1 : g0;
    const float w = c < H ? w3[t * w3_ts + c] : 0.0f;
    float sw = 0.0f, sb = 0.0f, sg = 0.0f;
    if (c < H) {
        const size_t base = (size_t)t * B * H + c;
#pragma unroll 4
        for (int r = rg; r < B; r += 32) {
            const float gv = g[r], av = a2[base + (size_t)r * H];
            const float d = av > 0.0f ? gv * w : 0.0f;
            dz2[base + (size_t)r * H] = d;
            sb += d;
            sw = __builtin_fmaf(gv, av, sw);
        }
    }
    if (db3 && blockIdx.x == 0 && cl == 0)
        for (int r = rg; r < B; r += 32) sg += g[r];
    red[0][rg][cl] = sw;
    red[1][rg][cl] = sb;
    red[2][rg][cl] = sg;
    __syncthreads();
    if (rg < 3) { // row group k of the first three finishes sum k
        float tsum = 0.0f;
#pragma unroll
        for (int k = 0; k < 32; ++k) tsum += red[rg][k][cl];
        if (rg == 0 && dw3 && c < H) dw3[(size_t)t * H + c] = tsum;
        if (rg == 1 && db2 && c < H) db2[(size_t)t * H + c] = tsum;
        if (rg == 2 && db3 && blockIdx.x == 0 && cl == 0) db3[t] = tsum;
    }
}

// k_mask_colsum: the ReLU gate of a hidden layer and the bias gradient of the Linear under it in one pass:
// dz[b][c] = (a[b][c] > 0) ? dz[b][c] : 0 in place, db[c] = sum_b dz[b][c]; the extra workgroups column-sum a second,
// narrow matrix ex (B,E) as it is (the output layer's bias gradient).  Same fixed-order sums as above.
__global__ __launch_bounds__(1024) void k_mask_colsum(float *__restrict__ dz, const float *__restrict__ a,
                                                      float *__restrict__ db, const float *__restrict__ ex,
                                                      float *__restrict__ dex, int B, int N, int E, int nblk_main)
{
    __shared__ float red[32][33];
    const int cl = threadIdx.x & 31, rg = threadIdx.x >> 5;
    const bool main_part = (int)blockIdx.x < nblk_main;
    const int c = (main_part ? (int)blockIdx.x : (int)blockIdx.x - nblk_main) * 32 + cl;
    float s = 0.0f;
    if (main_part) {
        if (c < N) {
#pragma unroll 4
            for (int r = rg; r < B; r += 32) {
                const size_t i = (size_t)r * N + c;
                const float d = a[i] > 0.0f ? dz[i] : 0.0f;
                dz[i] = d;
                s += d;
            }
        }
    } else if (c < E) {
        for (int r = rg; r < B; r += 32) s += ex[(size_t)r * E + c];
    }
    red[rg][cl] = s;
    __syncthreads();
    if (rg == 0) {
        float tsum = 0.0f;
#pragma unroll
        for (int k = 0; k < 32; ++k) tsum += red[k][cl];
        if (main_part) {
            if (db && c < N) db[c] = tsum;
        } else if (c < E) {
            dex[c] = tsum;
        }
    }
}

// ---- RLProjection backward, input side: dx (M,K) = dy (M,N) w (N,K) with K = 14112, N = 100 -- a GEMM with a tiny inner
// dimension whose 29-58 MB output is the cost.  hipBLASLt runs it at 0.7 TB/s (42 us at M = 512, a 64x64x16 macro tile);
// here a wave keeps the weights of its 32 output columns, w[n][k0 .. k0+31], in registers (NP = N/2 MFMA B operands, n on
// the MFMA's k index), the workgroup's four waves share each 32-row dy tile through LDS ([n][m], one conflict-free
// ds_read_b32 per MFMA), and every accumulator row leaves as 128 contiguous bytes.
template <int NP> // MFMA k-pairs: N <= 2 NP
__global__ __launch_bounds__(256, 2) void k_proj_dx(const float *__restrict__ dy, const float *__restrict__ w,
                                                    float *__restrict__ dx, int M, int N, int K, int mb_per_wg)
{
    __shared__ float at[2][2 * NP][33]; // two dy tiles, [n][m]
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, h = lane >> 5;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int k0 = blockIdx.x * 128 + 32 * wv;
    const bool wave_live = k0 < K; // the last column block may own fewer than four 32-column groups
    const int kc = min(k0 + l31, K - 1);
    // this lane's weights: w[n = 2 j + h][kc]
    float wreg[NP];
#pragma unroll
    for (int j = 0; j < NP; ++j) {
        const int n = 2 * j + h;
        wreg[j] = (wave_live && n < N) ? w[(size_t)n * K + kc] : 0.0f;
    }
    // dy tile fetch: 32 rows x N floats as 16-byte chunks, N/4 per row; thread -> chunks tid, tid + 256, ...
    const int n4 = N >> 2, nchunk = 32 * n4;
    constexpr int AU = (32 * 2 * NP / 4 + 255) / 256;
    typedef float vf4 __attribute__((ext_vector_type(4)));
    vf4 areg[AU];
    const int mb0 = blockIdx.y * mb_per_wg, mb1 = min(mb0 + mb_per_wg, (M + 31) >> 5);
    auto issue = [&](int mb) {
#pragma unroll
        for (int u = 0; u < AU; ++u) {
            const int q = min(u * 256 + tid, nchunk - 1);
            const int r = q / n4, c4 = q - r * n4;
            areg[u] = *reinterpret_cast<const vf4 *>(dy + (size_t)min(mb * 32 + r, M - 1) * N + 4 * c4);
        }
    };
    auto commit = [&](int buf) {
#pragma unroll
        for (int u = 0; u < AU; ++u) {
            const int q = min(u * 256 + tid, nchunk - 1);
            const int r = q / n4, c4 = q - r * n4;
#pragma unroll
            for (int e = 0; e < 4; ++e) at[buf][4 * c4 + e][r] = areg[u][e];
        }
    };
    if (2 * NP > N) { // rows n >= N of both tiles meet zero weights; keep them finite
        for (int i = tid; i < 2 * (2 * NP - N) * 33; i += 256) {
            const int b = i / ((2 * NP - N) * 33), rem = i - b * (2 * NP - N) * 33;
            at[b][N + rem / 33][rem % 33] = 0.0f;
        }
    }
    if (mb0 < mb1) issue(mb0);
    int buf = 0;
    for (int mb = mb0; mb < mb1; ++mb, buf ^= 1) {
        commit(buf);
        __syncthreads(); // tile `buf` complete.  Its previous readers (two iterations back) all passed the barrier in between.
        if (mb + 1 < mb1) issue(mb + 1);
        if (wave_live) {
            f32x16h acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
            // all NP A operands are requested before the first MFMA (they return in order; an MFMA of 64 cycles is shorter
            // than one LDS round trip, so a one-deep prefetch left the pipe waiting half the time)
            float av[NP];
#pragma unroll
            for (int j = 0; j < NP; ++j) av[j] = at[buf][2 * j + h][l31];
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < NP; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[j], wreg[j], acc, 0, 0, 0);
            // acc[r]: row m = mb*32 + (r&3) + 8 (r>>2) + 4 h, column k0 + l31
            if (k0 + l31 < K) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = mb * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                    if (m < M) dx[(size_t)m * K + k0 + l31] = acc[r];
                }
            }
        }
    }
}

} // namespace augcl

using namespace augcl;

extern "C" {

int augcl_gauss_head_forward(const float *raw, const float *noise, float *mu_t, float *pi_t, float *log_pi,
                             float *log_std, int B, int A, double log_std_min, double log_std_max,
                             augcl_stream_t stream)
{
    const char *name = "augcl_gauss_head_forward";
    AUGCL_REQUIRE(B >= 0 && A >= 1, "%s: bad shape", name);
    if (B == 0) return AUGCL_OK;
    AUGCL_REQUIRE(raw && mu_t && log_std && (!noise || pi_t), "%s: null tensor", name);
    GaussArgs a{};
    a.raw = raw; a.noise = noise; a.mu_t = mu_t; a.pi_t = pi_t; a.log_pi = log_pi; a.log_std = log_std;
    a.B = B; a.A = A;
    a.lo = (float)log_std_min; a.k = (float)(0.5 * (log_std_max - log_std_min));
    a.c = (float)(0.5 * 1.8378770664093453 /* log(2 pi) */ * (double)A);
    hipLaunchKernelGGL(k_gauss_head_fwd, dim3((B + 255) / 256), dim3(256), 0, as_stream(stream), a);
    return augcl::check_launch(name);
}

int augcl_gauss_head_backward(const float *raw, const float *noise, const float *g_mu_t, const float *g_pi_t,
                              int g_pi_ld, const float *g_log_pi, const float *g_log_std, float *g_raw, int B, int A,
                              double log_std_min, double log_std_max, augcl_stream_t stream)
{
    const char *name = "augcl_gauss_head_backward";
    AUGCL_REQUIRE(B >= 0 && A >= 1 && (!g_pi_t || g_pi_ld >= A), "%s: bad shape", name);
    if (B == 0) return AUGCL_OK;
    AUGCL_REQUIRE(raw && g_raw, "%s: null tensor", name);
    GaussArgs a{};
    a.raw = raw; a.noise = noise; a.g_mu_t = g_mu_t; a.g_pi_t = g_pi_t; a.g_log_pi = g_log_pi; a.g_log_std = g_log_std;
    a.g_raw = g_raw; a.B = B; a.A = A; a.g_pi_ld = g_pi_ld;
    a.lo = (float)log_std_min; a.k = (float)(0.5 * (log_std_max - log_std_min));
    hipLaunchKernelGGL(k_gauss_head_bwd, dim3((B + 255) / 256), dim3(256), 0, as_stream(stream), a);
    return augcl::check_launch(name);
}

int augcl_td_target_f32(const float *tq1, const float *tq2, const float *log_pi, const float *reward,
                        const float *not_done, const double *alpha_dev, int alpha_is_log, double discount, float *out,
                        int B, augcl_stream_t stream)
{
    const char *name = "augcl_td_target_f32";
    AUGCL_REQUIRE(B >= 0, "%s: bad shape", name);
    if (B == 0) return AUGCL_OK;
    AUGCL_REQUIRE(tq1 && tq2 && log_pi && reward && not_done && alpha_dev && out, "%s: null tensor", name);
    hipLaunchKernelGGL(k_td_target, dim3((B + 255) / 256), dim3(256), 0, as_stream(stream), tq1, tq2, log_pi, reward,
                       not_done, alpha_dev, alpha_is_log, (float)discount, out, B);
    return augcl::check_launch(name);
}

int augcl_actor_loss_forward(const float *log_pi, const float *q1, const float *q2, const double *log_alpha, float *out,
                             int B, augcl_stream_t stream)
{
    const char *name = "augcl_actor_loss_forward";
    AUGCL_REQUIRE(B >= 1, "%s: bad shape", name);
    AUGCL_REQUIRE(log_pi && q1 && q2 && log_alpha && out, "%s: null tensor", name);
    hipLaunchKernelGGL(k_actor_loss_fwd, dim3(1), dim3(256), 0, as_stream(stream), log_pi, q1, q2, log_alpha, out, B);
    return augcl::check_launch(name);
}

int augcl_actor_loss_backward(const float *q1, const float *q2, const double *log_alpha, const float *g, float *d_log_pi,
                              float *dq1, float *dq2, int B, augcl_stream_t stream)
{
    const char *name = "augcl_actor_loss_backward";
    AUGCL_REQUIRE(B >= 1, "%s: bad shape", name);
    AUGCL_REQUIRE(q1 && q2 && log_alpha && g && d_log_pi && dq1 && dq2, "%s: null tensor", name);
    hipLaunchKernelGGL(k_actor_loss_bwd, dim3((B + 255) / 256), dim3(256), 0, as_stream(stream), q1, q2, log_alpha, g,
                       d_log_pi, dq1, dq2, B);
    return augcl::check_launch(name);
}

int augcl_alpha_step_f64(double *log_alpha, const float *log_pi, int B, double target_entropy, double lr, double beta1,
                         double beta2, double eps, double *state, float *out, int mode, augcl_stream_t stream)
{
    const char *name = "augcl_alpha_step_f64";
    AUGCL_REQUIRE(B >= 1 && mode >= 1 && mode <= 3, "%s: bad argument", name);
    AUGCL_REQUIRE(log_alpha && state && (log_pi || !(mode & 1)), "%s: null tensor", name);
    hipLaunchKernelGGL(k_alpha_step, dim3(1), dim3(256), 0, as_stream(stream), log_alpha, log_pi, B, (float)target_entropy,
                       lr, beta1, beta2, eps, state, out, mode);
    return augcl::check_launch(name);
}

int augcl_q_loss_forward(const float *q1, const float *q2, const float *y, int M, int S, const double *w, float *out,
                         augcl_stream_t stream)
{
    const char *name = "augcl_q_loss_forward";
    AUGCL_REQUIRE(M >= 1 && S >= 1 && S <= 4, "%s: bad shape", name);
    AUGCL_REQUIRE(q1 && q2 && y && w && out, "%s: null tensor", name);
    float ws[4] = {0, 0, 0, 0};
    for (int s = 0; s < S; ++s) ws[s] = (float)w[s];
    hipLaunchKernelGGL(k_q_loss_fwd, dim3(1), dim3(256), 0, as_stream(stream), q1, q2, y, M, S, ws[0], ws[1], ws[2], ws[3], out);
    return augcl::check_launch(name);
}

int augcl_q_loss_backward(const float *q1, const float *q2, const float *y, const float *g_total, int M, int S,
                          const double *w, float *dq1, float *dq2, augcl_stream_t stream)
{
    const char *name = "augcl_q_loss_backward";
    AUGCL_REQUIRE(M >= 1 && S >= 1 && S <= 4, "%s: bad shape", name);
    AUGCL_REQUIRE(q1 && q2 && y && g_total && w && dq1 && dq2, "%s: null tensor", name);
    float ws[4] = {0, 0, 0, 0};
    for (int s = 0; s < S; ++s) ws[s] = (float)w[s];
    hipLaunchKernelGGL(k_q_loss_bwd, dim3((M * S + 255) / 256), dim3(256), 0, as_stream(stream), q1, q2, y, g_total, M, S,
                       ws[0], ws[1], ws[2], ws[3], dq1, dq2);
    return augcl::check_launch(name);
}

int augcl_ln_tanh_forward(const float *x, const float *gamma, const float *beta, float *y, int B, int D, double eps,
                          augcl_stream_t stream)
{
    const char *name = "augcl_ln_tanh_forward";
    AUGCL_REQUIRE(B >= 0 && D >= 1 && D <= 256, "%s: bad shape (D <= 256)", name);
    if (B == 0) return AUGCL_OK;
    AUGCL_REQUIRE(x && gamma && beta && y, "%s: null tensor", name);
    hipLaunchKernelGGL(k_ln_tanh_fwd, dim3((B + 3) / 4), dim3(256), 0, as_stream(stream), x, gamma, beta, y, B, D, (float)eps);
    return augcl::check_launch(name);
}

int augcl_ln_tanh_backward(const float *x, const float *gamma, const float *y, const float *gy, int gy_ld, float *dx,
                           float *dgamma, float *dbeta, float *dbias, float *workspace /* 2*B*D floats */, int B, int D,
                           double eps, augcl_stream_t stream)
{
    const char *name = "augcl_ln_tanh_backward";
    AUGCL_REQUIRE(B >= 1 && D >= 1 && D <= 256 && gy_ld >= D, "%s: bad shape (D <= 256, gy_ld >= D)", name);
    AUGCL_REQUIRE(x && gamma && y && gy && dx && dgamma && dbeta && workspace, "%s: null tensor", name);
    float *g1 = workspace, *g1x = workspace + (size_t)B * D;
    hipStream_t s = as_stream(stream);
    hipLaunchKernelGGL(k_ln_tanh_bwd, dim3((B + 3) / 4), dim3(256), 0, s, x, gamma, y, gy, dx, g1, g1x, B, D, (float)eps,
                       gy_ld);
    hipLaunchKernelGGL(k_colsum3, dim3((D + 31) / 32, dbias ? 3 : 2), dim3(256), 0, s, g1, g1x, dx, dbeta, dgamma, dbias, B, D);
    return augcl::check_launch(name);
}

int augcl_proj_ln_tanh_workspace_slabs(int M, int K, int N)
{
    if (M < 1 || N < 1 || N > 128 || K < 32 || (K & 31)) return 0;
    const int steps = K / 32;
    const int cand[8] = {63, 49, 32, 21, 16, 9, 8, 7};
    const int want = M >= 1024 ? 24 : 64; // wave tasks = (M/32) * S * 4: about 4 per SIMD without an oversized workspace
    int best = 0;
    for (int i = 0; i < 8; ++i) {
        if (steps % cand[i] != 0) continue;
        if (cand[i] <= want) return cand[i]; // the largest admissible slab count within the budget
        best = cand[i];                      // otherwise the smallest admissible one
    }
    return best;
}

static int proj_forward_impl(const float *x, const float *w, const float *bias, const float *gamma, const float *beta,
                             float *y, float *z, float *xcat, const float *tail, int A, int tail_rows, float *workspace,
                             size_t workspace_bytes, int M, int K, int N, double eps, augcl_stream_t stream, const char *name)
{
    const int S = augcl_proj_ln_tanh_workspace_slabs(M, K, N);
    AUGCL_REQUIRE(S > 0, "%s: shape (M=%d, K=%d, N=%d) not covered (N <= 128, K/32 divisible by one of 63,49,32,21,16,9,8,7)", name, M, K, N);
    AUGCL_REQUIRE(x && w && bias && gamma && beta && y && z && workspace, "%s: null tensor", name);
    AUGCL_REQUIRE(aligned16(x) && aligned16(w), "%s: x and w must be 16-byte aligned", name);
    AUGCL_REQUIRE(workspace_bytes >= (size_t)S * M * N * sizeof(float), "%s: workspace too small (%zu < %zu)", name,
                  workspace_bytes, (size_t)S * M * N * sizeof(float));
    AUGCL_REQUIRE(!xcat || (tail && A >= 1 && N + A <= 128 && tail_rows >= 1 && M % tail_rows == 0),
                  "%s: concatenated output needs a tail with N + A <= 128 and M a multiple of its rows", name);
    ProjArgs a{x, w, workspace, M, K, N, S, K / 32 / S};
    hipStream_t s = as_stream(stream);
    hipLaunchKernelGGL(k_proj_partial, dim3((M + 31) / 32, S), dim3(256), 0, s, a);
    hipLaunchKernelGGL(k_proj_reduce_ln_tanh, dim3((M + 1) / 2), dim3(256), 0, s, workspace, bias, gamma, beta, y, z, M, N, S,
                       (float)eps, xcat, tail, A, tail_rows);
    return augcl::check_launch(name);
}

int augcl_proj_ln_tanh_forward(const float *x, const float *w, const float *bias, const float *gamma, const float *beta,
                               float *y, float *z, float *workspace, size_t workspace_bytes, int M, int K, int N, double eps,
                               augcl_stream_t stream)
{
    return proj_forward_impl(x, w, bias, gamma, beta, y, z, nullptr, nullptr, 0, 1, workspace, workspace_bytes, M, K, N, eps,
                             stream, "augcl_proj_ln_tanh_forward");
}

int augcl_proj_ln_tanh_cat_forward(const float *x, const float *w, const float *bias, const float *gamma, const float *beta,
                                   float *y, float *z, float *xcat, const float *tail, int A, int tail_rows, float *workspace,
                                   size_t workspace_bytes, int M, int K, int N, double eps, augcl_stream_t stream)
{
    const char *name = "augcl_proj_ln_tanh_cat_forward";
    AUGCL_REQUIRE(xcat, "%s: null output", name);
    return proj_forward_impl(x, w, bias, gamma, beta, y, z, xcat, tail, A, tail_rows, workspace, workspace_bytes, M, K, N, eps,
                             stream, name);
}

int augcl_mlp_out_forward(float *z, const float *b2, int64_t b2_twin_stride, const float *w3, int64_t w3_twin_stride,
                          const float *b3, int64_t b3_twin_stride, float *q, int T, int B, int H, augcl_stream_t stream)
{
    const char *name = "augcl_mlp_out_forward";
    AUGCL_REQUIRE(T >= 1 && B >= 0 && H >= 1 && (size_t)T * B < (1u << 30), "%s: bad shape", name);
    if (B == 0) return AUGCL_OK;
    AUGCL_REQUIRE(z && b2 && w3 && b3 && q, "%s: null tensor", name);
    hipLaunchKernelGGL(k_mlp_out_fwd, dim3((T * B + 3) / 4), dim3(256), 0, as_stream(stream), z, b2, b2_twin_stride, w3,
                       w3_twin_stride, b3, b3_twin_stride, q, T, B, H);
    return augcl::check_launch(name);
}

int augcl_mlp_out_backward(const float *g0, const float *g1, const float *a2, const float *w3, int64_t w3_twin_stride,
                           float *dz2, float *dw3, float *db3, float *db2, int T, int B, int H, augcl_stream_t stream)
{
    const char *name = "augcl_mlp_out_backward";
    AUGCL_REQUIRE((T == 1 || T == 2) && B >= 1 && H >= 1, "%s: bad shape (T = 1 or 2 trunks, B >= 1)", name);
    AUGCL_REQUIRE(g0 && (T == 1 || g1) && a2 && w3 && dz2, "%s: null tensor", name);
    hipLaunchKernelGGL(k_mlp_out_bwd, dim3((H + 31) / 32, T), dim3(1024), 0, as_stream(stream), g0, g1, a2, w3,
                       w3_twin_stride, dz2, dw3, db3, db2, B, H);
    return augcl::check_launch(name);
}

int augcl_mask_colsum(float *dz, const float *a, float *db, const float *extra, float *dextra, int B, int N, int E,
                      augcl_stream_t stream)
{
    const char *name = "augcl_mask_colsum";
    AUGCL_REQUIRE(B >= 1 && N >= 1 && E >= 0, "%s: bad shape", name);
    AUGCL_REQUIRE(dz && a, "%s: null tensor", name);
    AUGCL_REQUIRE(E == 0 || (extra && dextra), "%s: extra matrix without its buffers", name);
    const int nb = (N + 31) / 32;
    hipLaunchKernelGGL(k_mask_colsum, dim3(nb + (E + 31) / 32), dim3(1024), 0, as_stream(stream), dz, a, db, extra, dextra, B,
                       N, E, nb);
    return augcl::check_launch(name);
}

int augcl_proj_dx_f32(const float *dy, const float *w, float *dx, int M, int N, int K, augcl_stream_t stream)
{
    const char *name = "augcl_proj_dx_f32";
    AUGCL_REQUIRE(M >= 1 && N >= 4 && N <= 128 && (N & 3) == 0 && K >= 1, "%s: shape not covered (N a multiple of 4, <= 128)", name);
    AUGCL_REQUIRE(dy && w && dx, "%s: null tensor", name);
    AUGCL_REQUIRE(aligned16(dy), "%s: dy must be 16-byte aligned", name);
    const int mblocks = (M + 31) / 32;
    // row tiles per workgroup (the weights' registers are loaded once per workgroup): ~444 workgroups at M = 256 and 512,
    // all resident at two per CU
    const int per = mblocks >= 16 ? 4 : mblocks >= 8 ? 2 : 1;
    const dim3 grid((K + 127) / 128, (mblocks + per - 1) / per);
    hipStream_t s = as_stream(stream);
    if (N <= 100)
        hipLaunchKernelGGL(k_proj_dx<50>, grid, dim3(256), 0, s, dy, w, dx, M, N, K, per);
    else
        hipLaunchKernelGGL(k_proj_dx<64>, grid, dim3(256), 0, s, dy, w, dx, M, N, K, per);
    return augcl::check_launch(name);
}

} // extern "C"
