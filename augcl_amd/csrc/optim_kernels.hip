// optim_kernels.hip -- fused optimiser / target-network kernels over FLAT parameter buffers.
//
// The reference walks ~40 parameter tensors per optimiser with 5-8 tiny kernels each
// (torch.optim.Adam, src/algorithms/sac.py:48-56) and ~40x3 more for the target EMA
// (utils.soft_update_params, src/utils.py:76-78).  Here every optimiser group lives in ONE
// contiguous f32 buffer (param / grad / exp_avg / exp_avg_sq), so a step is one streaming pass:
// 16 B/lane loads, 4 streams in, 3 out -- pure HBM work (28 B per parameter).
// Compiled with -ffp-contract=off: the EMA restates `tau*p + (1-tau)*t` as separate IEEE ops.
#include "augcl_common.h"

namespace augcl {

// torch.optim.Adam (no amsgrad, no weight decay), single-tensor formulation:
//   m = b1*m + (1-b1)*g ; v = b2*v + (1-b2)*g*g
//   p -= (lr/bc1) * m / (sqrt(v)/sqrt(bc2) + eps)
// g is first multiplied by grad_scale (1/world_size after a SUM all-reduce; 1 otherwise).
// hyper != nullptr: step_size / inv_bc2_sqrt come from device memory (written by adam_prep -- its own 1-thread launch or
// the tail of the gradient gather -- earlier in the same stream), so a captured graph replays with the step count of the
// replay, not of the capture.
__global__ __launch_bounds__(256) void k_adam_f32(float4 *__restrict__ p, const float4 *__restrict__ g,
                                                  float4 *__restrict__ m, float4 *__restrict__ v, size_t n4,
                                                  float b1, float one_m_b1, float b2, float one_m_b2,
                                                  float step_size, float inv_bc2_sqrt, float eps, float grad_scale,
                                                  const float *__restrict__ hyper)
{
    if (hyper) {
        step_size = hyper[0];
        inv_bc2_sqrt = hyper[1];
    }
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        float4 pp = p[i], gg = g[i], mm = m[i], vv = v[i];
        float *P = &pp.x, *G = &gg.x, *M = &mm.x, *V = &vv.x;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float gr = G[k] * grad_scale;
            M[k] = M[k] * b1 + one_m_b1 * gr;
            V[k] = V[k] * b2 + one_m_b2 * (gr * gr);
            const float denom = sqrtf(V[k]) * inv_bc2_sqrt + eps;
            P[k] = P[k] - step_size * (M[k] / denom);
        }
        p[i] = pp;
        m[i] = mm;
        v[i] = vv;
    }
}

__global__ __launch_bounds__(256) void k_adam_f32_tail(float *__restrict__ p, const float *__restrict__ g,
                                                       float *__restrict__ m, float *__restrict__ v, size_t begin,
                                                       size_t n, float b1, float one_m_b1, float b2, float one_m_b2,
                                                       float step_size, float inv_bc2_sqrt, float eps,
                                                       float grad_scale, const float *__restrict__ hyper)
{
    const size_t i = begin + (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    if (hyper) {
        step_size = hyper[0];
        inv_bc2_sqrt = hyper[1];
    }
    const float gr = g[i] * grad_scale;
    const float mm = m[i] * b1 + one_m_b1 * gr;
    const float vv = v[i] * b2 + one_m_b2 * (gr * gr);
    m[i] = mm;
    v[i] = vv;
    p[i] = p[i] - step_size * (mm / (sqrtf(vv) * inv_bc2_sqrt + eps));
}

// device-side step counter: state = {int64 step; float step_size; float inv_bc2_sqrt}.  ++step, then the same
// double-precision scalar prep the host variant does.
__device__ __forceinline__ void adam_prep(void *state, double lr, double beta1, double beta2)
{
    long long *step = reinterpret_cast<long long *>(state);
    float *hyper = reinterpret_cast<float *>(step + 1);
    const long long t = *step + 1;
    *step = t;
    const double bc1 = 1.0 - pow(beta1, (double)t);
    const double bc2 = 1.0 - pow(beta2, (double)t);
    hyper[0] = (float)(lr / bc1);
    hyper[1] = (float)(1.0 / sqrt(bc2));
}

__global__ void k_adam_prep(void *state, double lr, double beta1, double beta2) { adam_prep(state, lr, beta1, beta2); }

// Gradient gather: up to GATHER_MAX tensors (autograd's freshly allocated gradients) copied to their slots of the flat
// bucket in ONE launch -- the table travels in the kernel arguments (a hipGraph keeps it by value; the graph's memory
// pool hands autograd the same addresses at every replay).  blockIdx.y = tensor, blockIdx.x strides its elements;
// 16-byte copies when both ends allow.  Workgroup (0,0) also advances the optimiser's device-side step (adam_prep),
// so the Adam launch that follows needs no 1-thread launch in front of it.
constexpr int GATHER_MAX = 96;
struct GatherTable {
    const float *src[GATHER_MAX];
    long long dst_off[GATHER_MAX]; // in floats
    unsigned numel[GATHER_MAX];
};

__global__ __launch_bounds__(256) void k_gather_f32(GatherTable tb, float *__restrict__ dst, void *step_state, double lr,
                                                    double beta1, double beta2)
{
    if (step_state && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) adam_prep(step_state, lr, beta1, beta2);
    const float *__restrict__ s = tb.src[blockIdx.y];
    float *__restrict__ d = dst + tb.dst_off[blockIdx.y];
    const unsigned n = tb.numel[blockIdx.y];
    const unsigned first = blockIdx.x * 256 + threadIdx.x, stride = gridDim.x * 256;
    if (((reinterpret_cast<uintptr_t>(s) | reinterpret_cast<uintptr_t>(d)) & 15u) == 0) {
        const unsigned n4 = n >> 2;
        for (unsigned i = first; i < n4; i += stride) reinterpret_cast<float4 *>(d)[i] = reinterpret_cast<const float4 *>(s)[i];
        for (unsigned i = (n4 << 2) + first; i < n; i += stride) d[i] = s[i];
    } else {
        for (unsigned i = first; i < n; i += stride) d[i] = s[i];
    }
}

// utils.soft_update_params: target <- tau*p + (1-tau)*target   (exact op order, no FMA)
__device__ __forceinline__ void ema_segment(float *__restrict__ t, const float *__restrict__ p, size_t n, float tau,
                                            float one_m_tau, size_t first, size_t stride)
{
    const size_t n4 = n / 4;
    float4 *t4 = reinterpret_cast<float4 *>(t);
    const float4 *p4 = reinterpret_cast<const float4 *>(p);
    const bool vec = ((reinterpret_cast<uintptr_t>(t) | reinterpret_cast<uintptr_t>(p)) & 15u) == 0;
    if (vec) {
        for (size_t i = first; i < n4; i += stride) {
            float4 a = t4[i];
            const float4 b = p4[i];
            a.x = tau * b.x + one_m_tau * a.x;
            a.y = tau * b.y + one_m_tau * a.y;
            a.z = tau * b.z + one_m_tau * a.z;
            a.w = tau * b.w + one_m_tau * a.w;
            t4[i] = a;
        }
        for (size_t i = n4 * 4 + first; i < n; i += stride) t[i] = tau * p[i] + one_m_tau * t[i];
    } else {
        for (size_t i = first; i < n; i += stride) t[i] = tau * p[i] + one_m_tau * t[i];
    }
}

// two segments, each with its own tau, in one launch: blockIdx.y picks the segment
__global__ __launch_bounds__(256) void k_ema_f32(float *__restrict__ t0, const float *__restrict__ p0, size_t n0,
                                                 float tau0, float one_m_tau0, float *__restrict__ t1,
                                                 const float *__restrict__ p1, size_t n1, float tau1, float one_m_tau1)
{
    const size_t first = (size_t)blockIdx.x * 256 + threadIdx.x, stride = (size_t)gridDim.x * 256;
    if (blockIdx.y == 0) ema_segment(t0, p0, n0, tau0, one_m_tau0, first, stride);
    else ema_segment(t1, p1, n1, tau1, one_m_tau1, first, stride);
}

} // namespace augcl

using namespace augcl;

extern "C" {

static int adam_impl(float *param, const float *grad, float *exp_avg, float *exp_avg_sq, size_t n, double lr,
                     double beta1, double beta2, double eps, int64_t step, void *step_state, int advance,
                     double grad_scale, augcl_stream_t stream, const char *name)
{
    if (n == 0) return AUGCL_OK;
    AUGCL_REQUIRE(param && grad && exp_avg && exp_avg_sq, "%s: null buffer", name);
    AUGCL_REQUIRE(aligned16(param) && aligned16(grad) && aligned16(exp_avg) && aligned16(exp_avg_sq),
                  "%s: buffers must be 16-byte aligned", name);
    hipStream_t s = as_stream(stream);
    float step_size = 0.0f, inv_bc2_sqrt = 0.0f;
    const float *hyper = nullptr;
    if (step_state) { // device-side step: graph-capturable
        AUGCL_REQUIRE((reinterpret_cast<uintptr_t>(step_state) & 7u) == 0, "%s: step state must be 8-byte aligned", name);
        if (advance) hipLaunchKernelGGL(k_adam_prep, dim3(1), dim3(1), 0, s, step_state, lr, beta1, beta2);
        hyper = reinterpret_cast<const float *>(reinterpret_cast<const char *>(step_state) + 8);
    } else {
        AUGCL_REQUIRE(step >= 1, "%s: step must be >= 1 (got %lld)", name, (long long)step);
        // scalar prep in double, like torch.optim.Adam's python side
        const double bc1 = 1.0 - pow(beta1, (double)step);
        const double bc2 = 1.0 - pow(beta2, (double)step);
        step_size = (float)(lr / bc1);
        inv_bc2_sqrt = (float)(1.0 / sqrt(bc2));
    }
    const size_t n4 = n / 4;
    if (n4) {
        const size_t blocks = (n4 + 255) / 256;
        hipLaunchKernelGGL(k_adam_f32, dim3((unsigned)(blocks < 4096 ? blocks : 4096)), dim3(256), 0, s,
                           reinterpret_cast<float4 *>(param), reinterpret_cast<const float4 *>(grad),
                           reinterpret_cast<float4 *>(exp_avg), reinterpret_cast<float4 *>(exp_avg_sq), n4,
                           (float)beta1, (float)(1.0 - beta1), (float)beta2, (float)(1.0 - beta2), step_size,
                           inv_bc2_sqrt, (float)eps, (float)grad_scale, hyper);
    }
    if (n4 * 4 < n)
        hipLaunchKernelGGL(k_adam_f32_tail, dim3(1), dim3(256), 0, s, param, grad, exp_avg, exp_avg_sq, n4 * 4, n,
                           (float)beta1, (float)(1.0 - beta1), (float)beta2, (float)(1.0 - beta2), step_size,
                           inv_bc2_sqrt, (float)eps, (float)grad_scale, hyper);
    return augcl::check_launch(name);
}

int augcl_adam_f32(float *param, const float *grad, float *exp_avg, float *exp_avg_sq, size_t n, double lr,
                   double beta1, double beta2, double eps, int64_t step, double grad_scale, augcl_stream_t stream)
{
    return adam_impl(param, grad, exp_avg, exp_avg_sq, n, lr, beta1, beta2, eps, step, nullptr, 0, grad_scale, stream,
                     "augcl_adam_f32");
}

int augcl_adam_f32_devstep(float *param, const float *grad, float *exp_avg, float *exp_avg_sq, size_t n, double lr,
                           double beta1, double beta2, double eps, void *step_state, int advance, double grad_scale,
                           augcl_stream_t stream)
{
    const char *name = "augcl_adam_f32_devstep";
    AUGCL_REQUIRE(step_state, "%s: null step state", name);
    return adam_impl(param, grad, exp_avg, exp_avg_sq, n, lr, beta1, beta2, eps, 0, step_state, advance, grad_scale, stream,
                     name);
}

int augcl_gather_f32(const void *const *src, const int64_t *dst_offset, const int64_t *numel, int n_tensors, float *dst,
                     void *step_state, double lr, double beta1, double beta2, augcl_stream_t stream)
{
    const char *name = "augcl_gather_f32";
    AUGCL_REQUIRE(n_tensors >= 0, "%s: bad tensor count", name);
    AUGCL_REQUIRE(n_tensors == 0 || (src && dst_offset && numel && dst), "%s: null table", name);
    AUGCL_REQUIRE(!step_state || (reinterpret_cast<uintptr_t>(step_state) & 7u) == 0, "%s: step state must be 8-byte aligned",
                  name);
    for (int i = 0; i < n_tensors; ++i) {
        AUGCL_REQUIRE(src[i] && aligned4(src[i]), "%s: tensor %d: null or misaligned source", name, i);
        AUGCL_REQUIRE(dst_offset[i] >= 0 && numel[i] >= 0 && numel[i] < (1ll << 32), "%s: tensor %d: bad extent", name, i);
    }
    hipStream_t s = as_stream(stream);
    if (n_tensors == 0) {
        if (step_state) hipLaunchKernelGGL(k_adam_prep, dim3(1), dim3(1), 0, s, step_state, lr, beta1, beta2);
        return augcl::check_launch(name);
    }
    for (int base = 0; base < n_tensors; base += GATHER_MAX) {
        const int m = n_tensors - base < GATHER_MAX ? n_tensors - base : GATHER_MAX;
        GatherTable tb;
        memset(&tb, 0, sizeof(tb));
        int64_t longest = 1;
        for (int i = 0; i < m; ++i) {
            tb.src[i] = static_cast<const float *>(src[base + i]);
            tb.dst_off[i] = dst_offset[base + i];
            tb.numel[i] = (unsigned)numel[base + i];
            if (numel[base + i] > longest) longest = numel[base + i];
        }
        const int64_t bx = (longest / 4 + 255) / 256; // one 16-byte chunk per thread for the longest tensor, capped
        hipLaunchKernelGGL(k_gather_f32, dim3((unsigned)(bx < 1 ? 1 : bx > 128 ? 128 : bx), m), dim3(256), 0, s, tb, dst,
                           base == 0 ? step_state : nullptr, lr, beta1, beta2);
    }
    return augcl::check_launch(name);
}

int augcl_ema2_f32(float *target0, const float *src0, size_t n0, double tau0, float *target1, const float *src1, size_t n1,
                   double tau1, augcl_stream_t stream)
{
    const char *name = "augcl_ema2_f32";
    if (n0 == 0 && n1 == 0) return AUGCL_OK;
    AUGCL_REQUIRE((n0 == 0 || (target0 && src0)) && (n1 == 0 || (target1 && src1)), "%s: null buffer", name);
    AUGCL_REQUIRE(aligned4(target0) && aligned4(src0) && aligned4(target1) && aligned4(src1), "%s: misaligned buffer", name);
    const size_t nmax = n0 > n1 ? n0 : n1;
    const size_t blocks = (nmax / 4 + 255) / 256 + 1;
    hipLaunchKernelGGL(k_ema_f32, dim3((unsigned)(blocks < 4096 ? blocks : 4096), n1 ? 2 : 1), dim3(256), 0,
                       as_stream(stream), target0, src0, n0, (float)tau0, (float)(1.0 - tau0), target1, src1, n1,
                       (float)tau1, (float)(1.0 - tau1));
    return augcl::check_launch(name);
}

int augcl_ema_f32(float *target, const float *src, size_t n, double tau, augcl_stream_t stream)
{
    if (n == 0) return AUGCL_OK;
    if (!target || !src) return augcl::fail(AUGCL_E_INVALID, "augcl_ema_f32: null buffer");
    return augcl_ema2_f32(target, src, n, tau, nullptr, nullptr, 0, 0.0, stream);
}

} // extern "C"
