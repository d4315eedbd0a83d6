"""The SAC heads' element-wise tails as single HIP launches (include/augcl.h: augcl_gauss_head_*, augcl_td_target_f32,
augcl_q_loss_*): the squashed-Gaussian policy head (src/algorithms/modules.py:19-32, 225-244), the TD target
(src/algorithms/sac.py:94-101) and the twin-Q MSE loss (sac.py:108-110, svea.py:37-56).  The reference spends ~110
small ATen launches per update on these (B, A)-sized tensors; here each is one launch forward and one backward,
with the reference's operation order inside.  Callers keep the literal ATen chain only for tensors these launches do not
cover (another dtype, e.g. a float64 check of the same module)."""
import ctypes

import torch

from . import _lib


def _ok(*ts):
    return all(t.is_cuda and t.dtype == torch.float32 for t in ts)


class _GaussHead(torch.autograd.Function):
    @staticmethod
    def forward(ctx, raw, noise, lo, hi, want_log_pi):
        raw = raw.contiguous()
        B, A2 = raw.shape
        A = A2 // 2
        dev = raw.device
        out = torch.empty((3 * B * A + B,), dtype=torch.float32, device=dev)  # one allocation for the four outputs
        mu_t, pi_t, log_std = (out[i * B * A:(i + 1) * B * A].view(B, A) for i in range(3))
        log_pi = out[3 * B * A:].view(B, 1)
        n = None if noise is None else noise.contiguous()
        _lib.call("augcl_gauss_head_forward", raw.data_ptr(), _lib.ptr(n), mu_t.data_ptr(),
                  pi_t.data_ptr() if n is not None else None, log_pi.data_ptr() if (n is not None and want_log_pi) else None,
                  log_std.data_ptr(), B, A, float(lo), float(hi), _lib.current_stream(dev))
        ctx.save_for_backward(raw, n)
        ctx.lohi = (float(lo), float(hi))
        ctx.mark_non_differentiable()
        return mu_t, (pi_t if n is not None else None), (log_pi if (n is not None and want_log_pi) else None), log_std

    @staticmethod
    def backward(ctx, g_mu, g_pi, g_lp, g_ls):
        raw, n = ctx.saved_tensors
        B, A2 = raw.shape
        c = lambda t: None if t is None else t.contiguous()  # noqa: E731
        g_mu, g_pi, g_lp, g_ls = c(g_mu), c(g_pi), c(g_lp), c(g_ls)
        g_raw = torch.empty_like(raw)
        _lib.call("augcl_gauss_head_backward", raw.data_ptr(), _lib.ptr(n), _lib.ptr(g_mu), _lib.ptr(g_pi), _lib.ptr(g_lp),
                  _lib.ptr(g_ls), g_raw.data_ptr(), B, A2 // 2, ctx.lohi[0], ctx.lohi[1], _lib.current_stream(raw.device))
        return g_raw, None, None, None, None


def gauss_head(raw, noise, log_std_min, log_std_max, want_log_pi=True):
    """(mu_t, pi_t | None, log_pi | None, log_std) from the policy MLP's (B, 2A) output and N(0,1) noise."""
    return _GaussHead.apply(raw, noise, log_std_min, log_std_max, want_log_pi)


class _LnTanh(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, eps):
        x = x.contiguous()
        B, D = x.shape
        y = torch.empty_like(x)
        _lib.call("augcl_ln_tanh_forward", x.data_ptr(), gamma.data_ptr(), beta.data_ptr(), y.data_ptr(), B, D, float(eps),
                  _lib.current_stream(x.device))
        ctx.save_for_backward(x, gamma, y)
        ctx.eps = float(eps)
        return y

    @staticmethod
    def backward(ctx, gy):
        x, gamma, y = ctx.saved_tensors
        B, D = x.shape
        gy = gy.contiguous()
        dx = torch.empty_like(x)
        dgb = torch.empty((2, D), dtype=torch.float32, device=x.device)
        ws = torch.empty(2 * B * D, dtype=torch.float32, device=x.device)
        _lib.call("augcl_ln_tanh_backward", x.data_ptr(), gamma.data_ptr(), y.data_ptr(), gy.data_ptr(), dx.data_ptr(),
                  dgb[0].data_ptr(), dgb[1].data_ptr(), ws.data_ptr(), B, D, ctx.eps, _lib.current_stream(x.device))
        return dx, dgb[0], dgb[1], None


def ln_tanh(x, gamma, beta, eps=1e-5):
    """tanh(LayerNorm(x) * gamma + beta) over the last dimension of a (B, D <= 256) tensor."""
    return _LnTanh.apply(x, gamma, beta, eps)


def td_target(tq1, tq2, log_pi, reward, not_done, alpha, discount):
    """reward + (not_done * discount) * (min(tq1, tq2) - alpha * log_pi); alpha: 0-dim float64 device tensor."""
    B = tq1.shape[0]
    out = torch.empty((B, 1), dtype=torch.float32, device=tq1.device)
    a = alpha.detach()
    if a.dtype != torch.float64:
        a = a.double()
    _lib.call("augcl_td_target_f32", tq1.contiguous().data_ptr(), tq2.contiguous().data_ptr(), log_pi.contiguous().data_ptr(),
              reward.contiguous().data_ptr(), not_done.contiguous().data_ptr(), a.data_ptr(), float(discount),
              out.data_ptr(), B, _lib.current_stream(tq1.device))
    return out


class _QLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, q1, q2, y, weights):
        q1, q2, y = q1.contiguous(), q2.contiguous(), y.contiguous()
        M, S = y.shape[0], len(weights)
        assert q1.shape[0] == M * S and q2.shape == q1.shape
        w = (ctypes.c_double * S)(*[float(v) for v in weights])
        out = torch.empty(1 + S, dtype=torch.float32, device=q1.device)
        _lib.call("augcl_q_loss_forward", q1.data_ptr(), q2.data_ptr(), y.data_ptr(), M, S,
                  ctypes.cast(w, ctypes.c_void_p), out.data_ptr(), _lib.current_stream(q1.device))
        ctx.save_for_backward(q1, q2, y)
        ctx.w = tuple(float(v) for v in weights)
        segs = out[1:]
        ctx.mark_non_differentiable(segs)
        return out[0], segs

    @staticmethod
    def backward(ctx, g_total, _g_segs):
        q1, q2, y = ctx.saved_tensors
        M, S = y.shape[0], len(ctx.w)
        w = (ctypes.c_double * S)(*ctx.w)
        dq1, dq2 = torch.empty_like(q1), torch.empty_like(q2)
        g = g_total.contiguous().float()
        _lib.call("augcl_q_loss_backward", q1.data_ptr(), q2.data_ptr(), y.data_ptr(), g.data_ptr(), M, S,
                  ctypes.cast(w, ctypes.c_void_p), dq1.data_ptr(), dq2.data_ptr(), _lib.current_stream(q1.device))
        return dq1, dq2, None, None


def q_loss(q1, q2, y, weights=(1.0,)):
    """(total, per-segment losses): segment s = mse(q1[sM:(s+1)M], y) + mse(q2[...], y), total = sum w[s] seg[s].
    No gradient reaches y (it is a detached target everywhere on this path)."""
    return _QLoss.apply(q1, q2, y, tuple(weights))
