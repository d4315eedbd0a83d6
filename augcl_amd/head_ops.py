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
                  dgb[0].data_ptr(), dgb[1].data_ptr(), None, ws.data_ptr(), B, D, ctx.eps, _lib.current_stream(x.device))
        return dx, dgb[0], dgb[1], None


class _ProjLnTanh(torch.autograd.Function):
    """RLProjection (src/algorithms/modules.py:101-111) as one node: forward = split-K f32-MFMA GEMM + (slab reduction,
    bias, LayerNorm, tanh) in two launches (augcl_proj_ln_tanh_forward); backward = the fused LN/tanh backward, whose
    column-sum launch also yields the Linear's bias gradient, and two library GEMMs (dW, and dx when it is wanted)."""

    @staticmethod
    def forward(ctx, x, w, b, gamma, beta, eps, slabs):
        x = x.contiguous()
        M, K = x.shape
        N = w.shape[0]
        yz = torch.empty((2, M, N), dtype=torch.float32, device=x.device)
        ws = torch.empty(slabs * M * N, dtype=torch.float32, device=x.device)
        _lib.call("augcl_proj_ln_tanh_forward", x.data_ptr(), w.data_ptr(), b.data_ptr(), gamma.data_ptr(), beta.data_ptr(),
                  yz[0].data_ptr(), yz[1].data_ptr(), ws.data_ptr(), ws.numel() * 4, M, K, N, float(eps),
                  _lib.current_stream(x.device))
        ctx.save_for_backward(x, w, gamma, yz)
        ctx.eps = float(eps)
        return yz[1]

    @staticmethod
    def backward(ctx, gz):
        x, w, gamma, yz = ctx.saved_tensors
        M, N = yz.shape[1], yz.shape[2]
        gz = gz.contiguous()
        dy = torch.empty((M, N), dtype=torch.float32, device=x.device)
        small = torch.empty((3, N), dtype=torch.float32, device=x.device)  # dgamma, dbeta, dbias
        ws = torch.empty(2 * M * N, dtype=torch.float32, device=x.device)
        _lib.call("augcl_ln_tanh_backward", yz[0].data_ptr(), gamma.data_ptr(), yz[1].data_ptr(), gz.data_ptr(), dy.data_ptr(),
                  small[0].data_ptr(), small[1].data_ptr(), small[2].data_ptr(), ws.data_ptr(), M, N, ctx.eps,
                  _lib.current_stream(x.device))
        dw = dy.t().mm(x) if ctx.needs_input_grad[1] else None
        dx = dy.mm(w) if ctx.needs_input_grad[0] else None
        return dx, dw, small[2], small[0], small[1], None, None


def proj_ln_tanh(x, lin, ln):
    """tanh(LayerNorm(Linear(x))) for an nn.Linear / nn.LayerNorm pair, or None when the fused kernels do not cover the
    shape (the caller then runs the Linear through the GEMM library and `ln_tanh`)."""
    if lin.bias is None or x.dim() != 2 or not x.is_contiguous() and x.stride(-1) != 1:
        return None
    M, K = x.shape
    N = lin.weight.shape[0]
    if not lin.weight.is_contiguous() or x.data_ptr() % 16 or lin.weight.data_ptr() % 16:
        return None
    slabs = _lib.load().augcl_proj_ln_tanh_workspace_slabs(int(M), int(K), int(N))
    if slabs <= 0:
        return None
    return _ProjLnTanh.apply(x, lin.weight, lin.bias, ln.weight, ln.bias, ln.eps, slabs)


def ln_tanh(x, gamma, beta, eps=1e-5):
    """tanh(LayerNorm(x) * gamma + beta) over the last dimension of a (B, D <= 256) tensor."""
    return _LnTanh.apply(x, gamma, beta, eps)


def td_target(tq1, tq2, log_pi, reward, not_done, alpha, discount, log_alpha=None):
    """reward + (not_done * discount) * (min(tq1, tq2) - alpha * log_pi); alpha: 0-dim float64 device tensor, or
    log_alpha= the 0-dim float64 tensor it is the exponential of (saves the exp launch: the kernel takes it)."""
    B = tq1.shape[0]
    out = torch.empty((B, 1), dtype=torch.float32, device=tq1.device)
    a = (alpha if log_alpha is None else log_alpha).detach()
    if a.dtype != torch.float64:
        a = a.double()
    _lib.call("augcl_td_target_f32", tq1.contiguous().data_ptr(), tq2.contiguous().data_ptr(), log_pi.contiguous().data_ptr(),
              reward.contiguous().data_ptr(), not_done.contiguous().data_ptr(), a.data_ptr(), 0 if log_alpha is None else 1,
              float(discount), out.data_ptr(), B, _lib.current_stream(tq1.device))
    return out


class _ActorLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, log_pi, q1, q2, log_alpha):
        log_pi, q1, q2 = log_pi.contiguous(), q1.contiguous(), q2.contiguous()
        B = log_pi.shape[0]
        out = torch.empty(1, dtype=torch.float32, device=log_pi.device)
        la = log_alpha.detach()
        _lib.call("augcl_actor_loss_forward", log_pi.data_ptr(), q1.data_ptr(), q2.data_ptr(), la.data_ptr(), out.data_ptr(), B,
                  _lib.current_stream(log_pi.device))
        ctx.save_for_backward(q1, q2, la)
        return out[0]

    @staticmethod
    def backward(ctx, g):
        q1, q2, la = ctx.saved_tensors
        B = q1.shape[0]
        buf = torch.empty((3, B, 1), dtype=torch.float32, device=q1.device)
        g = g.contiguous().float()
        _lib.call("augcl_actor_loss_backward", q1.data_ptr(), q2.data_ptr(), la.data_ptr(), g.data_ptr(), buf[0].data_ptr(),
                  buf[1].data_ptr(), buf[2].data_ptr(), B, _lib.current_stream(q1.device))
        return buf[0], buf[1], buf[2], None


def actor_loss(log_pi, q1, q2, log_alpha):
    """mean(alpha.detach() * log_pi - min(q1, q2)) with alpha = exp(log_alpha) (src/algorithms/sac.py:119-131): one launch
    forward, one backward; log_alpha (0-dim float64) receives no gradient, exactly like the reference's .detach()."""
    assert log_alpha.dtype == torch.float64 and log_pi.shape[0] == q1.shape[0] == q2.shape[0]
    return _ActorLoss.apply(log_pi, q1, q2, log_alpha)


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


# ---- twin Q trunks as ONE batched network -----------------------------------------------------------------------------
def _stacked(p1, p2):
    """(2, *shape) view over two same-shaped tensors that live at a fixed distance in ONE storage (the flat parameter
    buffer lays Q1.x / Q2.x out next to each other, flat.critic_order); a stacked COPY otherwise."""
    p1, p2 = p1.detach(), p2.detach()
    if (p1.untyped_storage().data_ptr() == p2.untyped_storage().data_ptr() and p1.is_contiguous() and p2.is_contiguous()):
        delta = p2.storage_offset() - p1.storage_offset()
        if delta >= p1.numel():
            return torch.as_strided(p1, (2, *p1.shape), (delta, *p1.stride()))
    return torch.stack([p1, p2])


class _TwinQ(torch.autograd.Function):
    """Q1(h, a), Q2(h, a) of src/algorithms/modules.py:247-276 -- two identical 3-layer MLPs over the same input -- as ONE
    batched network: layer 1 is a single GEMM against the two weight matrices side by side, layers 2 and 3 are batched
    GEMMs over the twin dimension, forward and backward.  6 launches forward / 13 backward (7 when only the input
    gradient is wanted: the actor step) instead of 14 / 26 for two nn.Sequential trunks; the same additions in the same
    order inside every dot product's GEMM, so values agree with the two-trunk form to GEMM rounding."""

    @staticmethod
    def forward(ctx, h, a, w1a, b1a, w2a, b2a, w3a, b3a, w1b, b1b, w2b, b2b, w3b, b3b):
        B, H = h.shape[0], w1a.shape[0]
        x = torch.cat([h, a], dim=1)
        W1, b1, W2, b2, W3, b3 = (_stacked(p, q) for p, q in ((w1a, w1b), (b1a, b1b), (w2a, w2b), (b2a, b2b), (w3a, w3b), (b3a, b3b)))
        if W1.is_contiguous() and b1.is_contiguous():  # one GEMM for both first layers: (B, D) x (D, 2H)
            # bias + ReLU in the GEMM's epilogue (one launch): torch._addmm_activation(bias, x, W^T) = relu(x W^T + bias)
            a1 = torch._addmm_activation(b1.reshape(2 * H), x, W1.reshape(2 * H, -1).t()).view(B, 2, H).transpose(0, 1)
        else:
            a1 = torch.relu_(torch.baddbmm(b1.unsqueeze(1), x.unsqueeze(0).expand(2, B, -1), W1.transpose(1, 2)))
        a2 = torch.relu_(torch.baddbmm(b2.unsqueeze(1), a1, W2.transpose(1, 2)))
        q = torch.baddbmm(b3.unsqueeze(1), a2, W3.transpose(1, 2))  # (2, B, 1)
        ctx.save_for_backward(x, a1, a2, W1, W2, W3)
        ctx.n_h = h.shape[1]
        return q[0], q[1]

    @staticmethod
    def backward(ctx, g1, g2):
        x, a1, a2, W1, W2, W3 = ctx.saved_tensors
        B = x.shape[0]
        want_w = any(ctx.needs_input_grad[2:])
        dq = torch.stack([g1, g2])                                       # (2, B, 1)
        dz2 = torch.ops.aten.threshold_backward(torch.bmm(dq, W3), a2, 0)  # (2, B, H): ReLU gate of layer 2
        dz1 = torch.ops.aten.threshold_backward(torch.bmm(dz2, W2), a1, 0)
        dx = torch.bmm(dz1, W1).sum(0) if (ctx.needs_input_grad[0] or ctx.needs_input_grad[1]) else None
        gw = [None] * 12
        if want_w:
            dW3, db3 = torch.bmm(dq.transpose(1, 2), a2), dq.sum(1)
            dW2, db2 = torch.bmm(dz2.transpose(1, 2), a1), dz2.sum(1)
            dW1, db1 = torch.bmm(dz1.transpose(1, 2), x.unsqueeze(0).expand(2, B, -1)), dz1.sum(1)
            for t in (0, 1):
                gw[6 * t:6 * t + 6] = [dW1[t], db1[t], dW2[t], db2[t], dW3[t], db3[t]]
        dh = dx[:, :ctx.n_h] if dx is not None and ctx.needs_input_grad[0] else None
        da = dx[:, ctx.n_h:] if dx is not None and ctx.needs_input_grad[1] else None
        return (dh, da, *gw)


def twin_q(h, a, q1, q2):
    """(Q1(h, a), Q2(h, a)) for two modules.QFunction with 3-layer trunks (Linear, ReLU, Linear, ReLU, Linear)."""
    t1, t2 = q1.trunk, q2.trunk
    return _TwinQ.apply(h, a, t1[0].weight, t1[0].bias, t1[2].weight, t1[2].bias, t1[4].weight, t1[4].bias,
                        t2[0].weight, t2[0].bias, t2[2].weight, t2[2].bias, t2[4].weight, t2[4].bias)
