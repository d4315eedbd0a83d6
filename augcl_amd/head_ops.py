"""The SAC heads' element-wise tails as single HIP launches (include/augcl.h: augcl_gauss_head_*, augcl_td_target_f32,
augcl_q_loss_*): the squashed-Gaussian policy head (src/algorithms/modules.py:19-32, 225-244), the TD target
(src/algorithms/sac.py:94-101) and the twin-Q MSE loss (sac.py:108-110, svea.py:37-56).  The reference spends ~110
small ATen launches per update on these (B, A)-sized tensors; here each is one launch forward and one backward,
with the reference's operation order inside.  Callers keep the literal ATen chain only for tensors these launches do not
cover (another dtype, e.g. a float64 check of the same module)."""
import ctypes

import weakref

import torch

from . import _lib


def _ok(*ts):
    return all(t.is_cuda and t.dtype == torch.float32 for t in ts)


# Test instrumentation: when set, called as gate_hook(kind, a1, a2) with the post-ReLU hidden activations of every fused
# MLP forward ("twin": a1 (B, 2H) = [Q1 | Q2], a2 (2, B, H); "mlp3": (B, H) each).  The ReLUs of these trunks run inside
# GEMM epilogues and HIP kernels, where no nn.ReLU forward hook can see them (tests/test_referee_gpu.py reads the gates).
gate_hook = None


def _rows(t, width):
    """(tensor, row stride in floats) for a (B, width) gradient the kernels can read in place: unit column stride, any
    row stride >= width (a column slice of a wider matrix); anything else is made contiguous first."""
    if t is None:
        return None, width
    if t.dim() == 2 and t.stride(1) == 1 and t.stride(0) >= width and t.dtype == torch.float32:
        return t, t.stride(0)
    return t.contiguous().float(), width


class _GaussHead(torch.autograd.Function):
    @staticmethod
    def forward(ctx, raw, noise, lo, hi, want_log_pi):
        ctx.set_materialize_grads(False)  # absent output gradients arrive as None, not as zero-filled tensors
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
        g_mu, g_lp, g_ls = c(g_mu), c(g_lp), c(g_ls)
        g_pi, ld = _rows(g_pi, A2 // 2)  # usually the action columns of the critic's input gradient: read in place
        g_raw = torch.empty_like(raw)
        _lib.call("augcl_gauss_head_backward", raw.data_ptr(), _lib.ptr(n), _lib.ptr(g_mu), _lib.ptr(g_pi), ld, _lib.ptr(g_lp),
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
        gy, ld = _rows(gy, D)
        dx = torch.empty_like(x)
        dgb = torch.empty((2, D), dtype=torch.float32, device=x.device)
        ws = torch.empty(2 * B * D, dtype=torch.float32, device=x.device)
        _lib.call("augcl_ln_tanh_backward", x.data_ptr(), gamma.data_ptr(), y.data_ptr(), gy.data_ptr(), ld, dx.data_ptr(),
                  dgb[0].data_ptr(), dgb[1].data_ptr(), None, ws.data_ptr(), B, D, ctx.eps, _lib.current_stream(x.device))
        return dx, dgb[0], dgb[1], None


class _ProjLnTanh(torch.autograd.Function):
    """RLProjection (src/algorithms/modules.py:101-111) as one node: forward = split-K f32-MFMA GEMM + (slab reduction,
    bias, LayerNorm, tanh) in two launches (augcl_proj_ln_tanh_forward); backward = the fused LN/tanh backward, whose
    column-sum launch also yields the Linear's bias gradient, a library GEMM for dW and, when dx is wanted,
    augcl_proj_dx_f32 (the output-bound GEMM with the 100-wide inner dimension)."""

    @staticmethod
    def forward(ctx, x, w, b, gamma, beta, eps, slabs, tail):
        ctx.set_materialize_grads(False)
        x = x.contiguous()
        M, K = x.shape
        N = w.shape[0]
        yz = torch.empty((2, M, N), dtype=torch.float32, device=x.device)
        ws = torch.empty(slabs * M * N, dtype=torch.float32, device=x.device)
        st = _lib.current_stream(x.device)
        ctx.eps, ctx.tail_rows = float(eps), 0
        if tail is None:
            _lib.call("augcl_proj_ln_tanh_forward", x.data_ptr(), w.data_ptr(), b.data_ptr(), gamma.data_ptr(), beta.data_ptr(),
                      yz[0].data_ptr(), yz[1].data_ptr(), ws.data_ptr(), ws.numel() * 4, M, K, N, float(eps), st)
            ctx.save_for_backward(x, w, gamma, yz)
            return yz[1]
        # the critic's input rows [z | action] leave the same launch (no torch.cat after it); tail rows repeat over M
        tail = tail.contiguous()
        A = tail.shape[1]
        xcat = torch.empty((M, N + A), dtype=torch.float32, device=x.device)
        _lib.call("augcl_proj_ln_tanh_cat_forward", x.data_ptr(), w.data_ptr(), b.data_ptr(), gamma.data_ptr(), beta.data_ptr(),
                  yz[0].data_ptr(), yz[1].data_ptr(), xcat.data_ptr(), tail.data_ptr(), A, tail.shape[0], ws.data_ptr(),
                  ws.numel() * 4, M, K, N, float(eps), st)
        ctx.save_for_backward(x, w, gamma, yz)
        ctx.tail_rows = tail.shape[0]
        return yz[1], xcat

    @staticmethod
    def backward(ctx, gz, gcat=None):
        x, w, gamma, yz = ctx.saved_tensors
        M, N = yz.shape[1], yz.shape[2]
        d_tail = None
        if gcat is not None:  # gradient of [z | tail]: its first N columns are z's, the rest the tail's (summed over repeats)
            if ctx.needs_input_grad[7]:
                d_tail = gcat[:, N:]
                if ctx.tail_rows != M:
                    d_tail = d_tail.reshape(M // ctx.tail_rows, ctx.tail_rows, -1).sum(0)
            gz = gcat[:, :N] if gz is None else gz + gcat[:, :N]
        if gz is None:
            return None, None, None, None, None, None, None, d_tail
        gz, ld = _rows(gz, N)  # usually the feature columns of the critic's input gradient: read in place
        dy = torch.empty((M, N), dtype=torch.float32, device=x.device)
        small = torch.empty((3, N), dtype=torch.float32, device=x.device)  # dgamma, dbeta, dbias
        ws = torch.empty(2 * M * N, dtype=torch.float32, device=x.device)
        _lib.call("augcl_ln_tanh_backward", yz[0].data_ptr(), gamma.data_ptr(), yz[1].data_ptr(), gz.data_ptr(), ld, dy.data_ptr(),
                  small[0].data_ptr(), small[1].data_ptr(), small[2].data_ptr(), ws.data_ptr(), M, N, ctx.eps,
                  _lib.current_stream(x.device))
        dw = dy.t().mm(x) if ctx.needs_input_grad[1] else None
        dx = None
        if ctx.needs_input_grad[0]:
            if N % 4 == 0 and N <= 128 and w.is_contiguous():  # the skinny-inner-dimension GEMM as its own MFMA kernel
                dx = torch.empty_like(x)
                _lib.call("augcl_proj_dx_f32", dy.data_ptr(), w.data_ptr(), dx.data_ptr(), M, N, x.shape[1],
                          _lib.current_stream(x.device))
            else:
                dx = dy.mm(w)
        return dx, dw, small[2], small[0], small[1], None, None, d_tail


def proj_ln_tanh(x, lin, ln, tail=None):
    """tanh(LayerNorm(Linear(x))) for an nn.Linear / nn.LayerNorm pair, or None when the fused kernels do not cover the
    shape (the caller then runs the Linear through the GEMM library and `ln_tanh`).  tail (R, A) with x's row count a
    multiple of R: returns (z, [z | tail rows repeated]) -- the critic's input, built by the same launch."""
    if lin.bias is None or x.dim() != 2 or not x.is_contiguous() and x.stride(-1) != 1:
        return None
    M, K = x.shape
    N = lin.weight.shape[0]
    if not lin.weight.is_contiguous() or x.data_ptr() % 16 or lin.weight.data_ptr() % 16:
        return None
    slabs = _lib.load().augcl_proj_ln_tanh_workspace_slabs(int(M), int(K), int(N))
    if slabs <= 0:
        return None
    if tail is not None and not (_ok(tail) and tail.dim() == 2 and N + tail.shape[1] <= 128 and tail.shape[0] >= 1
                                 and M % tail.shape[0] == 0):
        return None
    return _ProjLnTanh.apply(x, lin.weight, lin.bias, ln.weight, ln.bias, ln.eps, slabs, tail)


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
        ctx.set_materialize_grads(False)
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
        if g_total is None:
            return None, None, None, None
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


def _mask_colsum(dz, a, want_db, extra=None):
    """dz <- (a > 0) ? dz : 0 in place (ReLU's backward); returns (column sums of the gated dz | None, column sums of
    `extra` | None): the bias gradients of the Linear under the ReLU and -- riding along -- of the output layer."""
    B, N = dz.shape
    E = 0 if extra is None else extra.shape[1]
    sums = torch.empty(N + E, dtype=torch.float32, device=dz.device) if (want_db or E) else None
    _lib.call("augcl_mask_colsum", dz.data_ptr(), a.data_ptr(), sums.data_ptr() if want_db else None, _lib.ptr(extra),
              (sums.data_ptr() + 4 * N) if E else None, B, N, E, _lib.current_stream(dz.device))
    return (sums[:N] if want_db else None), (sums[N:] if E else None)


_twin_cache = {}


def _twin_views(*ps):
    """The six stacked (2, ...) tensors of a twin's 12 parameters.  When all of them are VIEWS of one flat buffer (the
    agents' layout) they are built once per parameter storage and reused: building them is ~30 us of host time per call, and
    the bf16 mode without hipGraphs is host-bound.  (Copies -- parameters that are not neighbours in one storage -- are
    rebuilt every call: they go stale with the next optimiser step.)"""
    key = tuple(p.data_ptr() for p in ps)
    hit = _twin_cache.get(key)
    if hit is not None and hit[0]() is ps[0]:  # (same addresses AND the same live parameter object: a freed agent's buffer
        return hit[1]                          # may be handed out again at the same address)
    a, b = ps[:6], ps[6:]
    out = [_stacked(p, q) for p, q in zip(a, b)]
    views = all(t.data_ptr() == p.data_ptr() for t, p in zip(out, a))
    if not (out[0].is_contiguous() and out[1].is_contiguous()):  # padding between the twins' tensors: stacked copies
        out[0], out[1] = out[0].contiguous(), out[1].contiguous()
        views = False
    out = tuple(out)
    if views:
        # the cached views keep their flat buffer alive: entries of agents that are gone (their parameter objects have been
        # collected) are dropped here, so a discarded agent's device memory is released by the next agent's first forward
        for k in [k for k, (ref, _) in _twin_cache.items() if ref() is None]:
            del _twin_cache[k]
        try:
            _twin_cache[key] = (weakref.ref(ps[0]), out)
        except TypeError:  # a plain tensor that cannot be weakly referenced: not cached
            pass
    return out


class _TwinQ(torch.autograd.Function):
    """Q1(h, a), Q2(h, a) of src/algorithms/modules.py:247-276 -- two identical 3-layer MLPs over the same input -- as ONE
    batched network.  Forward, 4 launches: concat, ONE GEMM for both first layers with bias + ReLU in its epilogue
    ((B, D) x (D, 2H), torch._addmm_activation), the batched second-layer GEMM, and augcl_mlp_out_forward (second bias +
    ReLU in place, the H -> 1 output layer as a fixed-order dot product per row).  Backward, 6 launches (4 when only the
    input gradient is wanted): augcl_mlp_out_backward (dz2 and the sums dW3, db3, db2), the batched GEMM to layer 1 written
    in the (B, 2H) layout, augcl_mask_colsum (ReLU gate + db1), then plain GEMMs over the 2H dimension for dx (both
    trunks' contributions summed inside the GEMM), dW1 and the batched dW2.  Two nn.Sequential trunks are 14 + 26."""

    @staticmethod
    def forward(ctx, h, a, w1a, b1a, w2a, b2a, w3a, b3a, w1b, b1b, w2b, b2b, w3b, b3b):
        ctx.set_materialize_grads(False)
        B, H = h.shape[0], w1a.shape[0]
        x = h if a is None else torch.cat([h, a], dim=1)  # a = None: h already is [features | action]
        W1, b1, W2, b2, W3, b3 = _twin_views(w1a, b1a, w2a, b2a, w3a, b3a, w1b, b1b, w2b, b2b, w3b, b3b)
        # bias + ReLU in the GEMM's epilogue (one launch): torch._addmm_activation(bias, x, W^T) = relu(x W^T + bias)
        a1 = torch._addmm_activation(b1.reshape(2 * H), x, W1.reshape(2 * H, -1).t())  # (B, 2H) = [Q1 | Q2]
        z2 = torch.bmm(a1.view(B, 2, H).transpose(0, 1), W2.transpose(1, 2))            # (2, B, H), no bias yet
        q = torch.empty((2, B, 1), dtype=torch.float32, device=h.device)
        _lib.call("augcl_mlp_out_forward", z2.data_ptr(), b2.data_ptr(), b2.stride(0), W3.data_ptr(), W3.stride(0),
                  b3.data_ptr(), b3.stride(0), q.data_ptr(), 2, B, H, _lib.current_stream(h.device))
        if gate_hook is not None:
            gate_hook("twin", a1, z2)
        ctx.save_for_backward(x, a1, z2, W1, W2, W3)  # z2 now holds a2 = relu(z2 + b2)
        ctx.n_h = h.shape[1] if a is not None else -1
        return q[0], q[1]

    @staticmethod
    def backward(ctx, g1, g2):
        x, a1, a2, W1, W2, W3 = ctx.saved_tensors
        B, H = x.shape[0], a2.shape[2]
        dev = x.device
        want_w = any(ctx.needs_input_grad[2:])
        want_x = ctx.needs_input_grad[0] or ctx.needs_input_grad[1]
        if g1 is None and g2 is None:
            return (None,) * 14
        g1 = torch.zeros((B, 1), dtype=torch.float32, device=dev) if g1 is None else g1.contiguous()
        g2 = torch.zeros((B, 1), dtype=torch.float32, device=dev) if g2 is None else g2.contiguous()
        dz2 = torch.empty_like(a2)
        small = torch.empty(2 * H + 2 + 2 * H, dtype=torch.float32, device=dev) if want_w else None  # dW3 | db3 | db2
        sp = (lambda off: small.data_ptr() + 4 * off) if want_w else (lambda off: None)
        _lib.call("augcl_mlp_out_backward", g1.data_ptr(), g2.data_ptr(), a2.data_ptr(), W3.data_ptr(), W3.stride(0),
                  dz2.data_ptr(), sp(0), sp(2 * H), sp(2 * H + 2), 2, B, H, _lib.current_stream(dev))
        dz1 = torch.empty((B, 2 * H), dtype=torch.float32, device=dev)  # [Q1 | Q2] per row, like a1
        torch.bmm(dz2, W2, out=dz1.view(B, 2, H).transpose(0, 1))
        db1, _ = _mask_colsum(dz1, a1, want_w)
        dx = dz1.mm(W1.reshape(2 * H, -1)) if want_x else None  # the sum over the two trunks happens inside the GEMM
        gw = [None] * 12
        if want_w:
            dW1 = dz1.t().mm(x).view(2, H, -1)
            dW2 = torch.bmm(dz2.transpose(1, 2), a1.view(B, 2, H).transpose(0, 1))
            dW3, db3, db2 = small[:2 * H].view(2, 1, H), small[2 * H:2 * H + 2].view(2, 1), small[2 * H + 2:].view(2, H)
            db1 = db1.view(2, H)
            for t in (0, 1):
                gw[6 * t:6 * t + 6] = [dW1[t], db1[t], dW2[t], db2[t], dW3[t], db3[t]]
        if ctx.n_h < 0:  # pre-concatenated input: one gradient for the whole row
            return (dx, None, *gw)
        dh = dx[:, :ctx.n_h] if dx is not None and ctx.needs_input_grad[0] else None
        da = dx[:, ctx.n_h:] if dx is not None and ctx.needs_input_grad[1] else None
        return (dh, da, *gw)


def twin_q(h, a, q1, q2, freeze=False):
    """(Q1(h, a), Q2(h, a)) for two modules.QFunction with 3-layer trunks (Linear, ReLU, Linear, ReLU, Linear).
    a = None: h already holds the concatenated rows [features | action] (proj_ln_tanh(..., tail=action)).
    freeze: the trunks' weights take part as constants (the actor step, src/algorithms/sac.py:119-133: the reference
    builds their gradients there and throws them away at the next critic zero_grad)."""
    t1, t2 = q1.trunk, q2.trunk
    ps = (t1[0].weight, t1[0].bias, t1[2].weight, t1[2].bias, t1[4].weight, t1[4].bias,
          t2[0].weight, t2[0].bias, t2[2].weight, t2[2].bias, t2[4].weight, t2[4].bias)
    if freeze:
        ps = tuple(p.detach() for p in ps)
    return _TwinQ.apply(h, a, *ps)


class _Mlp3(torch.autograd.Function):
    """Linear-ReLU-Linear-ReLU-Linear (Actor.mlp, src/algorithms/modules.py:200-215) as one node: forward = three library
    GEMMs, bias + ReLU in the epilogue of the first two; backward = the five GEMMs a 3-layer MLP needs (six with dx) and
    two augcl_mask_colsum launches that gate dz AND deliver all three bias gradients.  3 + 8 launches instead of 5 + 13."""

    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2, w3, b3):
        ctx.set_materialize_grads(False)
        x = x.contiguous()
        a1 = torch._addmm_activation(b1, x, w1.t())
        a2 = torch._addmm_activation(b2, a1, w2.t())
        out = torch.addmm(b3, a2, w3.t())
        if gate_hook is not None:
            gate_hook("mlp3", a1, a2)
        ctx.save_for_backward(x, a1, a2, w1, w2, w3)
        return out

    @staticmethod
    def backward(ctx, g):
        if g is None:
            return (None,) * 7
        x, a1, a2, w1, w2, w3 = ctx.saved_tensors
        g = g.contiguous()
        want_w = any(ctx.needs_input_grad[1:])
        dz2 = g.mm(w3)
        db2, db3 = _mask_colsum(dz2, a2, want_w, g if want_w else None)
        dz1 = dz2.mm(w2)
        db1, _ = _mask_colsum(dz1, a1, want_w)
        dx = dz1.mm(w1) if ctx.needs_input_grad[0] else None
        if not want_w:
            return (dx,) + (None,) * 6
        return dx, dz1.t().mm(x), db1, dz2.t().mm(a1), db2, g.t().mm(a2), db3


def mlp3(x, seq):
    """seq(x) for an nn.Sequential(Linear, ReLU, Linear, ReLU, Linear) on float32 device tensors."""
    return _Mlp3.apply(x, seq[0].weight, seq[0].bias, seq[2].weight, seq[2].bias, seq[4].weight, seq[4].bias)
