"""GPU: fused SAC-head launches (augcl_amd/head_ops.py) against the literal ATen chains of the reference
(src/algorithms/modules.py:19-32, 225-244; sac.py:94-110; svea.py:37-56), values and gradients."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from augcl_amd import head_ops
from augcl_amd.algorithms import modules as m

pytestmark = pytest.mark.gpu


def ref_head(raw, noise, lo, hi):
    mu, log_std = raw.chunk(2, dim=-1)
    log_std = torch.tanh(log_std)
    log_std = lo + 0.5 * (hi - lo) * (log_std + 1)
    if noise is None:
        return torch.tanh(mu), None, None, log_std
    pi = mu + noise * log_std.exp()
    log_pi = m.gaussian_logprob(noise, log_std)
    mu, pi, log_pi = m.squash(mu, pi, log_pi)
    return mu, pi, log_pi, log_std


@pytest.mark.parametrize("B,A", [(1, 1), (7, 6), (256, 6), (512, 21)])
def test_gauss_head_matches_aten_chain(cuda, B, A):
    g = torch.Generator(device=cuda).manual_seed(B + A)
    raw0 = torch.randn(B, 2 * A, device=cuda, generator=g) * 2.0
    raw0[0, 0] = 9.0  # saturated tanh: relu(1 - pi^2) hits its floor
    noise = torch.randn(B, A, device=cuda, generator=g)
    # Near tanh saturation the chain is ill-conditioned (1 - tanh(pi)^2 cancels; one ulp of tanh moves the gradient
    # by ~1e-4 relative), so the yardstick is the same chain in float64: the fused launch must be as close to it
    # as the f32 ATen chain is.
    outs = []
    for mode in ("f64", "aten", "fused"):
        dt = torch.float64 if mode == "f64" else torch.float32
        raw = raw0.to(dt).clone().requires_grad_(True)
        o = head_ops.gauss_head(raw, noise, -10.0, 2.0) if mode == "fused" else ref_head(raw, noise.to(dt), -10.0, 2.0)
        gs = [torch.randn(t.shape, device=cuda, generator=torch.Generator(device=cuda).manual_seed(i)).to(dt) for i, t in enumerate(o)]
        sum((t * w).sum() for t, w in zip(o, gs)).backward()
        outs.append([t.detach().double() for t in o] + [raw.grad.double()])
    for truth, aten, fused in zip(*outs):
        scale = max(1.0, float(truth.abs().max()))
        e_aten, e_fused = float((aten - truth).abs().max()), float((fused - truth).abs().max())
        assert e_fused <= max(2.0 * e_aten, 2e-6 * scale), (e_fused, e_aten)
    # only some outputs used (actor loss uses pi and log_pi; the entropy log uses log_std)
    for use in ([1, 2], [0], [3], [2]):
        grads = []
        for fused in (False, True):
            raw = raw0.clone().requires_grad_(True)
            o = head_ops.gauss_head(raw, noise, -10.0, 2.0) if fused else ref_head(raw, noise, -10.0, 2.0)
            sum(o[i].sum() for i in use).backward()
            grads.append(raw.grad)
        assert float((grads[0] - grads[1]).abs().max()) <= 2e-4 * max(1.0, float(grads[0].abs().max())), use
    # compute_pi = False
    mu_t, pi, lp, ls = head_ops.gauss_head(raw0, None, -10.0, 2.0, False)
    r = ref_head(raw0, None, -10.0, 2.0)
    assert pi is None and lp is None and torch.allclose(mu_t, r[0], atol=1e-6) and torch.allclose(ls, r[3], atol=1e-5)


def test_td_target_and_q_loss(cuda):
    g = torch.Generator(device=cuda).manual_seed(5)
    B = 256
    tq1, tq2, lp = (torch.randn(B, 1, device=cuda, generator=g) * 30 for _ in range(3))
    rew, nd = torch.rand(B, 1, device=cuda, generator=g) * 4, (torch.rand(B, 1, device=cuda, generator=g) > 0.1).float()
    alpha = torch.tensor(np.log(0.1), device=cuda).exp()  # float64 0-dim, like SAC.alpha
    want = rew + (nd * 0.99 * (torch.min(tq1, tq2) - alpha.detach() * lp))
    got = head_ops.td_target(tq1, tq2, lp, rew, nd, alpha, 0.99)
    assert got.shape == (B, 1) and float((got - want).abs().max()) <= 1e-5 * float(want.abs().max())

    y = want
    for S, w in ((1, (1.0,)), (2, (0.5, 0.5)), (2, (0.3, 0.9))):
        q0 = torch.randn(S * B, 1, device=cuda, generator=g) * 30
        res = []
        for fused in (False, True):
            q1, q2 = q0.clone().requires_grad_(True), (q0 * 0.9 + 1).detach().requires_grad_(True)
            if fused:
                total, segs = head_ops.q_loss(q1, q2, y, w)
            else:
                segs = torch.stack([F.mse_loss(q1[s * B:(s + 1) * B], y) + F.mse_loss(q2[s * B:(s + 1) * B], y) for s in range(S)])
                total = sum(w[s] * segs[s] for s in range(S))
            (total * 1.7).backward()
            res.append((total.detach(), segs.detach(), q1.grad, q2.grad))
        for a, b in zip(*res):
            assert float((a - b).abs().max()) <= 1e-5 * max(1.0, float(a.abs().max())), (S, w)


@pytest.mark.parametrize("B,D", [(1, 100), (7, 50), (512, 100), (33, 256)])
def test_ln_tanh_matches_aten(cuda, B, D):
    """RLProjection's tail (modules.py:101-111): tanh(LayerNorm(x)) forward and all three gradients."""
    g = torch.Generator(device=cuda).manual_seed(B * 1000 + D)
    x0 = torch.randn(B, D, device=cuda, generator=g) * 3.0
    gamma0 = torch.rand(D, device=cuda, generator=g) + 0.5
    beta0 = torch.randn(D, device=cuda, generator=g) * 0.1
    gy = torch.randn(B, D, device=cuda, generator=g)
    res = []
    for mode in ("f64", "aten", "fused"):
        dt = torch.float64 if mode == "f64" else torch.float32
        x, ga, be = (t.to(dt).clone().requires_grad_(True) for t in (x0, gamma0, beta0))
        y = head_ops.ln_tanh(x, ga, be, 1e-5) if mode == "fused" else torch.tanh(F.layer_norm(x, (D,), ga, be, 1e-5))
        (y * gy.to(dt)).sum().backward()
        res.append([t.detach().double() for t in (y, x.grad, ga.grad, be.grad)])
    for truth, aten, fused in zip(*res):
        scale = max(1.0, float(truth.abs().max()))
        e_aten, e_fused = float((aten - truth).abs().max()), float((fused - truth).abs().max())
        assert e_fused <= max(3.0 * e_aten, 3e-6 * scale), (e_fused, e_aten)


@pytest.mark.parametrize("B,A,flat_layout", [(1, 1, True), (37, 6, True), (512, 6, True), (64, 2, False)])
def test_twin_q_matches_two_trunks(cuda, B, A, flat_layout):
    """Q1, Q2 (src/algorithms/modules.py:247-276) as ONE batched network (head_ops.twin_q: stacked views of the flat
    parameter buffer, one GEMM for both first layers, batched GEMMs after) against the two nn.Sequential trunks:
    values, every parameter gradient and the input gradients, judged against the same trunks in float64."""
    from augcl_amd import flat
    torch.manual_seed(B + A)
    D, Hd = 100 + A, 1024
    q1, q2 = m.QFunction(100, A, Hd).to(cuda), m.QFunction(100, A, Hd).to(cuda)
    for p in list(q1.parameters()) + list(q2.parameters()):
        p.data.add_(0.05 * torch.randn_like(p))  # non-zero biases
    if flat_layout:  # what the agents do: every (Q1.x, Q2.x) pair at a fixed distance in one buffer
        named = [(f"Q1.{n}", p) for n, p in q1.named_parameters()] + [(f"Q2.{n}", p) for n, p in q2.named_parameters()]
        grp = flat.FlatGroup(flat.critic_order(named))
        assert grp.names[:4] == ["Q1.trunk.0.weight", "Q2.trunk.0.weight", "Q1.trunk.0.bias", "Q2.trunk.0.bias"]
        W = head_ops._stacked(q1.trunk[0].weight, q2.trunk[0].weight)
        assert W.data_ptr() == q1.trunk[0].weight.data_ptr() and W.is_contiguous()  # a view, not a copy
    g = torch.Generator(device=cuda).manual_seed(1)
    h0 = torch.tanh(torch.randn(B, 100, device=cuda, generator=g))
    a0 = torch.rand(B, A, device=cuda, generator=g) * 2 - 1
    go1, go2 = torch.randn(B, 1, device=cuda, generator=g), torch.randn(B, 1, device=cuda, generator=g)
    res = []
    for mode in ("f64", "aten", "fused"):
        dt = torch.float64 if mode == "f64" else torch.float32
        ps = [p.detach().to(dt).clone().requires_grad_(True) for p in list(q1.parameters()) + list(q2.parameters())]
        h, a = h0.to(dt).clone().requires_grad_(True), a0.to(dt).clone().requires_grad_(True)
        if mode == "fused":
            for p in list(q1.parameters()) + list(q2.parameters()):
                p.grad = None
            o1, o2 = head_ops.twin_q(h, a, q1, q2)
        else:
            def trunk(x, w):
                x = torch.relu(F.linear(x, w[0], w[1]))
                x = torch.relu(F.linear(x, w[2], w[3]))
                return F.linear(x, w[4], w[5])
            x = torch.cat([h, a], dim=1)
            o1, o2 = trunk(x, ps[:6]), trunk(x, ps[6:])
        ((o1 * go1.to(dt)).sum() + (o2 * go2.to(dt)).sum() * 0.7).backward()
        grads = [p.grad for p in (list(q1.parameters()) + list(q2.parameters()) if mode == "fused" else ps)]
        res.append([t.detach().double() for t in [o1, o2, h.grad, a.grad] + grads])
    for i, (truth, aten, fused) in enumerate(zip(*res)):
        scale = max(1e-3, float(truth.abs().max()))
        e_aten, e_fused = float((aten - truth).abs().max()), float((fused - truth).abs().max())
        assert e_fused <= max(3.0 * e_aten, 3e-6 * scale), (i, e_fused, e_aten)
    # only the input gradient (the actor step: critic weights are not differentiated)
    h = h0.clone().requires_grad_(True)
    o1, o2 = head_ops.twin_q(h, a0, q1, q2)
    (gh,) = torch.autograd.grad((o1 * go1).sum() + (o2 * go2).sum() * 0.7, [h])
    assert torch.equal(gh.double(), res[2][2])  # the same launches as the full backward's input-gradient chain


def test_twin_q_frozen_weights_same_input_gradient(cuda):
    """freeze=True (the actor step): same values and input gradients, no weight gradient is built."""
    torch.manual_seed(5)
    B, A = 96, 4
    q1, q2 = m.QFunction(50, A, 256).to(cuda), m.QFunction(50, A, 256).to(cuda)
    h0, a0 = torch.tanh(torch.randn(B, 50, device=cuda)), torch.rand(B, A, device=cuda) * 2 - 1
    out = []
    for freeze in (False, True):
        h, a = h0.clone().requires_grad_(True), a0.clone().requires_grad_(True)
        for p in list(q1.parameters()) + list(q2.parameters()):
            p.grad = None
        o1, o2 = head_ops.twin_q(h, a, q1, q2, freeze=freeze)
        (o1.sum() - 2 * o2.sum()).backward()
        out.append((o1.detach(), o2.detach(), h.grad, a.grad))
        assert all((p.grad is None) == freeze for p in list(q1.parameters()) + list(q2.parameters()))
    for x, y in zip(*out):
        assert torch.equal(x, y)


@pytest.mark.parametrize("B,D,Hd,E", [(256, 50, 1024, 12), (37, 50, 96, 2), (512, 100, 1024, 6)])
def test_mlp3_matches_sequential(cuda, B, D, Hd, E):
    """Actor.mlp (src/algorithms/modules.py:200-215) as one autograd node (head_ops.mlp3: GEMM-epilogue ReLUs forward,
    augcl_mask_colsum backward) against the nn.Sequential it replaces, both judged against float64."""
    torch.manual_seed(B + E)
    seq = torch.nn.Sequential(torch.nn.Linear(D, Hd), torch.nn.ReLU(), torch.nn.Linear(Hd, Hd), torch.nn.ReLU(),
                              torch.nn.Linear(Hd, E)).to(cuda)
    for p in seq.parameters():
        p.data.add_(0.05 * torch.randn_like(p))
    x0 = torch.tanh(torch.randn(B, D, device=cuda))
    go = torch.randn(B, E, device=cuda)
    res = []
    for mode in ("f64", "aten", "fused"):
        net = seq.double() if mode == "f64" else seq.float()
        dt = torch.float64 if mode == "f64" else torch.float32
        for p in net.parameters():
            p.grad = None
        x = x0.to(dt).clone().requires_grad_(True)
        out = head_ops.mlp3(x, net) if mode == "fused" else net(x)
        (out * go.to(dt)).sum().backward()
        res.append([t.detach().double().clone() for t in [out, x.grad] + [p.grad for p in net.parameters()]])
    for i, (truth, aten, fused) in enumerate(zip(*res)):
        scale = max(1e-3, float(truth.abs().max()))
        e_aten, e_fused = float((aten - truth).abs().max()), float((fused - truth).abs().max())
        assert e_fused <= max(3.0 * e_aten, 3e-6 * scale), (i, e_fused, e_aten)
    # no weight gradients wanted: only dx comes back, identical
    x = x0.clone().requires_grad_(True)
    frozen = head_ops._Mlp3.apply(x, *[p.detach() for p in seq.float().parameters()])
    (gx,) = torch.autograd.grad((frozen * go).sum(), [x])
    assert torch.equal(gx.double(), res[2][1])


def test_mask_colsum_and_mlp_out_kernels(cuda):
    """The C-ABI pieces on their own: ReLU gate + column sums (ragged sizes, with and without the extra matrix), and
    the twin output layer forward / backward against the ATen expressions."""
    g = torch.Generator(device=cuda).manual_seed(9)
    for B, N, E in ((1, 1, 0), (33, 70, 3), (512, 2048, 12), (100, 31, 40)):
        dz = torch.randn(B, N, device=cuda, generator=g)
        a = torch.relu(torch.randn(B, N, device=cuda, generator=g))
        ex = torch.randn(B, E, device=cuda, generator=g) if E else None
        want = torch.where(a > 0, dz, torch.zeros_like(dz))
        got = dz.clone()
        db, dex = head_ops._mask_colsum(got, a, True, ex)
        assert torch.equal(got, want)
        assert float((db.double() - want.double().sum(0)).abs().max()) <= 1e-5 * max(1.0, B ** 0.5)
        if E:
            assert float((dex.double() - ex.double().sum(0)).abs().max()) <= 1e-5 * max(1.0, B ** 0.5)
        got2 = dz.clone()
        assert head_ops._mask_colsum(got2, a, False) == (None, None) and torch.equal(got2, want)
    from augcl_amd import _lib
    T, B, H = 2, 77, 200
    z = torch.randn(T, B, H, device=cuda, generator=g)
    b2, w3, b3 = torch.randn(T, H, device=cuda, generator=g), torch.randn(T, H, device=cuda, generator=g), torch.randn(T, 1, device=cuda, generator=g)
    a2_want = torch.relu(z + b2[:, None])
    q_want = (a2_want.double() * w3.double()[:, None]).sum(-1) + b3.double()
    q = torch.empty(T, B, device=cuda)
    zz = z.clone()
    st = _lib.current_stream(cuda)
    _lib.call("augcl_mlp_out_forward", zz.data_ptr(), b2.data_ptr(), H, w3.data_ptr(), H, b3.data_ptr(), 1, q.data_ptr(), T, B, H, st)
    assert torch.equal(zz, a2_want) and float((q.double() - q_want).abs().max()) <= 1e-5 * float(q_want.abs().max())
    g0, g1 = torch.randn(B, device=cuda, generator=g), torch.randn(B, device=cuda, generator=g)
    gg = torch.stack([g0, g1]).double()
    dz2 = torch.empty_like(z)
    dw3, db3, db2 = torch.empty(T, H, device=cuda), torch.empty(T, device=cuda), torch.empty(T, H, device=cuda)
    _lib.call("augcl_mlp_out_backward", g0.data_ptr(), g1.data_ptr(), zz.data_ptr(), w3.data_ptr(), H, dz2.data_ptr(),
              dw3.data_ptr(), db3.data_ptr(), db2.data_ptr(), T, B, H, st)
    want = torch.where(zz > 0, (gg[:, :, None] * w3.double()[:, None]).float(), torch.zeros_like(z))
    assert torch.equal(dz2, want)
    for got_, want_ in ((dw3, (gg[:, :, None] * zz.double()).sum(1)), (db3, gg.sum(1)), (db2, want.double().sum(1))):
        assert float((got_.double() - want_).abs().max()) <= 1e-5 * max(1.0, float(want_.abs().max()))


def test_actor_loss_and_alpha_step_match_aten(cuda):
    """sac.py:119-146: the actor loss as one launch each way, and the temperature update (loss, gradient, Adam on the
    float64 scalar) as ONE launch, against the literal ATen chain + torch.optim.Adam over several steps."""
    from augcl_amd import flat
    g = torch.Generator(device=cuda).manual_seed(2)
    B = 256
    lp0 = torch.randn(B, 1, device=cuda, generator=g) * 5
    qa0 = torch.randn(B, 1, device=cuda, generator=g) * 20
    qb0 = qa0 + torch.randn(B, 1, device=cuda, generator=g)
    qb0[:5] = qa0[:5]  # ties: the gradient is split like ATen's minimum backward
    log_alpha = torch.tensor(np.log(0.1), device=cuda, requires_grad=True)
    res = []
    for fused in (False, True):
        lp, qa, qb = (t.clone().requires_grad_(True) for t in (lp0, qa0, qb0))
        loss = head_ops.actor_loss(lp, qa, qb, log_alpha) if fused else (log_alpha.exp().detach() * lp - torch.min(qa, qb)).mean()
        (loss * 1.3).backward()
        res.append((loss.detach(), lp.grad, qa.grad, qb.grad))
    for a, b in zip(*res):
        assert float((a - b).abs().max()) <= 1e-6 * max(1.0, float(a.abs().max()))
    assert log_alpha.grad is None
    # temperature: 6 steps, fresh log_pi each
    ref = torch.tensor(np.log(0.1), device=cuda, requires_grad=True)
    opt = torch.optim.Adam([ref], lr=1e-4, betas=(0.5, 0.999))
    mine = torch.tensor(np.log(0.1), device=cuda, requires_grad=True)
    fopt = flat.ScalarAdam(mine, 1e-4, (0.5, 0.999))
    for k in range(6):
        lp = torch.randn(B, 1, device=cuda, generator=g) * 3 - 2
        opt.zero_grad()
        want_loss = (ref.exp() * (-lp - (-6.0)).detach()).mean()
        want_alpha = float(ref.exp())
        want_loss.backward()
        opt.step()
        loss, alpha = fopt.fused_step(lp, -6.0)
        assert float(loss) == pytest.approx(float(want_loss), rel=2e-6, abs=1e-7) and float(alpha) == pytest.approx(want_alpha, rel=1e-6)
        assert float(mine) == pytest.approx(float(ref), rel=1e-9), k
    sd = fopt.state_dict()
    opt2 = torch.optim.Adam([torch.nn.Parameter(mine.detach().cpu().clone())], lr=1e-4, betas=(0.5, 0.999))
    opt2.load_state_dict(sd)  # torch's own Adam accepts the layout
    assert float(sd["state"][0]["step"]) == 6.0
    # td_target reads log_alpha directly
    tq1, tq2 = qa0, qb0
    rew, nd = torch.rand(B, 1, device=cuda, generator=g), torch.ones(B, 1, device=cuda)
    a = head_ops.td_target(tq1, tq2, lp0, rew, nd, mine.exp(), 0.99)
    b = head_ops.td_target(tq1, tq2, lp0, rew, nd, None, 0.99, log_alpha=mine)
    assert torch.equal(a, b)


@pytest.mark.parametrize("M", [1, 2, 33, 256, 512, 1024])
def test_projection_fused_gemm_ln_tanh(cuda, M):
    """RLProjection (modules.py:101-111) through the split-K MFMA GEMM + LN/tanh kernels: values and all five gradients
    against the float64 chain, as close as the ATen float32 chain; bitwise reproducible run to run."""
    torch.manual_seed(M)
    K, N = 14112, 100
    proj = m.RLProjection((K,), N).to(cuda)
    proj.projection[0].bias.data.normal_(0, 0.1)
    proj.projection[1].weight.data.uniform_(0.5, 1.5)
    proj.projection[1].bias.data.normal_(0, 0.1)
    lin, ln = proj.projection[0], proj.projection[1]
    g = torch.Generator(device=cuda).manual_seed(7)
    x0 = torch.relu(torch.randn(M, K, device=cuda, generator=g)) * 0.2  # CNN features: non-negative, sparse-ish
    gz = torch.randn(M, N, device=cuda, generator=g)
    assert head_ops.proj_ln_tanh(x0, lin, ln) is not None  # this shape is covered
    res = []
    for mode in ("f64", "aten", "fused"):
        dt = torch.float64 if mode == "f64" else torch.float32
        x = x0.to(dt).clone().requires_grad_(True)
        ps = [p.detach().to(dt).clone().requires_grad_(True) for p in (lin.weight, lin.bias, ln.weight, ln.bias)]
        if mode == "fused":
            for p in proj.parameters():
                p.grad = None
            z = proj(x)
        else:
            z = torch.tanh(F.layer_norm(F.linear(x, ps[0], ps[1]), (N,), ps[2], ps[3], 1e-5))
        (z * gz.to(dt)).sum().backward()
        grads = [p.grad for p in ((lin.weight, lin.bias, ln.weight, ln.bias) if mode == "fused" else ps)]
        res.append([t.detach().double() for t in [z, x.grad] + grads])
    for i, (truth, aten, fused) in enumerate(zip(*res)):
        scale = max(1e-6, float(truth.abs().max()))
        e_aten, e_fused = float((aten - truth).abs().max()), float((fused - truth).abs().max())
        assert e_fused <= max(3.0 * e_aten, 3e-6 * scale), (i, e_fused, e_aten, scale)
    with torch.no_grad():
        assert torch.equal(proj(x0), proj(x0))  # fixed-order split-K: the same bits every time


@pytest.mark.parametrize("M,N,K", [(1, 4, 1), (37, 52, 300), (256, 100, 14112), (100, 128, 1000), (65, 100, 129)])
def test_proj_dx_kernel(cuda, M, N, K):
    """augcl_proj_dx_f32: dx = dy w for ragged row / column counts (partial 32-row tiles, partial 128-column blocks,
    inner dimensions below the register tile), against the float64 product."""
    from augcl_amd import _lib
    g = torch.Generator(device=cuda).manual_seed(M + K)
    dy, w = torch.randn(M, N, device=cuda, generator=g), torch.randn(N, K, device=cuda, generator=g)
    dx = torch.full((M, K), float("nan"), device=cuda)
    _lib.call("augcl_proj_dx_f32", dy.data_ptr(), w.data_ptr(), dx.data_ptr(), M, N, K, _lib.current_stream(cuda))
    want = dy.double() @ w.double()
    assert float((dx.double() - want).abs().max()) <= 2e-6 * N ** 0.5 * max(1.0, float(want.abs().max()))


@pytest.mark.parametrize("M,R,A", [(256, 256, 6), (512, 256, 6), (66, 33, 21), (3, 1, 1)])
def test_projection_builds_the_critic_input_row(cuda, M, R, A):
    """proj_ln_tanh(..., tail=action): the launch that finishes the projection also writes [features | action] (action rows
    repeating over the batch, SVEA's 2B-row critic pass) -- equal, values and every gradient, to projection + repeat +
    torch.cat; and the critic takes that path by itself."""
    torch.manual_seed(M + A)
    K, N = 14112, 100
    proj = m.RLProjection((K,), N).to(cuda)
    lin, ln = proj.projection[0], proj.projection[1]
    g = torch.Generator(device=cuda).manual_seed(3)
    x0 = torch.relu(torch.randn(M, K, device=cuda, generator=g)) * 0.2
    t0 = torch.rand(R, A, device=cuda, generator=g) * 2 - 1
    go = torch.randn(M, N + A, device=cuda, generator=g)
    res = []
    for fused in (False, True):
        for p_ in proj.parameters():
            p_.grad = None
        x, t = x0.clone().requires_grad_(True), t0.clone().requires_grad_(True)
        if fused:
            z, xcat = head_ops.proj_ln_tanh(x, lin, ln, tail=t)
        else:
            z = head_ops.proj_ln_tanh(x, lin, ln)
            xcat = torch.cat([z, t.repeat(M // R, 1)], dim=1)
        (xcat * go).sum().backward()
        res.append([xcat.detach(), z.detach(), x.grad, t.grad] + [p_.grad.clone() for p_ in proj.parameters()])
    for i, (a, b) in enumerate(zip(*res)):
        assert torch.equal(a, b) if i != 3 else float((a - b).abs().max()) <= 1e-6 * max(1.0, float(a.abs().max())), i


def test_critic_uses_the_concatenating_projection(cuda):
    """Critic.forward: fused [features | action] rows (actions repeating for a 2B batch) == encoder, repeat, concat, trunks."""
    torch.manual_seed(11)
    B, A = 64, 6

    class Feat(torch.nn.Module):
        def forward(self, x):
            return x

    enc = m.Encoder(Feat(), Feat(), m.RLProjection((14112,), 100)).to(cuda)
    critic = m.Critic(enc, (A,), 256).to(cuda)
    x0 = torch.relu(torch.randn(2 * B, 14112, device=cuda)) * 0.2
    a0 = torch.rand(B, A, device=cuda) * 2 - 1
    out = []
    for fused in (True, False):
        for p_ in critic.parameters():
            p_.grad = None
        a = a0.clone().requires_grad_(True)
        if fused:
            q1, q2 = critic(x0, a)
        else:
            q1, q2 = critic._twin(critic.encoder(x0), a)
        (q1.sum() + 0.5 * q2.sum()).backward()
        out.append([q1.detach(), q2.detach(), a.grad] + [p_.grad.clone() for p_ in critic.parameters()])
    for i, (u, v) in enumerate(zip(*out)):
        assert float((u - v).abs().max()) <= 2e-6 * max(1e-3, float(v.abs().max())), i

