"""GPU parity of update(): augcl_amd agents (HIP augmentations + HIP encoder + fused Adam/EMA)
vs the golden vectors minted from the reference's own agents and vs the CPU oracle, on the same
seeded batch with the same injected noise and augmentation parameters."""
import os

import numpy as np
import pytest
import torch

import test_oracle_golden_sac as T
from augcl_amd import augmentations as aug
from augcl_amd import utils as au
from augcl_amd.algorithms import make_agent
from oracle import sac_oracle as so

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RTOL0 = 1e-5     # first update from identical state (north_star)
RTOL_TRAJ = 1e-4  # later steps: chaotic amplification of f32 rounding (see test_oracle_golden_sac.py)


class RecLogger:
    def __init__(self):
        self.rows = {}

    def log(self, key, value, step, n=1):
        self.rows[(step, key)] = float(value.item() if torch.is_tensor(value) else value)


class FakeBuffer:
    def __init__(self, batch):
        self.batch = batch

    def sample(self, n=None, idxs=None):
        return tuple(t.clone() for t in self.batch)


def build_agent(algo, data_aug, A, actor_state, critic_state, cuda, hp=None):
    args = au.dotdict(algorithm=algo, data_aug=data_aug, aug_params=None, **(hp or so.DEFAULTS))

    def load(agent):
        cs = {k: v.to(cuda) for k, v in critic_state.items()}
        as_ = {k: (cs[k] if so.is_cnn(k) else v.to(cuda)) for k, v in actor_state.items()}
        agent.critic.load_state_dict(cs)
        agent.actor.load_state_dict(as_)
        agent.critic_target.load_state_dict(cs)
        return agent

    if algo == "augcl":
        prev = load(make_agent(T.OBS, (A,), au.dotdict(dict(args, algorithm="non_naive_rad"))))
        agent = make_agent(T.OBS, (A,), args)
        agent.load_pretrained_agent(prev)
        return agent
    return load(make_agent(T.OBS, (A,), args))


@pytest.mark.parametrize("name,algo,data_aug,A,ci", T.CASES)
def test_update_matches_reference_golden(cuda, monkeypatch, name, algo, data_aug, A, ci):
    G = np.load(os.path.join(ROOT, "tests", "golden", "sac_golden.npz"))
    actor_state, critic_state, batch, noises, ov = T.make_case(name, algo, data_aug, A, ci)
    agent = build_agent(algo, data_aug, A, actor_state, critic_state, cuda, hp=T.case_hp(name))
    agent.fuse_front_end = False  # explicit parameters go through the un-fused ops (fused == chain is tested elsewhere)
    dbatch = tuple(t.to(cuda) for t in batch)
    buf, L = FakeBuffer(dbatch), RecLogger()
    queue = [n.to(cuda) for n in noises]
    monkeypatch.setattr(torch, "randn_like", lambda t, *a, **k: queue.pop(0).clone())
    keys = [str(k) for k in G[f"{name}/log_keys"]]
    want = G[f"{name}/logs"]
    state = {"s": 0, "shift_calls": 0}
    real_shift, real_conv, real_overlay = aug.random_shift, aug.random_conv, aug.random_overlay

    def shift_explicit(x, pad=4, **kw):
        s = state["s"]
        if f"{name}/shift_draws" in G.files:  # DrQ K, M > 1: K draws for obs, then M for next_obs, in call order
            oy, ox = G[f"{name}/shift_draws"][s][state["shift_calls"]]
            state["shift_calls"] += 1
            return real_shift(x, pad, oy=oy, ox=ox)
        second = state["shift_calls"] == 1 and (algo == "drq" or (algo == "rad" and data_aug == "shift"))
        state["shift_calls"] += 1
        oy = G[f"{name}/oy2" if second else f"{name}/oy"][s]
        ox = G[f"{name}/ox2" if second else f"{name}/ox"][s]
        return real_shift(x, pad, oy=oy, ox=ox)

    def conv_explicit(x, **kw):
        return real_conv(x, weights=G[f"{name}/conv_w"][state["s"]])

    def overlay_explicit(x, **kw):
        return real_overlay(x, imgs=torch.from_numpy(ov).to(cuda))

    import augcl_amd.algorithms.augcl as m_augcl
    import augcl_amd.algorithms.drq as m_drq
    import augcl_amd.algorithms.svea as m_svea
    for mod in (m_svea, m_drq, m_augcl):
        monkeypatch.setattr(mod.augmentations, "random_shift", shift_explicit)
    if hasattr(agent, "aug_funcs"):
        for k, fn in (("shift", shift_explicit), ("conv", conv_explicit), ("overlay", overlay_explicit)):
            if k in agent.aug_funcs:
                agent.aug_funcs[k]["func"] = fn

    for s in range(3):
        state["s"], state["shift_calls"] = s, 0
        step = 2 + s
        agent.update(buf, L, step)
        tol = RTOL0 if s == 0 else RTOL_TRAJ
        for j, k in enumerate(keys):
            if np.isnan(want[s, j]):
                continue
            got = L.rows[(step, k)]
            atol = tol * (T.ENT_CONST(A) + abs(want[s, j])) if k.endswith("mean_entropy") else tol
            if "variance" in k:  # drq.py:36-61 (population variance over the K / M copies; ~0 on these nets for K)
                atol = max(atol * abs(want[s, j]), T.variance_atol(want[s, j]))
            assert got == pytest.approx(want[s, j], rel=tol, abs=atol), (s, k)
        with torch.no_grad():
            q1, q2 = agent.critic(dbatch[0], dbatch[1])
        got = np.stack([q1.cpu().numpy().ravel(), q2.cpu().numpy().ravel()])
        np.testing.assert_allclose(got, G[f"{name}/q_after"][s], rtol=tol, atol=tol)
    assert not queue or len(queue) == len(noises) - int(G[f"{name}/noise_used"].sum())
    cn = list(so.critic_shapes(T.OBS, A, so.DEFAULTS).keys())
    for ck, net in (("critic_ck", agent.critic), ("target_ck", agent.critic_target)):
        got = so.checksum({k: v.detach().cpu() for k, v in net.named_parameters()}, cn)
        np.testing.assert_allclose(got[:, 1], G[f"{name}/{ck}"][:, 1], rtol=2e-5)
    assert float(agent.log_alpha) == pytest.approx(float(G[f"{name}/log_alpha"][0]), rel=1e-7)


def test_networks_match_reference_golden(cuda, monkeypatch):
    """Module-level goldens minted from the reference's own Critic / Actor (make_golden_sac.py, `nets/*`): Q1, Q2, the
    critic's parameter gradients, and the actor's (mu, pi, log_pi, log_std), against the HIP agent's modules -- HIP
    encoder forward AND backward, fused LN+tanh, fused Gaussian head.  (`nets/dx`, the gradient with respect to the
    image, has no counterpart: nothing on the update path differentiates with respect to pixels.)"""
    import inputs as gi
    G = np.load(os.path.join(ROOT, "tests", "golden", "sac_golden.npz"))
    A, hp = 6, dict(so.DEFAULTS)
    cs = so.synth_state(31, so.critic_shapes(T.OBS, A, hp))
    as_ = so.synth_state(32, so.actor_shapes(T.OBS, A, hp))
    agent = build_agent("sac", "identity", A, as_, cs, cuda)
    ob, ac, _, _, _ = gi.transition_batch(33, 3, A)
    x = torch.from_numpy(ob.astype(np.float32)).to(cuda)
    a = torch.from_numpy(ac).to(cuda)
    q1, q2 = agent.critic(x, a)
    (q1.sum() + 2.0 * q2.sum()).backward()
    np.testing.assert_allclose(q1.detach().cpu().numpy(), G["nets/q1"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(q2.detach().cpu().numpy(), G["nets/q2"], rtol=1e-5, atol=1e-6)
    names = list(so.critic_shapes(T.OBS, A, hp).keys())
    got = so.checksum({n: p.grad.cpu() for n, p in agent.critic.named_parameters()}, names)
    want = G["nets/critic_grad_ck"]
    np.testing.assert_allclose(got[:, 1], want[:, 1], rtol=1e-5)                # abs-sum of every gradient tensor
    shapes = so.critic_shapes(T.OBS, A, hp)
    for n, g_row, w_row in zip(names, got, want):                               # first values, on the tensor's own scale
        scale = max(np.abs(w_row[2:]).max(), w_row[1] / int(np.prod(shapes[n])))
        np.testing.assert_allclose(g_row[2:], w_row[2:], rtol=1e-4, atol=1e-4 * scale, err_msg=n)
    noise = torch.from_numpy(np.random.RandomState(34).standard_normal((3, A)).astype(np.float32)).to(cuda)
    monkeypatch.setattr(torch, "randn_like", lambda t, *a_, **k: noise.clone())
    with torch.no_grad():
        mu, pi, log_pi, log_std = agent.actor(x)
    for k, v in (("mu", mu), ("pi", pi), ("log_pi", log_pi), ("log_std", log_std)):
        np.testing.assert_allclose(v.cpu().numpy(), G[f"nets/{k}"], rtol=1e-5, atol=1e-6, err_msg=k)


def test_adam_and_ema_kernels_match_torch(cuda):
    from augcl_amd import flat
    rs = np.random.RandomState(0)
    n = 100003  # not a multiple of 4: exercises the tail kernel
    p0 = torch.from_numpy(rs.standard_normal(n).astype(np.float32))
    grads = [torch.from_numpy((rs.standard_normal(n) * 10.0 ** rs.uniform(-6, 1, n)).astype(np.float32)) for _ in range(4)]
    ref = p0.clone().requires_grad_(True)
    opt = torch.optim.Adam([ref], lr=1e-3, betas=(0.9, 0.999), foreach=False)
    par = torch.nn.Parameter(p0.clone().to(cuda))
    grp = flat.FlatGroup([("p", par)])
    fopt = flat.FlatAdam(grp, 1e-3, (0.9, 0.999))
    for g in grads:
        ref.grad = g.clone()
        opt.step()
        fopt.zero_grad()
        par.grad = g.to(cuda)
        fopt.step()
    torch.cuda.synchronize()
    np.testing.assert_allclose(par.detach().cpu().numpy(), ref.detach().numpy(), rtol=2e-6, atol=2e-8)
    # EMA: bit-exact with the reference expression
    t0 = torch.from_numpy(rs.standard_normal(n).astype(np.float32))
    want = 0.05 * p0 + (1 - 0.05) * t0
    tg = t0.clone().to(cuda)
    flat.ema_(tg, p0.to(cuda), 0, n, 0.05)
    assert torch.equal(tg.cpu(), want)
    tg = t0.clone().to(cuda)
    flat.ema_(tg, p0.to(cuda), 1, n, 0.01)  # unaligned segment start
    assert torch.equal(tg.cpu()[1:], (0.01 * p0 + (1 - 0.01) * t0)[1:]) and tg.cpu()[0] == t0[0]


def test_adam_device_side_step_and_two_segment_ema(cuda):
    """The graph-capturable Adam (step count on the device, bias corrections derived inside the ONE launch, the count
    stored by the last workgroup to retire) walks exactly like the host-step variant; augcl_ema2_f32 = two augcl_ema_f32."""
    from augcl_amd import flat
    rs = np.random.RandomState(3)
    n = 300000
    p0 = torch.from_numpy(rs.standard_normal(n).astype(np.float32))
    grads = [torch.from_numpy((rs.standard_normal(n) * 10.0 ** rs.uniform(-5, 1, n)).astype(np.float32)).to(cuda) for _ in range(6)]
    opts = []
    for dev_step in (False, True):
        par = torch.nn.Parameter(p0.clone().to(cuda))
        fopt = flat.FlatAdam(flat.FlatGroup([("p", par)]), 3e-4, (0.9, 0.999))
        if dev_step:
            fopt.make_capturable()
        opts.append((par, fopt))
    for i, g in enumerate(grads):
        for par, fopt in opts:
            fopt.zero_grad()
            par.grad = g.clone()
            fopt.step()
        assert opts[1][1].step_count == i + 1
    assert torch.equal(opts[0][0].detach(), opts[1][0].detach())
    assert torch.equal(opts[0][1].exp_avg_sq, opts[1][1].exp_avg_sq)
    # the gather launch on its own: several tensors, sources at odd offsets (views), big and tiny, one launch
    shapes = [(7, 5), (33,), (1024, 56), (3,)]
    pars = [torch.nn.Parameter(torch.zeros(sh, device=cuda)) for sh in shapes]
    grp = flat.FlatGroup([(f"p{i}", p) for i, p in enumerate(pars)])
    big = torch.randn(70000, device=cuda)
    srcs = [big[o:o + int(np.prod(sh))].view(sh) for o, sh in zip((1, 40, 101, 4), shapes)]
    assert grp.scatter_into_bucket(srcs) == [] and not grp.prepped
    for name, src in zip(grp.names, srcs):
        a, b = grp.slices[name]
        assert torch.equal(grp.grad[a:b], src.reshape(-1)), name
    assert grp.scatter_into_bucket([srcs[0], None, srcs[2].t().contiguous().t(), srcs[3]]) == ["p1"]  # a strided gradient
    a, b = grp.slices["p2"]
    assert torch.equal(grp.grad[a:b], srcs[2].reshape(-1)) and float(grp.grad[slice(*grp.slices["p1"])].abs().max()) == 0.0
    # a state that starts at a later step (checkpoint resume)
    opts[1][1].step_count = 41
    assert opts[1][1].step_count == 41
    t0 = torch.from_numpy(rs.standard_normal(n).astype(np.float32)).to(cuda)
    src = p0.to(cuda)
    one, two = t0.clone(), t0.clone()
    flat.ema_(one, src, 5, 1000, 0.05)
    flat.ema_(one, src, 1000, n - 3, 0.01)
    flat.ema2_(two, src, (5, 1000, 0.05), (1000, n - 3, 0.01))
    assert torch.equal(one, two) and not torch.equal(one, t0)
    flat.ema2_(two, src, (0, 0, 0.5), (7, 7, 0.5))  # two empty segments: nothing happens
    assert torch.equal(one, two)


def test_flat_adam_skips_missing_gradients_like_torch(cuda):
    """torch.optim.Adam leaves a parameter without gradient untouched (no moment decay, no step, its own step count);
    FlatAdam steps the runs of tensors that do have one and keeps per-tensor counts from then on."""
    from augcl_amd import flat
    rs = np.random.RandomState(1)
    shapes = [(7, 5), (33,), (4, 4, 3), (10,)]
    p0 = [torch.from_numpy(rs.standard_normal(s).astype(np.float32)) for s in shapes]
    ref = [t.clone().requires_grad_(True) for t in p0]
    opt = torch.optim.Adam(ref, lr=1e-2, betas=(0.9, 0.999), foreach=False)
    par = [torch.nn.Parameter(t.clone().to(cuda)) for t in p0]
    grp = flat.FlatGroup([(f"p{i}", p) for i, p in enumerate(par)])
    fopt = flat.FlatAdam(grp, 1e-2, (0.9, 0.999))
    present = [(1, 1, 1, 1), (1, 0, 1, 1), (0, 1, 1, 0), (1, 1, 1, 1), (1, 0, 0, 1)]
    for mask in present:
        opt.zero_grad(set_to_none=True)
        fopt.zero_grad()
        for t, p, m, s in zip(ref, par, mask, shapes):
            if m:
                g = torch.from_numpy(rs.standard_normal(s).astype(np.float32))
                t.grad, p.grad = g.clone(), g.to(cuda)
        opt.step()
        fopt.step()
    torch.cuda.synchronize()
    for i, (t, p) in enumerate(zip(ref, par)):
        np.testing.assert_allclose(p.detach().cpu().numpy(), t.detach().numpy(), rtol=2e-6, atol=2e-8, err_msg=f"p{i}")
        assert fopt.tensor_step(f"p{i}") == int(opt.state[t]["step"])
    # the same through backward_step(loss): gradients only for the tensors the loss depends on
    loss_r = (ref[0].sum() * 2.0 + (ref[2] ** 2).sum())
    opt.zero_grad(set_to_none=True)
    loss_r.backward()
    opt.step()
    fopt.backward_step(par[0].sum() * 2.0 + (par[2] ** 2).sum())
    for i, (t, p) in enumerate(zip(ref, par)):
        np.testing.assert_allclose(p.detach().cpu().numpy(), t.detach().numpy(), rtol=2e-6, atol=2e-8, err_msg=f"p{i} (backward_step)")


def test_agent_surface_and_actions(cuda):
    args = au.dotdict(algorithm="svea", data_aug="conv", aug_params=None, **so.DEFAULTS)
    agent = make_agent(T.OBS, (6,), args)
    assert agent.actor.encoder.shared_cnn is agent.critic.encoder.shared_cnn  # shared CNN object (sac.py:17-32)
    assert agent.critic_target.encoder.shared_cnn is not agent.critic.encoder.shared_cnn
    assert agent.log_alpha.dtype == torch.float64 and agent.training
    n_c = sum(p.numel() for p in agent.critic.parameters())
    n_a = sum(p.numel() for p in agent.actor.parameters())
    assert (n_c, n_a) == (3826990, 2671928)
    obs = np.random.RandomState(0).randint(0, 256, (9, 84, 84)).astype(np.uint8)
    a1 = agent.select_action(obs)
    a2 = agent.sample_action(au.LazyFrames([obs[:3], obs[3:6], obs[6:]]))
    assert a1.shape == (6,) and a2.shape == (6,) and np.all(np.abs(a1) <= 1) and np.all(np.abs(a2) <= 1)
    with au.eval_mode(agent):
        assert not agent.training
    assert agent.training
    # a full default-path update (fused u8 front end, in-kernel Philox parameters) runs and changes the nets
    B = 16
    g = torch.Generator(device=cuda).manual_seed(0)
    batch = (torch.randint(0, 256, (B, 9, 84, 84), device=cuda, generator=g).float(), torch.rand(B, 6, device=cuda) * 2 - 1,
             torch.rand(B, 1, device=cuda), torch.randint(0, 256, (B, 9, 84, 84), device=cuda, generator=g).float(),
             torch.ones(B, 1, device=cuda))
    before = agent.critic_group.param.clone()
    L = RecLogger()
    for step in (0, 1):
        agent.update(FakeBuffer(batch), L, step)
    assert not torch.equal(before, agent.critic_group.param)
    assert all(np.isfinite(v) for v in L.rows.values())
    assert {k for _, k in L.rows} >= {"train_critic/loss", "train_critic/shift_loss", "train_critic/aug_loss",
                                      "train_actor/loss", "train_alpha/value"}


def test_data_parallel_rank_draws_its_shard_of_the_global_augmentation(cuda):
    """Data parallel: in-kernel augmentation parameters are keyed by the GLOBAL sample index, and update() tells the
    kernels where this rank's shard starts (rank * local batch) -- rank 1 of 2 reproduces the second half of what a
    single process draws for the whole batch, so the global batch keeps its full augmentation diversity."""
    class Rank1Of2:
        rank, world = 1, 2

        def allreduce_sum_(self, flat_grad):
            return 1.0

        def mean_scalar(self, v):
            return v

    args = au.dotdict(algorithm="svea", data_aug="conv", aug_params=None, **so.DEFAULTS)
    agent = make_agent(T.OBS, (6,), args)
    B = 8
    g = torch.Generator(device=cuda).manual_seed(9)
    full = (torch.randint(0, 256, (B, 9, 84, 84), device=cuda, generator=g).float(), torch.rand(B, 6, device=cuda) * 2 - 1,
            torch.rand(B, 1, device=cuda), torch.randint(0, 256, (B, 9, 84, 84), device=cuda, generator=g).float(),
            torch.ones(B, 1, device=cuda))
    seen = []
    agent._learn = lambda t, *a, **k: seen.append(t[0].clone())
    try:
        aug.manual_seed(5)
        agent.update(FakeBuffer(full), None, 0)                       # one process, whole batch
        agent._reducer = Rank1Of2()
        aug.manual_seed(5)
        agent.update(FakeBuffer(tuple(t[B // 2:] for t in full)), None, 0)  # rank 1 of 2, its shard
    finally:
        aug.set_sample_base(0)
    whole, shard = seen
    assert whole.shape[0] == 2 * B and shard.shape[0] == B  # [shifted; shifted + conv] stacks
    assert torch.equal(shard[:B // 2], whole[B // 2:B]) and torch.equal(shard[B // 2:], whole[B + B // 2:])
    assert not torch.equal(whole[:B // 2], whole[B // 2:B])


def test_action_graphs_match_eager_and_follow_training(cuda, tmp_path):
    """select_action / sample_action (src/algorithms/sac.py:73-92) replayed from a captured hipGraph: the same action as
    the eager path bit for bit, still so after an update() has moved the weights (the graph reads the live flat buffers),
    stochastic actions bounded and varying, and the agent still pickles."""
    args = au.dotdict(algorithm="svea", data_aug="conv", aug_params=None, **so.DEFAULTS)
    torch.manual_seed(5)
    agent = make_agent(T.OBS, (6,), args)
    rs = np.random.RandomState(3)
    frames = [rs.randint(0, 256, (9, 84, 84)).astype(np.uint8) for _ in range(4)]
    eager = [agent.select_action(f) for f in frames]
    agent.enable_action_graphs()
    for f, e in zip(frames, eager):
        assert np.array_equal(agent.select_action(f), e)
    assert np.array_equal(agent.select_action(au.LazyFrames([frames[0][:3], frames[0][3:6], frames[0][6:]])), eager[0])
    s1, s2 = agent.sample_action(frames[0]), agent.sample_action(frames[0])
    assert s1.shape == (6,) and np.all(np.abs(s1) <= 1) and not np.array_equal(s1, s2)
    B = 8
    g = torch.Generator(device=cuda).manual_seed(1)
    batch = (torch.randint(0, 256, (B, 9, 84, 84), device=cuda, generator=g).float(), torch.rand(B, 6, device=cuda) * 2 - 1,
             torch.rand(B, 1, device=cuda), torch.randint(0, 256, (B, 9, 84, 84), device=cuda, generator=g).float(),
             torch.ones(B, 1, device=cuda))
    # the weight-fragment workspace the acting graph baked in (one weight set) must be the very buffer the update's dual
    # forward (two weight sets) uses: a buffer replaced on growth would leave the graph writing into freed memory
    from augcl_amd import encoder_hip
    fws_ptr, fws_len = encoder_hip._fws[cuda].data_ptr(), encoder_hip._fws[cuda].numel()
    for step in range(2):
        agent.update(FakeBuffer(batch), None, step)
    assert encoder_hip._fws[cuda].data_ptr() == fws_ptr and encoder_hip._fws[cuda].numel() == fws_len
    assert encoder_hip._frag_ws(cuda, 32, 2).data_ptr() == fws_ptr  # the ABI's maximum fits without a new allocation
    after = agent.select_action(frames[1])
    assert not np.array_equal(after, eager[1])
    agent.set_precision("f32_x3")  # another arithmetic: the acting graphs are dropped with the update graphs
    assert agent._act["graphs"] == {}
    x3 = agent.select_action(frames[1])
    assert np.allclose(x3, after, atol=1e-5)
    agent.set_precision("f32")
    assert np.array_equal(agent.select_action(frames[1]), after)
    agent._act = None
    assert np.array_equal(agent.select_action(frames[1]), after)  # eager on the trained weights == the old graph
    agent.enable_action_graphs()
    agent.select_action(frames[1])
    torch.save(agent, tmp_path / "a.pt")
    loaded = torch.load(tmp_path / "a.pt", weights_only=False)  # our own file
    assert np.array_equal(loaded.select_action(frames[1]), after)


def test_agent_pickle_roundtrip_and_state_dict_names(cuda, tmp_path):
    """train.py:229 saves the whole agent with torch.save; eval.py reloads it.  Flat-buffer views must survive."""
    args = au.dotdict(algorithm="svea", data_aug="conv", aug_params=None, **so.DEFAULTS)
    agent = make_agent(T.OBS, (6,), args)
    want_keys = list(so.critic_shapes(T.OBS, 6, so.DEFAULTS).keys())
    assert list(agent.critic.state_dict().keys()) == want_keys  # the reference's state-dict names, in order
    assert list(agent.actor.state_dict().keys()) == list(so.actor_shapes(T.OBS, 6, so.DEFAULTS).keys())
    path = tmp_path / "agent.pt"
    torch.save(agent, path)
    loaded = torch.load(path, weights_only=False)  # our own file
    obs = np.random.RandomState(1).randint(0, 256, (9, 84, 84)).astype(np.uint8)
    assert np.array_equal(agent.select_action(obs), loaded.select_action(obs))
    # the reloaded agent still trains through the flat groups
    B = 8
    g = torch.Generator(device=cuda).manual_seed(1)
    batch = (torch.randint(0, 256, (B, 9, 84, 84), device=cuda, generator=g).float(), torch.rand(B, 6, device=cuda) * 2 - 1,
             torch.rand(B, 1, device=cuda), torch.randint(0, 256, (B, 9, 84, 84), device=cuda, generator=g).float(),
             torch.ones(B, 1, device=cuda))
    before = loaded.critic.Q1.trunk[0].weight.detach().clone()
    loaded.update(FakeBuffer(batch), None, 0)
    assert not torch.equal(before, loaded.critic.Q1.trunk[0].weight)
    assert loaded.critic.Q1.trunk[0].weight.data_ptr() >= loaded.critic_group.param.data_ptr()


def test_full_batch_update_runs_all_algorithms(cuda):
    """BASELINE configs' batch sizes through the default (RNG-internal, fused) paths: finite losses, nets move."""
    from augcl_amd import augmentations as aug
    aug.set_overlay_source(aug.SyntheticOverlaySource(seed=1))
    cases = (("svea", "conv", 6, 256), ("drq", "identity", 2, 512), ("non_naive_rad", "shift", 1, 128),
             ("augcl", "conv-overlay", 6, 1024))  # BASELINE configs 2, 3, 1, 4 (config 4's global batch on one GPU)
    for algo, data_aug, A, B in cases:
        args = au.dotdict(algorithm=algo, data_aug=data_aug, aug_params=None, **so.DEFAULTS)
        agent = make_agent(T.OBS, (A,), args)
        if algo == "augcl":  # the curriculum hand-off: weak critic = the pretrained agent's critic
            agent.load_pretrained_agent(make_agent(T.OBS, (A,), au.dotdict(dict(args, algorithm="non_naive_rad", data_aug="shift"))))
        g = torch.Generator(device=cuda).manual_seed(2)
        batch = (torch.randint(0, 256, (B, 9, 84, 84), device=cuda, generator=g).float(), torch.rand(B, A, device=cuda) * 2 - 1,
                 torch.rand(B, 1, device=cuda), torch.randint(0, 256, (B, 9, 84, 84), device=cuda, generator=g).float(),
                 torch.ones(B, 1, device=cuda))
        L = RecLogger()
        before = agent.critic_group.param.clone()
        for step in (0, 1):
            agent.update(FakeBuffer(batch), L, step)
        assert all(np.isfinite(v) for v in L.rows.values()), (algo, L.rows)
        assert not torch.equal(before, agent.critic_group.param)
    aug.set_overlay_source(None)


@pytest.mark.parametrize("algo,data_aug,A", [("svea", "conv", 6), ("drq", "identity", 2), ("augcl", "overlay", 3)])
def test_hip_graph_update_matches_eager(cuda, algo, data_aug, A):
    """update() replayed from captured hipGraphs (SAC.enable_hip_graphs: device-side Adam step counts, eager front
    end writing into the graph's static inputs) walks the same trajectory as the eager path: same seeds, same
    batches, 10 steps with logging on and off, every (actor step?, target step?) variant."""
    from augcl_amd import augmentations as aug
    B = 16
    g = torch.Generator(device=cuda).manual_seed(3)
    batches = [(torch.randint(0, 256, (B, 9, 84, 84), device=cuda, generator=g).float(),
                torch.rand(B, A, device=cuda, generator=g) * 2 - 1, torch.rand(B, 1, device=cuda, generator=g),
                torch.randint(0, 256, (B, 9, 84, 84), device=cuda, generator=g).float(), torch.ones(B, 1, device=cuda))
               for _ in range(3)]

    class Ring:
        def __init__(self):
            self.i = 0

        def sample(self, n=None, idxs=None):
            self.i += 1
            return tuple(t.clone() for t in batches[self.i % 3])

    results = []
    for graph in (False, True):
        args = au.dotdict(algorithm=algo, data_aug=data_aug, aug_params=None, **so.DEFAULTS)
        torch.manual_seed(11)
        aug.manual_seed(7)
        aug.set_overlay_source(aug.SyntheticOverlaySource(seed=5))
        try:
            agent = make_agent(T.OBS, (A,), args)
            if graph:
                agent.enable_hip_graphs(warmup=2)
            L, buf = RecLogger(), Ring()
            for step in range(10):
                agent.update(buf, L if step % 3 else None, step)
            torch.cuda.synchronize()
            snap = (agent.critic_group.param.clone(), agent.actor_group.param.clone(), agent.target_group.param.clone(),
                    agent.log_alpha.detach().clone(), dict(L.rows), agent.critic_optimizer.step_count)
            if graph:
                assert len(agent._graph["graphs"]) >= 3
                # another batch size: the graphs are recaptured instead of replayed on stale shapes
                half = tuple(t[:B // 2].clone() for t in batches[0])
                for step in (10, 11):
                    agent.update(FakeBuffer(half), None, step)
                torch.cuda.synchronize()
                assert agent._graph["shapes"][0] == B // 2 and agent.critic_optimizer.step_count == 12
        finally:
            aug.set_overlay_source(None)
        results.append(snap)
    e, gr = results
    assert e[5] == gr[5] == 10
    for a, b in zip(e[:4], gr[:4]):
        assert float((a.double() - b.double()).abs().max()) <= 1e-5 * max(1.0, float(a.abs().max()))
    assert e[4].keys() == gr[4].keys()
    for k in e[4]:
        assert abs(e[4][k] - gr[4][k]) <= 1e-4 * max(1.0, abs(e[4][k])), k


def test_checkpoint_state_roundtrip_and_reference_layout(cuda, tmp_path):
    """checkpoint.agent_state / load_agent_state: reference parameter names, torch.optim.Adam state layout indexed by
    position in module.parameters(), and a save -> load into a fresh agent that continues on the same trajectory."""
    from augcl_amd import checkpoint
    args = au.dotdict(algorithm="svea", data_aug="conv", aug_params=None, **so.DEFAULTS)
    B, A = 8, 6
    g = torch.Generator(device=cuda).manual_seed(4)
    batch = (torch.randint(0, 256, (B, 9, 84, 84), device=cuda, generator=g).float(), torch.rand(B, A, device=cuda) * 2 - 1,
             torch.rand(B, 1, device=cuda), torch.randint(0, 256, (B, 9, 84, 84), device=cuda, generator=g).float(),
             torch.ones(B, 1, device=cuda))
    from augcl_amd import augmentations as aug

    def run(agent, steps, first):
        for s in range(first, first + steps):
            torch.manual_seed(100 + s)
            aug.manual_seed(200 + s)
            agent.update(FakeBuffer(batch), None, s)

    torch.manual_seed(1)
    a1 = make_agent(T.OBS, (A,), args)
    run(a1, 3, 0)
    st = checkpoint.agent_state(a1)
    # reference naming + Adam layout
    assert "encoder.shared_cnn.layers.2.weight" in st["critic"] and "Q1.trunk.0.weight" in st["critic"] and "mlp.4.bias" in st["actor"]
    n_critic = len(list(a1.critic.parameters()))
    assert sorted(st["critic_optimizer"]["state"]) == list(range(n_critic))
    assert st["critic_optimizer"]["param_groups"][0]["params"] == list(range(n_critic))
    shared = {id(p) for p in a1.critic.parameters()}
    moved = [i for i, p in enumerate(a1.actor.parameters()) if id(p) not in shared]
    assert sorted(st["actor_optimizer"]["state"]) == moved  # the shared CNN has no state in the actor's optimiser
    assert float(st["critic_optimizer"]["state"][0]["step"]) == 3.0
    ref_opt = torch.optim.Adam([torch.nn.Parameter(p.detach().cpu().clone()) for p in a1.critic.parameters()], lr=1e-3)
    ref_opt.load_state_dict(st["critic_optimizer"])  # torch's own Adam accepts it

    path = tmp_path / "agent_state.pt"
    checkpoint.save(a1, path)
    torch.manual_seed(2)
    a2 = make_agent(T.OBS, (A,), args)
    checkpoint.load(a2, path)
    for n in ("critic_group", "actor_group", "target_group"):
        assert torch.equal(getattr(a1, n).param, getattr(a2, n).param), n
    assert a2.critic_optimizer.step_count == 3 and torch.equal(a1.critic_optimizer.exp_avg, a2.critic_optimizer.exp_avg)
    assert a2.actor.encoder.shared_cnn.layers[2].weight.data_ptr() == a2.critic.encoder.shared_cnn.layers[2].weight.data_ptr()
    run(a1, 2, 3)
    run(a2, 2, 3)
    for n in ("critic_group", "actor_group", "target_group"):
        assert torch.equal(getattr(a1, n).param, getattr(a2, n).param), n
    assert torch.equal(a1.log_alpha, a2.log_alpha)


# ------------------------------------------------------------------------------------------------ bf16 leg of config 2
def _svea_batches(cuda, B, A, n=3, seed=3):
    g = torch.Generator(device=cuda).manual_seed(seed)
    return [(torch.randint(0, 256, (B, 9, 84, 84), device=cuda, generator=g).float(),
             torch.rand(B, A, device=cuda, generator=g) * 2 - 1, torch.rand(B, 1, device=cuda, generator=g),
             torch.randint(0, 256, (B, 9, 84, 84), device=cuda, generator=g).float(), torch.ones(B, 1, device=cuda))
            for _ in range(n)]


def test_bf16_update_hip_graph_matches_eager(cuda):
    """BASELINE config 2's bf16 leg (walker_walk SVEA + random_conv, precision="bf16": bf16 NHWC activations, f32 heads,
    f32 weight gradients from bf16 operands, device-step Adam) at update() level: the trajectory replayed from captured
    hipGraphs -- what bench.py times -- equals the eager one over 10 steps (same kernels, same order: the bounds are the
    f32 path's)."""
    A, B = 6, 16
    batches = _svea_batches(cuda, B, A)

    class Ring:
        def __init__(self):
            self.i = 0

        def sample(self, n=None, idxs=None):
            self.i += 1
            return tuple(t.clone() for t in batches[self.i % 3])

    results = []
    for graph in (False, True):
        args = au.dotdict(algorithm="svea", data_aug="conv", aug_params=None, precision="bf16", **so.DEFAULTS)
        torch.manual_seed(11)
        aug.manual_seed(7)
        agent = make_agent(T.OBS, (A,), args)
        assert agent.critic.encoder.shared_cnn.precision == "bf16" and agent.critic_target.encoder.shared_cnn.precision == "bf16"
        if graph:
            agent.enable_hip_graphs(warmup=2)
        L, buf = RecLogger(), Ring()
        for step in range(10):
            agent.update(buf, L if step % 3 else None, step)
        torch.cuda.synchronize()
        if graph:
            assert len(agent._graph["graphs"]) >= 3
        results.append((agent.critic_group.param.clone(), agent.actor_group.param.clone(), agent.target_group.param.clone(),
                        agent.log_alpha.detach().clone(), dict(L.rows)))
    e, gr = results
    for a, b in zip(e[:4], gr[:4]):
        assert bool(torch.isfinite(a).all())
        assert float((a.double() - b.double()).abs().max()) <= 1e-5 * max(1.0, float(a.abs().max()))
    assert e[4].keys() == gr[4].keys() and len(e[4]) > 0
    for k in e[4]:
        assert abs(e[4][k] - gr[4][k]) <= 1e-4 * max(1.0, abs(e[4][k])), k
    # the process default is untouched: the precision lives on the agent's modules
    from augcl_amd import encoder_ops
    assert encoder_ops.PRECISION == "f32"


def test_bf16_update_tracks_f32_update(cuda, monkeypatch):
    """The bf16 agent against the f32 agent from the SAME state (gate-safe synthetic nets: no ReLU input near zero, so
    what differs is bf16 rounding, not gates), the same batch, injected noise and explicit augmentation parameters:
    every logged scalar of updates 1, 2 and 3.  Parameters: the two trajectories stay within 3 Adam steps of each other (Adam's first steps
    are lr * sign(g): an element whose gradient sits within bf16 noise of zero may go the other way)."""
    A, B = 6, 32
    hp = dict(so.DEFAULTS)
    as_ = so.synth_state(41, so.actor_shapes(T.OBS, A, hp))
    cs = so.synth_state(42, so.critic_shapes(T.OBS, A, hp))
    for k in cs:
        if so.is_cnn(k):
            as_[k] = cs[k]
    rs = np.random.RandomState(9)
    batch = tuple(torch.from_numpy(v).to(cuda) for v in (
        rs.randint(0, 256, (B, *T.OBS)).astype(np.float32), rs.uniform(-1, 1, (B, A)).astype(np.float32),
        rs.uniform(0, 4, (B, 1)).astype(np.float32), rs.randint(0, 256, (B, *T.OBS)).astype(np.float32), np.ones((B, 1), np.float32)))
    noises = [torch.from_numpy(rs.standard_normal((B, A)).astype(np.float32)).to(cuda) for _ in range(8)]
    draws = [(rs.randint(0, 9, B).astype(np.int32), rs.randint(0, 9, B).astype(np.int32)) for _ in range(3)]
    cw = [rs.standard_normal((B, 3, 3, 3, 3)).astype(np.float32) for _ in range(3)]
    real_shift, real_conv = aug.random_shift, aug.random_conv
    runs = {}
    for prec in ("f32", "bf16"):
        agent = build_agent("svea", "conv", A, as_, cs, cuda, hp=dict(hp, precision=prec))
        agent.fuse_front_end = False
        queue = [n.clone() for n in noises]
        state = {"s": 0}
        monkeypatch.setattr(torch, "randn_like", lambda t, *a, **k: queue.pop(0))
        monkeypatch.setattr(aug, "random_shift", lambda x, pad=4, **kw: real_shift(x, pad, oy=draws[state["s"]][0], ox=draws[state["s"]][1]))
        monkeypatch.setattr(aug, "random_conv", lambda x, **kw: real_conv(x, weights=cw[state["s"]]))
        L = RecLogger()
        for step in range(3):
            state["s"] = step
            agent.update(FakeBuffer(batch), L, step)
        torch.cuda.synchronize()
        runs[prec] = (dict(L.rows), agent.critic_group.param.clone(), agent.actor_group.param.clone())
        monkeypatch.undo()
    (l32, c32, a32), (l16, c16, a16) = runs["f32"], runs["bf16"]
    assert l32.keys() == l16.keys() and any(s == 2 for s, _ in l32)
    worst = {0: 0.0, 1: 0.0, 2: 0.0}
    for k in l32:
        assert np.isfinite(l16[k]), k
        rel = abs(l16[k] - l32[k]) / max(abs(l32[k]), 1e-2)
        worst[k[0]] = max(worst[k[0]], rel)
    print("bf16 vs f32 update: worst logged-scalar deviation per step", {s: "%.2e" % v for s, v in worst.items()})
    # measured (MI355X): 1.4e-2 on the first update (same state, same batch: bf16 rounding through 11 layers and the loss's
    # square), 4.7e-2 / 5.9e-2 one / two Adam steps later (steps of lr * sign(g) amplify every sign a bf16 gradient flips)
    assert worst[0] <= 3e-2, worst
    assert max(worst[1], worst[2]) <= 1.5e-1, worst
    lr = hp["critic_lr"]
    assert float((c32 - c16).abs().max()) <= 3 * 2 * lr * 1.01 and float((a32 - a16).abs().max()) <= 3 * 2 * hp["actor_lr"] * 1.01
    # ... and the critic moved the same way in both (bf16 gradients are not noise): direction of the 3-step displacement
    c0 = build_agent("svea", "conv", A, as_, cs, cuda, hp=hp).critic_group.param
    d32, d16 = (c32 - c0).double(), (c16 - c0).double()
    assert float(d32 @ d16 / (d32.norm() * d16.norm())) > 0.9


def test_head_layers_match_reference_golden(cuda, monkeypatch):
    """Agents WITH head layers (num_head_layers = 2: HeadCNN = [ReLU, Conv]* on the shared CNN's un-activated output, shared
    by actor and critic -- src/algorithms/modules.py:168-181, sac.py:20-33) against vectors minted from the reference
    (tests/golden/make_golden_head.py): the relu_first per-layer conv path forward and backward (its input gradient feeds the
    shared stack's backward), Q values, the critic's parameter gradients, the policy head, and one full update()."""
    import test_oracle_golden_sac as TO
    G = np.load(os.path.join(ROOT, "tests", "golden", "head_golden.npz"))
    mh, as_, cs, batch, nz = TO.head_case()
    hp, A = mh.HP, mh.A
    agent = build_agent("sac", "identity", A, as_, cs, cuda, hp=hp)
    obs, act = batch[0].to(cuda), batch[1].to(cuda)
    enc = agent.critic.encoder
    assert agent.actor.encoder.head_cnn is enc.head_cnn and len(list(enc.head_cnn.parameters())) == 4
    with torch.no_grad():
        feat = enc.head_cnn(enc.shared_cnn(obs))
    np.testing.assert_allclose(feat.cpu().numpy().reshape(G["head/features"].shape), G["head/features"], rtol=1e-5, atol=1e-6)
    q1, q2 = agent.critic(obs, act)
    (q1.sum() + 2.0 * q2.sum()).backward()
    np.testing.assert_allclose(q1.detach().cpu().numpy(), G["head/q1"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(q2.detach().cpu().numpy(), G["head/q2"], rtol=1e-5, atol=1e-6)
    names = list(so.critic_shapes(T.OBS, A, hp).keys())
    got = so.checksum({n: p.grad.cpu() for n, p in agent.critic.named_parameters()}, names)
    want = G["head/critic_grad_ck"]
    np.testing.assert_allclose(got[:, 1], want[:, 1], rtol=1e-5)
    shapes = so.critic_shapes(T.OBS, A, hp)
    for n, g_row, w_row in zip(names, got, want):
        scale = max(np.abs(w_row[2:]).max(), w_row[1] / int(np.prod(shapes[n])))
        np.testing.assert_allclose(g_row[2:], w_row[2:], rtol=1e-4, atol=1e-4 * scale, err_msg=n)
    monkeypatch.setattr(torch, "randn_like", lambda t, *a_, **k: nz[0].to(cuda))
    with torch.no_grad():
        mu, pi, log_pi, _ = agent.actor(obs)
    for k, v in (("mu", mu), ("pi", pi), ("log_pi", log_pi)):
        np.testing.assert_allclose(v.cpu().numpy(), G[f"head/{k}"], rtol=1e-5, atol=1e-6, err_msg=k)
    # one update() from the golden state (critic, actor + alpha, target EMA at step 0)
    agent = build_agent("sac", "identity", A, as_, cs, cuda, hp=hp)
    queue = [nz[1].to(cuda), nz[2].to(cuda)]
    monkeypatch.setattr(torch, "randn_like", lambda t, *a_, **k: queue.pop(0))
    L = RecLogger()
    agent.update(FakeBuffer(tuple(t.to(cuda) for t in batch)), L, 0)
    torch.cuda.synchronize()
    for k, want in zip([str(k) for k in G["head/log_keys"]], G["head/logs"]):
        atol = RTOL0 * (T.ENT_CONST(A) + abs(want)) if k.endswith("mean_entropy") else RTOL0
        assert L.rows[(0, k)] == pytest.approx(want, rel=RTOL0, abs=atol), k
    cn, an = list(so.critic_shapes(T.OBS, A, hp).keys()), list(so.actor_shapes(T.OBS, A, hp).keys())
    for ck, mod, nm in (("critic_ck", agent.critic, cn), ("actor_ck", agent.actor, an), ("target_ck", agent.critic_target, cn)):
        got = so.checksum({n: p.detach().cpu() for n, p in mod.named_parameters()}, nm)
        np.testing.assert_allclose(got[:, 1], G[f"head/{ck}"][:, 1], rtol=1e-5, err_msg=ck)
        np.testing.assert_allclose(got[:, 2:], G[f"head/{ck}"][:, 2:], rtol=1e-4, atol=2e-6, err_msg=ck)
    assert float(agent.log_alpha) == pytest.approx(float(G["head/log_alpha"][0]), rel=1e-7)


def test_head_layers_hip_graph_update_matches_eager(cuda):
    """The same architecture through captured hipGraphs: 8 SVEA updates, every (actor step?, target step?) variant, equal to
    the eager trajectory (the head CNN's parameters get no gradient in the actor step: the device-step Adam must cope)."""
    import test_oracle_golden_sac as TO
    mh, as_, cs, _, _ = TO.head_case()
    hp, A, B = mh.HP, mh.A, 8
    batches = _svea_batches(cuda, B, A, seed=5)

    class Ring:
        def __init__(self):
            self.i = 0

        def sample(self, n=None, idxs=None):
            self.i += 1
            return tuple(t.clone() for t in batches[self.i % 3])

    snaps = []
    for graph in (False, True):
        torch.manual_seed(21)
        aug.manual_seed(22)
        agent = build_agent("svea", "conv", A, as_, cs, cuda, hp=hp)
        if graph:
            agent.enable_hip_graphs(warmup=2)
        buf = Ring()
        for step in range(8):
            agent.update(buf, None, step)
        torch.cuda.synchronize()
        if graph:
            assert len(agent._graph["graphs"]) >= 2
        snaps.append((agent.critic_group.param.clone(), agent.actor_group.param.clone(), agent.target_group.param.clone()))
    for x, y in zip(*snaps):
        assert bool(torch.isfinite(x).all())
        assert float((x.double() - y.double()).abs().max()) <= 1e-5 * max(1.0, float(x.abs().max()))
