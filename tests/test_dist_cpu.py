"""CPU, world_size=2 (gloo): the data-parallel path of update() -- batch sharding, ONE flat gradient bucket per
optimiser group with shared tensors de-duplicated, sum all-reduce with the 1/world folded into the optimiser
step, logged-scalar averaging -- reproduces the single-process full-batch gradient."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn as nn

from augcl_amd import dist as apdist
from augcl_amd import flat


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


class TwoHeads(nn.Module):
    """A trunk shared by two heads (like the CNN shared by actor and critic, src/algorithms/sac.py:17-32)."""

    def __init__(self):
        super().__init__()
        self.trunk = nn.Sequential(nn.Linear(12, 16), nn.LayerNorm(16), nn.Tanh())
        self.h1 = nn.Linear(16, 1)
        self.h2 = nn.Linear(16, 1)

    def forward(self, x):
        t = self.trunk(x)
        return self.h1(t), self.h2(t)


def _model_and_data():
    torch.manual_seed(0)
    m = TwoHeads()
    rs = np.random.RandomState(1)
    x = torch.from_numpy(rs.standard_normal((8, 12)).astype(np.float32))
    y = torch.from_numpy(rs.standard_normal((8, 1)).astype(np.float32))
    return m, x, y


def _loss(m, x, y):
    a, b = m(x)
    return nn.functional.mse_loss(a, y) + nn.functional.mse_loss(b, y)  # batch MEANS, like sac.py:109


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    red = apdist.init_from_env(backend="gloo")
    assert red.world == world and red.rank == rank
    m, x, y = _model_and_data()
    # de-duplicated flat bucket: every tensor once even if reachable through two modules
    named, seen = [], set()
    for n, p in list(m.named_parameters()) + [("alias." + n, p) for n, p in m.trunk.named_parameters()]:
        if id(p) not in seen:
            seen.add(id(p))
            named.append((n, p))
    grp = flat.FlatGroup(named)
    assert grp.numel >= sum(p.numel() for p in m.parameters())
    a, b = red.shard(x.shape[0])
    loss = _loss(m, x[a:b], y[a:b])
    loss.backward()
    grp.gather_grads()  # one multi-tensor copy of the autograd gradients into the bucket
    scale = red.allreduce_sum_(grp.grad)
    mean_loss = float(red.mean_scalar(loss.detach()))
    shard_of_16 = red.shard(16)
    import pickle
    red2 = pickle.loads(pickle.dumps(red))  # agents are pickled whole; the reducer degrades to a no-op
    q.put((rank, (grp.grad * scale).clone().numpy(), mean_loss, shard_of_16, red2.world))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gradient_equals_full_batch():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    m, x, y = _model_and_data()
    grp = flat.FlatGroup(m.named_parameters())
    full = _loss(m, x, y)
    full.backward()
    grp.gather_grads()
    want = grp.grad.numpy()
    for rank, g, mean_loss, shard16, w2 in res:
        np.testing.assert_allclose(g, want, rtol=1e-5, atol=1e-7)  # mean of shard-mean grads == full-batch grad
        assert mean_loss == pytest.approx(float(full), rel=1e-6)
        assert shard16 == (8 * rank, 8 * rank + 8)
        assert w2 == 1
    np.testing.assert_array_equal(res[0][1], res[1][1])  # replicas see bit-identical reduced gradients


def test_flat_group_views_and_segments():
    m, _, _ = _model_and_data()
    grp = flat.FlatGroup(m.named_parameters())
    for n, p in m.named_parameters():
        a, b = grp.slices[n]
        assert a % 4 == 0  # every tensor 16-byte aligned inside the bucket
        assert p.data.data_ptr() == grp.param[a:b].data_ptr() and p.grad is None
    ta, tb = grp.segment("trunk.")
    ha, hb = grp.segment("h")
    assert ta == 0 and tb == ha and hb == grp.numel
    p0 = m.h1.weight
    p0.data = p0.data.clone()  # something re-pointed the parameter
    grp.rebind()
    assert p0.data.data_ptr() == grp.param[grp.slices["h1.weight"][0]:].data_ptr()


def test_single_process_has_no_reducer(monkeypatch):
    monkeypatch.setenv("WORLD_SIZE", "1")
    assert apdist.init_from_env() is None and apdist.default_reducer() is None


# ---- two optimiser groups, two steps, overlapped exchange ------------------------------------------------------------
class _TorchAdamLaunch(flat.FlatAdam):
    """FlatAdam with its ONE device-specific line (the fused HIP launch) replaced by the same arithmetic in torch, so that
    everything around it -- bucket layout, the two-piece overlapped exchange, 1/world scaling, missing gradients -- runs
    on the CPU under gloo exactly as it does on the GPUs under RCCL."""

    def _launch_adam(self, a, b, step, scale):
        g = self.group
        grad = g.grad[a:b] * scale
        b1, b2 = self.betas
        self.exp_avg[a:b].mul_(b1).add_(grad, alpha=1 - b1)
        self.exp_avg_sq[a:b].mul_(b2).addcmul_(grad, grad, value=1 - b2)
        denom = (self.exp_avg_sq[a:b].sqrt() / (1 - b2 ** step) ** 0.5).add_(self.eps)
        g.param[a:b].addcdiv_(self.exp_avg[a:b], denom, value=-self.lr / (1 - b1 ** step))


def _two_group_trajectory(red, x, y, steps=2):
    """A critic-like group (trunk under the cut + head h1) and an actor-like group (head h2 over the DETACHED trunk), each
    with its own optimiser, `steps` updates in the order of SAC.update: critic step, then the actor step on the
    updated trunk.  Returns the final parameters of both groups."""
    torch.manual_seed(0)
    m = TwoHeads()
    ga = flat.FlatGroup([(n, p) for n, p in m.named_parameters() if not n.startswith("h2.")])
    ga.mark_cut(lambda n: n.startswith("trunk."))
    gb = flat.FlatGroup([(n, p) for n, p in m.named_parameters() if n.startswith("h2.")])
    oa = _TorchAdamLaunch(ga, 1e-2, (0.9, 0.999), reducer=red)
    ob = _TorchAdamLaunch(gb, 1e-2, (0.5, 0.999), reducer=red)
    waits = []
    if red is not None:  # count the exchanges and check they were issued asynchronously in two pieces
        real = red.allreduce_sum_async
        red.allreduce_sum_async = lambda t: (waits.append(t.numel()), real(t))[1]
    for _ in range(steps):
        t = m.trunk(x)
        loss_a = nn.functional.mse_loss(m.h1(t), y)  # batch MEAN, like sac.py:109
        oa.backward_step(loss_a, cut=[t])
        loss_b = nn.functional.mse_loss(m.h2(m.trunk(x).detach()), y)
        ob.backward_step(loss_b)
    return ga.param.clone(), gb.param.clone(), waits, ga


def _worker2(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    red = apdist.init_from_env(backend="gloo")
    _, x, y = _model_and_data()
    a, b = red.shard(x.shape[0])
    pa, pb, waits, ga = _two_group_trajectory(red, x[a:b], y[a:b])
    split = ga.slices[ga.names[ga.cut_index]][0]
    assert waits == [ga.numel - split, split] * 2, waits  # per critic step: the heads' slice first, then the trunk's
    apdist.seed_all(7)
    q.put((rank, pa.numpy(), pb.numpy(), float(torch.rand(1)), apdist.sample_base(red, 4)))
    dist.barrier()
    dist.destroy_process_group()


def test_two_groups_two_steps_overlapped_exchange_matches_single_process():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker2, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    _, x, y = _model_and_data()
    want_a, want_b, _, _ = _two_group_trajectory(None, x, y)
    for rank, pa, pb, draw, base in res:
        np.testing.assert_allclose(pa, want_a.numpy(), rtol=2e-5, atol=1e-7)
        np.testing.assert_allclose(pb, want_b.numpy(), rtol=2e-5, atol=1e-7)
        assert base == 4 * rank  # rank r's first sample in the global batch (augmentations.set_sample_base)
    np.testing.assert_array_equal(res[0][1], res[1][1])  # replicas stay bit-identical
    assert res[0][3] != res[1][3]  # ... while their noise streams differ (dist.seed_all: seed + rank)


class _RecordingLogger:
    def __init__(self):
        self.rows = []

    def log(self, key, value, step, n=1):
        assert isinstance(value, float), f"{key}: the logger is handed host floats, not device tensors ({type(value)})"
        self.rows.append((key, value, step))


def _worker_logs(rank, world, port, q):
    """The host logic of SAC.update()'s logging on CPU tensors: the real SAC._log / _learn_logged / _pack_logs / _emit_logs on a
    bare object (building an agent needs the GPU), a _learn that logs what SVEA + the actor step log, the real GradReducer."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    red = apdist.init_from_env(backend="gloo")
    from augcl_amd.algorithms.sac import SAC
    calls = []
    real = dist.all_reduce

    def counting_all_reduce(t, *a, **k):
        calls.append(tuple(t.shape))
        return real(t, *a, **k)

    dist.all_reduce = counting_all_reduce
    agent = SAC.__new__(SAC)
    agent.device, agent._reducer, agent._graph = torch.device("cpu"), red, None
    keys = ["train_critic/shift_loss", "train_critic/aug_loss", "train_critic/loss", "train_actor/loss",
            "train_actor/mean_entropy", "train_alpha/loss", "train_alpha/value"]

    def learn(t, L, step, do_actor, do_target):
        for i, k in enumerate(keys if do_actor else keys[:3]):
            v = torch.tensor(float(10 * rank + i + step), dtype=torch.float64 if "alpha" in k else torch.float32)
            agent._log(L, k, v, step)

    agent._learn = learn
    L = _RecordingLogger()
    for step, do_actor in ((0, True), (1, False)):
        ks, vec = agent._learn_logged(None, L, step, do_actor, False)
        agent._emit_logs(L, ks, vec, step)
    none_keys, none_vec = agent._learn_logged(None, None, 2, True, False)  # no logger: nothing packed, nothing exchanged
    q.put((rank, L.rows, calls, none_keys, none_vec))
    dist.barrier()
    dist.destroy_process_group()


def test_logged_scalars_one_collective_and_host_floats_per_update():
    """the <= 7 logged scalars of an update() travel as ONE vector -- one all-reduce under data parallelism, one
    device-to-host copy -- and reach L.log(key, value, step) as floats under the reference's keys (src/algorithms/sac.py:113,
    130-131, 142-143; svea.py:57-60), each the mean over the ranks' shard values."""
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_logs, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, rows, calls, none_keys, none_vec in res:
        assert calls == [(7,), (3,)], calls  # exactly one scalar all-reduce per update(), none without a logger
        assert none_keys == () and none_vec is None
        assert [k for k, _, s in rows if s == 0] == ["train_critic/shift_loss", "train_critic/aug_loss", "train_critic/loss",
                                                     "train_actor/loss", "train_actor/mean_entropy", "train_alpha/loss",
                                                     "train_alpha/value"]
        for i, (k, v, s) in enumerate(r for r in rows if r[2] == 0):
            assert v == pytest.approx((0 + i + 10 + i) / 2.0)  # mean of rank 0's i and rank 1's 10 + i
        assert [round(v, 6) for _, v, s in rows if s == 1] == [6.0, 7.0, 8.0]
    assert res[0][1] == res[1][1]  # both ranks log the same numbers
