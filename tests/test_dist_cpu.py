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
