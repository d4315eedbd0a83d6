"""GPU, world_size 2 on ONE card (gloo): the data-parallel update() of a real agent -- batch sharded over two ranks, in-kernel
augmentation parameters keyed by the global sample index, local-mean losses, the overlapped two-piece gradient exchange,
1/world folded into the fused Adam, the temperature's gradient averaged between the halves of its one-launch step -- walks
the same trajectory as ONE process on the whole batch.  (RCCL needs one GPU per rank; on this box both ranks share GPU 0
and exchange through gloo: everything but the transport is the product path.)  Gate-safe synthetic weights, so that 1e-5 is
a property of the arithmetic."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
OBS, A, B, STEPS = (9, 84, 84), 6, 8, 2


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run(rank, world, port, q):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path[:0] = [root, os.path.join(root, "tests", "golden")]
    import torch.distributed as dist

    from augcl_amd import augmentations as aug
    from augcl_amd import dist as apdist
    from augcl_amd import utils as au
    from augcl_amd.algorithms import make_agent
    from oracle import sac_oracle as so
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    red = None
    if world > 1:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
        red = apdist.init_from_env(backend="gloo")
    hp = dict(so.DEFAULTS)
    cs = so.synth_state(7, so.critic_shapes(OBS, A, hp))
    as_ = so.synth_state(8, so.actor_shapes(OBS, A, hp))
    agent = make_agent(OBS, (A,), au.dotdict(algorithm="svea", data_aug="conv", aug_params=None, **hp))
    csd = {k: v.to(dev) for k, v in cs.items()}
    agent.critic.load_state_dict(csd)
    agent.actor.load_state_dict({k: (csd[k] if so.is_cnn(k) else v.to(dev)) for k, v in as_.items()})
    agent.critic_target.load_state_dict(csd)
    assert (agent._reducer is None) == (world == 1)
    rs = np.random.RandomState(3)
    G = B * 2  # global batch; rank r owns rows [r*G/world, (r+1)*G/world)
    obs = rs.randint(0, 256, (G, *OBS)).astype(np.float32)
    nobs = rs.randint(0, 256, (G, *OBS)).astype(np.float32)
    act = rs.uniform(-1, 1, (G, A)).astype(np.float32)
    rew = rs.uniform(0, 4, (G, 1)).astype(np.float32)
    noise = [rs.standard_normal((G, A)).astype(np.float32) for _ in range(2 * STEPS)]
    lo, hi = (rank * G // world, (rank + 1) * G // world)
    t = lambda a: torch.from_numpy(a[lo:hi]).to(dev)  # noqa: E731
    queue = [t(n) for n in noise]
    torch.randn_like = lambda x, *a_, **k: queue.pop(0).clone()

    class Buf:
        batch_size = hi - lo

        def sample(self, n=None, idxs=None):
            return t(obs), t(act), t(rew), t(nobs), torch.ones((hi - lo, 1), device=dev)

    class Log:
        def __init__(self):
            self.rows = {}

        def log(self, k, v, step, n=1):
            self.rows[(step, k)] = float(v)

    L = Log()
    aug.manual_seed(11)  # ONE seed on every rank: the shards are told apart by the sample base update() sets
    for step in range(STEPS):
        agent.update(Buf(), L, step)
    torch.cuda.synchronize()
    q.put((rank, agent.critic_group.param.cpu().numpy(), agent.actor_group.param.cpu().numpy(), agent.target_group.param.cpu().numpy(),
           float(agent.log_alpha), dict(L.rows)))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def _launch(world):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_run, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=300) for _ in range(world)], key=lambda x: x[0])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    return res


def test_two_rank_update_equals_single_process(cuda):
    one = _launch(1)[0]
    two = _launch(2)
    for name, i in (("critic", 1), ("actor", 2), ("target", 3)):
        np.testing.assert_array_equal(two[0][i], two[1][i], err_msg=f"{name}: replicas diverged")
        scale = np.abs(one[i]).max()
        assert np.abs(two[0][i] - one[i]).max() <= 2e-5 * scale, (name, np.abs(two[0][i] - one[i]).max(), scale)
    assert two[0][4] == two[1][4] and abs(two[0][4] - one[4]) <= 1e-7 * abs(one[4])
    for k, v in one[5].items():  # logged scalars are averaged over ranks = the full-batch means
        assert abs(two[0][5][k] - v) <= 1e-5 * max(1.0, abs(v)), (k, two[0][5][k], v)
        assert two[0][5][k] == two[1][5][k]
