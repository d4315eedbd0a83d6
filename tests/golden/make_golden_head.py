#!/usr/bin/env python3
"""Golden vectors for agents WITH head layers (num_head_layers = 2: src/algorithms/modules.py:168-181, HeadCNN = [ReLU, Conv]* on
the shared CNN's output), minted from the UNMODIFIED reference:  PYTHONDONTWRITEBYTECODE=1 python3 -O tests/golden/make_golden_head.py

Same import recipe and conventions as make_golden_sac.py (placeholder modules for absent third-party packages, .cuda()
neutralised, deterministic gate-safe synthetic weights, actor noise injected through torch.randn_like).  Stored: the
encoder / critic / actor forward values, the critic's parameter-gradient checksum, and one SAC update() (critic, actor + alpha,
target EMA) -- logged scalars and parameter checksums.  Numbers only."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)
import inputs as gi  # noqa: E402
import make_golden as mg  # noqa: E402
from oracle import sac_oracle as so  # noqa: E402

OBS, A, B = (9, 84, 84), 6, 3
HP = dict(so.DEFAULTS, num_shared_layers=8, num_head_layers=2)
SEEDS = dict(critic=51, actor=52, batch=53, noise=54)


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


def states():
    cs = so.synth_state(SEEDS["critic"], so.critic_shapes(OBS, A, HP))
    as_ = so.synth_state(SEEDS["actor"], so.actor_shapes(OBS, A, HP))
    for k in cs:
        if so.is_cnn(k):
            as_[k] = cs[k]
    return as_, cs


def batch():
    ob, ac, rw, nob, nd = gi.transition_batch(SEEDS["batch"], B, A)
    return tuple(torch.from_numpy(np.ascontiguousarray(v, dtype=np.float32)) for v in (ob, ac, rw, nob, nd))


def noises():
    rs = np.random.RandomState(SEEDS["noise"])
    return [torch.from_numpy(rs.standard_normal((B, A)).astype(np.float32)) for _ in range(3)]


def main():
    mg._install_stubs()
    import utils as ref_utils
    from algorithms.factory import make_agent

    real_randn_like = torch.randn_like
    out = {}
    args = ref_utils.dotdict(algorithm="sac", data_aug="identity", aug_params=None, **HP)
    agent = make_agent(OBS, (A,), args)
    as_, cs = states()
    agent.critic.load_state_dict(cs)
    agent.actor.load_state_dict(as_)
    agent.critic_target.load_state_dict(cs)
    assert agent.actor.encoder.head_cnn is agent.critic.encoder.head_cnn  # shared, like the CNN below it (sac.py:20-33)
    obs, act, rew, nobs, nd = batch()
    x = obs.clone().requires_grad_(True)
    out["head/features"] = agent.critic.encoder.head_cnn(agent.critic.encoder.shared_cnn(x)).detach().numpy()
    q1, q2 = agent.critic(x, act)
    (q1.sum() + 2.0 * q2.sum()).backward()
    out["head/q1"], out["head/q2"] = q1.detach().numpy(), q2.detach().numpy()
    cn = list(so.critic_shapes(OBS, A, HP).keys())
    out["head/critic_grad_ck"] = so.checksum({n: p.grad for n, p in agent.critic.named_parameters()}, cn)
    for p in agent.critic.parameters():
        p.grad = None
    nz = noises()
    torch.randn_like = lambda t, *a_, **k: nz[0].clone()
    try:
        mu, pi, log_pi, log_std = agent.actor(obs)
    finally:
        torch.randn_like = real_randn_like
    out["head/mu"], out["head/pi"], out["head/log_pi"] = mu.detach().numpy(), pi.detach().numpy(), log_pi.detach().numpy()
    # one update(): critic (noise 1 feeds calculate_target_Q's policy sample), actor + alpha (noise 2), target EMA (step 0)
    queue = [nz[1].clone(), nz[2].clone()]
    torch.randn_like = lambda t, *a_, **k: queue.pop(0)
    L = RecLogger()
    try:
        agent.update(FakeBuffer((obs, act, rew, nobs, nd)), L, 0)
    finally:
        torch.randn_like = real_randn_like
    keys = sorted(k for _, k in L.rows)
    out["head/log_keys"] = np.array(keys)
    out["head/logs"] = np.array([L.rows[(0, k)] for k in keys], dtype=np.float64)
    an = list(so.actor_shapes(OBS, A, HP).keys())
    out["head/critic_ck"] = so.checksum(dict(agent.critic.named_parameters()), cn)
    out["head/actor_ck"] = so.checksum(dict(agent.actor.named_parameters()), an)
    out["head/target_ck"] = so.checksum(dict(agent.critic_target.named_parameters()), cn)
    out["head/log_alpha"] = np.array([agent.log_alpha.item()], dtype=np.float64)
    np.savez_compressed(os.path.join(HERE, "head_golden.npz"), **out)
    print("wrote head_golden.npz", sum(v.nbytes for v in out.values()) // 1024, "KiB", dict(zip(keys, np.round(out["head/logs"], 5))))


if __name__ == "__main__":
    main()
