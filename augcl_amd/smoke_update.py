"""One tiny SVEA update() on cuda:0 (HIP augmentations + HIP encoder + fused Adam/EMA), checked against the
CPU oracle on the same seeded batch, injected noise and augmentation parameters.  Called by
__graft_entry__.smoke(); imports the oracle ONLY as the checker."""
import numpy as np
import torch


def run(dev):
    from oracle import aug_oracle as ao
    from oracle import sac_oracle as so

    from . import augmentations as aug
    from . import utils as au
    from .algorithms import make_agent

    OBS, A, B = (9, 84, 84), 6, 2
    hp = dict(so.DEFAULTS)
    cs = so.synth_state(7, so.critic_shapes(OBS, A, hp))
    as_ = so.synth_state(8, so.actor_shapes(OBS, A, hp))
    for k in cs:
        if so.is_cnn(k):
            as_[k] = cs[k]
    rs = np.random.RandomState(9)
    obs = rs.randint(0, 256, (B, *OBS)).astype(np.float32)
    nobs = rs.randint(0, 256, (B, *OBS)).astype(np.float32)
    act = rs.uniform(-1, 1, (B, A)).astype(np.float32)
    rew = rs.uniform(0, 4, (B, 1)).astype(np.float32)
    nd = np.ones((B, 1), np.float32)
    noises = [torch.from_numpy(rs.standard_normal((B, A)).astype(np.float32)) for _ in range(2)]
    oy, ox = rs.randint(0, 9, B).astype(np.int32), rs.randint(0, 9, B).astype(np.int32)
    cw = rs.standard_normal((B, 3, 3, 3, 3)).astype(np.float32)
    t = torch.from_numpy

    oracle = so.OracleAgent("svea", OBS, A, as_, cs)
    want = oracle.update(tuple(map(t, (obs, act, rew, nobs, nd))), 0, noises[0], noises[1],
                         aug=lambda x: t(ao.random_conv_c(x.numpy(), cw)), shift=lambda x: t(ao.random_shift_c(x.numpy(), oy, ox)))

    args = au.dotdict(algorithm="svea", data_aug="conv", aug_params=None, **hp)
    agent = make_agent(OBS, (A,), args)
    agent.critic.load_state_dict({k: v.to(dev) for k, v in cs.items()})
    agent.actor.load_state_dict({k: v.to(dev) for k, v in as_.items()})
    agent.critic_target.load_state_dict({k: v.to(dev) for k, v in cs.items()})
    agent.fuse_front_end = False
    agent.aug_funcs["conv"]["func"] = lambda x: aug.random_conv(x, weights=cw)
    real_shift, real_randn = aug.random_shift, torch.randn_like
    queue = [n.to(dev) for n in noises]

    class Buf:
        def sample(self):
            return tuple(t(v).to(dev) for v in (obs, act, rew, nobs, nd))

    class Log:
        rows = {}

        def log(self, k, v, step):
            self.rows[k] = float(v)

    aug.random_shift = lambda x, pad=4, out=None: real_shift(x, pad, oy=oy, ox=ox, out=out)
    torch.randn_like = lambda x, *a, **k: queue.pop(0).clone()
    try:
        agent.update(Buf(), Log(), 0)
    finally:
        aug.random_shift, torch.randn_like = real_shift, real_randn
    torch.cuda.synchronize()
    for k in ("train_critic/loss", "train_critic/aug_loss", "train_actor/loss", "train_alpha/loss"):
        got, w = Log.rows[k], want[k]
        assert abs(got - w) <= 1e-5 * max(1.0, abs(w)), f"SVEA update mismatch vs oracle: {k} {got} vs {w}"
    with torch.no_grad():
        q1, _ = agent.critic(t(obs).to(dev), t(act).to(dev))
        o1, _ = so.critic_forward(oracle.critic, t(obs), t(act), oracle.hp)
    err = float((q1.cpu() - o1).abs().max())
    assert err <= 1e-4, f"post-update Q mismatch vs oracle: {err}"
    print(f"smoke update ok: SVEA step matches the oracle (critic loss {Log.rows['train_critic/loss']:.6f}, post-update max|dQ| {err:.1e})")
