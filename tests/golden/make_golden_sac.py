#!/usr/bin/env python3
"""Mint golden vectors for the network stack and update() from the UNMODIFIED reference.

Run:  PYTHONDONTWRITEBYTECODE=1 python3 -O tests/golden/make_golden_sac.py

Same import recipe as make_golden.py (placeholder modules for absent third-party packages,
.cuda() neutralised).  For every algorithm on the hot path the reference agent is built with
make_agent(), given deterministic synthetic weights (oracle.sac_oracle.synth_state -- torch's own
orthogonal init is LAPACK-dependent), fed a seeded batch, and stepped K times with
  * actor noise injected by patching torch.randn_like (modules.py:231 is its only call site),
  * augmentation parameters drawn from torch's default generator after torch.manual_seed(s):
    the stand-in RandomCrop records its offsets, random_conv's per-sample torch.randn(3,3,3,3)
    draws are replayed from the same seed.
Only numbers are stored (logged losses, Q values, parameter fingerprints, the drawn parameters).
"""
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

OBS = (9, 84, 84)
K_STEPS = 3
FIRST_STEP = 2
B = 2
N_SEEDS = 10  # candidate scenario seeds per case; the one with the widest ReLU gate margin is kept (see run_case)
CASES = [  # (name, algorithm, data_aug, action_dim)
    ("sac_cartpole", "sac", "identity", 1),            # config 1 (no aug)
    ("nnrad_shift_cartpole", "non_naive_rad", "shift", 1),  # config 1: SAC + random_shift
    ("rad_shift", "rad", "shift", 2),
    ("svea_conv_walker", "svea", "conv", 6),           # config 2
    ("svea_overlay", "svea", "overlay", 6),
    ("drq_finger", "drq", "identity", 2),              # config 3
    ("augcl_conv_cheetah", "augcl", "conv", 6),        # config 4
    ("drq_k2m2", "drq", "identity", 2),                # DrQ with K = M = 2 (drq.py:15-65): variance logs live
]
HP_OVERRIDES = {"drq_k2m2": dict(drq_k=2, drq_m=2)}


class RecLogger:
    def __init__(self):
        self.rows = {}

    def log(self, key, value, step, n=1):
        self.rows[(step, key)] = float(value.item() if torch.is_tensor(value) else value)


class FakeBuffer:
    def __init__(self, batch):
        self.batch = batch

    def sample(self, n=None, idxs=None):
        o, a, r, no, nd = self.batch
        return o.clone(), a.clone(), r.clone(), no.clone(), nd.clone()


def main(_what="all"):
    RandomCrop, _ = mg._install_stubs()
    import augmentations as ref_aug
    import utils as ref_utils
    from algorithms.factory import make_agent

    real_randn_like = torch.randn_like
    out = {}

    class GateMargin:
        """min |pre-activation| over every ReLU input seen during a run.  ReLU is continuous in value but its
        GRADIENT is not: an input within f32 rounding (~1e-8..1e-7 here) of zero can gate differently under another
        summation order (MFMA vs MKL-DNN), which perturbs that update by ~1e-3 relative -- for ANY independent
        implementation, the reference on a GPU included.  Scenarios are therefore chosen (among N_SEEDS seeded
        candidates) to keep every gate clear of zero, and the margin is stored with the vectors."""

        def __init__(self):
            self.m = float("inf")
            self.handles = []

        def watch(self, *modules):
            for mod in modules:
                for sub in mod.modules():
                    if isinstance(sub, torch.nn.ReLU):
                        self.handles.append(sub.register_forward_hook(self._hook))

        def _hook(self, mod, inp, outp):
            self.m = min(self.m, float(inp[0].detach().abs().min()))

        def close(self):
            for h in self.handles:
                h.remove()

    def run_case(ci, name, algo, data_aug, A, seed):
        res = {}
        hp = dict(so.DEFAULTS, **HP_OVERRIDES.get(name, {}))
        K, M = hp["drq_k"], hp["drq_m"]
        args = ref_utils.dotdict(algorithm=algo, data_aug=data_aug, aug_params=None, **hp)
        actor_state = so.synth_state(seed, so.actor_shapes(OBS, A, hp))
        critic_state = so.synth_state(seed + 1000, so.critic_shapes(OBS, A, hp))
        for k in critic_state:  # shared CNN: actor sees the critic's tensors
            if so.is_cnn(k):
                actor_state[k] = critic_state[k]

        def build(algorithm):
            a = ref_utils.dotdict(dict(args, algorithm=algorithm))
            ag = make_agent(OBS, (A,), a)
            ag.critic.load_state_dict(critic_state)
            ag.actor.load_state_dict(actor_state)  # CNN tensors are shared objects: same values either way
            ag.critic_target.load_state_dict(critic_state)
            return ag

        if algo == "augcl":
            prev = build("non_naive_rad")
            agent = make_agent(OBS, (A,), args)
            agent.load_pretrained_agent(prev)
        else:
            agent = build(algo)
        gm = GateMargin()
        gm.watch(agent.actor, agent.critic, agent.critic_target, *( [agent.critic_weak] if algo == "augcl" else []))

        ob, ac, rw, nob, nd = gi.transition_batch(seed + 7, B, A)
        batch = tuple(torch.from_numpy(v.astype(np.float32)) for v in (ob, ac, rw, nob, nd))
        buf = FakeBuffer(batch)
        L = RecLogger()
        ov_imgs = torch.from_numpy(gi.overlay_imgs(seed + 9, B))
        ref_aug.dataloader = object()
        ref_aug._get_data_batch = lambda batch_size: ov_imgs.clone()

        noise_rs = np.random.RandomState(seed + 3)
        n_noise = (M + 1) * K_STEPS if algo == "drq" else 2 * K_STEPS  # one actor call per next_obs copy + the actor update
        noises = [torch.from_numpy(noise_rs.standard_normal((B, A)).astype(np.float32)) for _ in range(n_noise)]
        queue = list(noises)
        torch.randn_like = lambda t, *a, **k: queue.pop(0).clone()
        used_noise, oys, oxs, oys2, oxs2, cws, qs, all_draws = [], [], [], [], [], [], [], []
        # On a GPU `t.cpu()` COPIES; here everything already lives on the CPU, where it returns `t` itself and
        # `.numpy()` then aliases it -- drq.py:29-33 keeps such views of target_Q and goes on modifying it in place,
        # so the variance it logs would be computed from aliased (all equal) arrays.  Restore device semantics.
        real_cpu = torch.Tensor.cpu
        if algo == "drq" and (K > 1 or M > 1):
            torch.Tensor.cpu = lambda self, *a, **k: self.clone()
        try:
            for s in range(K_STEPS):
                step = FIRST_STEP + s
                n_before = len(queue)
                torch.manual_seed(5000 + 10 * ci + s)
                agent.update(buf, L, step)
                # replay the default-generator draws of this update() (order: shift(s), then conv weights)
                torch.manual_seed(5000 + 10 * ci + s)
                z = np.zeros(B, np.int32)
                oy = ox = oy2 = ox2 = z
                uses_shift = algo in ("svea", "drq", "augcl") or data_aug == "shift"
                if uses_shift:
                    oy = torch.randint(0, 9, (B,)).numpy().astype(np.int32)
                    ox = torch.randint(0, 9, (B,)).numpy().astype(np.int32)
                if algo == "drq" and (K > 1 or M > 1):
                    # K draws for obs (the first was replayed above), then M for next_obs: (K+M, 2, B)
                    draws = [(oy, ox)] + [(torch.randint(0, 9, (B,)).numpy().astype(np.int32),
                                           torch.randint(0, 9, (B,)).numpy().astype(np.int32)) for _ in range(K + M - 1)]
                    assert np.array_equal(RandomCrop.last[0], draws[-1][0]) and np.array_equal(RandomCrop.last[1], draws[-1][1])
                    oy2, ox2 = draws[K]
                    all_draws.append(np.array(draws))
                elif algo == "drq" or (algo == "rad" and data_aug == "shift"):
                    oy2 = torch.randint(0, 9, (B,)).numpy().astype(np.int32)
                    ox2 = torch.randint(0, 9, (B,)).numpy().astype(np.int32)
                    assert np.array_equal(RandomCrop.last[0], oy2) and np.array_equal(RandomCrop.last[1], ox2)
                elif uses_shift:
                    assert np.array_equal(RandomCrop.last[0], oy) and np.array_equal(RandomCrop.last[1], ox)
                cw = np.zeros((B, 3, 3, 3, 3), np.float32)
                if data_aug == "conv":
                    cw = torch.stack([torch.randn(3, 3, 3, 3) for _ in range(B)]).numpy()
                oys.append(oy); oxs.append(ox); oys2.append(oy2); oxs2.append(ox2); cws.append(cw)
                used_noise.append(n_before - len(queue))
                margin = gm.m  # the probe below is not part of the scenario
                with torch.no_grad():
                    q1, q2 = agent.critic(batch[0], batch[1])
                gm.m = margin
                qs.append(np.stack([q1.numpy().ravel(), q2.numpy().ravel()]))
        finally:
            torch.randn_like = real_randn_like
            torch.Tensor.cpu = real_cpu
            gm.close()
        keys = sorted({k for (_, k) in L.rows})
        logs = np.full((K_STEPS, len(keys)), np.nan)
        for (st, k), v in L.rows.items():
            logs[st - FIRST_STEP, keys.index(k)] = v
        pre = name + "/"
        res[pre + "seed"] = np.array([seed])
        res[pre + "gate_margin"] = np.array([gm.m])
        res[pre + "log_keys"] = np.array(keys)
        res[pre + "logs"] = logs
        res[pre + "noise_used"] = np.array(used_noise)
        res[pre + "oy"], res[pre + "ox"] = np.stack(oys), np.stack(oxs)
        res[pre + "oy2"], res[pre + "ox2"] = np.stack(oys2), np.stack(oxs2)
        if data_aug == "conv":
            res[pre + "conv_w"] = np.stack(cws)
        if all_draws:
            res[pre + "shift_draws"] = np.stack(all_draws)  # (steps, K+M, 2, B)
        res[pre + "q_after"] = np.stack(qs)
        cn = list(so.critic_shapes(OBS, A, hp).keys())
        an = list(so.actor_shapes(OBS, A, hp).keys())
        res[pre + "critic_ck"] = so.checksum(dict(agent.critic.named_parameters()), cn)
        res[pre + "actor_ck"] = so.checksum(dict(agent.actor.named_parameters()), an)
        res[pre + "target_ck"] = so.checksum(dict(agent.critic_target.named_parameters()), cn)
        if algo == "augcl":
            res[pre + "weak_ck"] = so.checksum(dict(agent.critic_weak.named_parameters()), cn)
        res[pre + "log_alpha"] = np.array([agent.log_alpha.item()], dtype=np.float64)
        return res, gm.m, {k: np.round(logs[:, i], 5).tolist() for i, k in enumerate(keys)}

    for ci, (name, algo, data_aug, A) in enumerate(CASES):
        best = None
        for trial in range(N_SEEDS):
            seed = 100 + ci + 37 * trial
            res, margin, summary = run_case(ci, name, algo, data_aug, A, seed)
            if best is None or margin > best[1]:
                best = (res, margin, summary, seed)
        out.update(best[0])
        print(name, "seed", best[3], "gate margin %.2e" % best[1], best[2])

    # module-level golden: forward values + input/parameter gradients of encoder, actor, critic (A=6)
    A, hp = 6, dict(so.DEFAULTS)
    args = ref_utils.dotdict(algorithm="sac", data_aug="identity", aug_params=None, **hp)
    agent = make_agent(OBS, (A,), args)
    cs = so.synth_state(31, so.critic_shapes(OBS, A, hp))
    as_ = so.synth_state(32, so.actor_shapes(OBS, A, hp))
    for k in cs:
        if so.is_cnn(k):
            as_[k] = cs[k]
    agent.critic.load_state_dict(cs)
    agent.actor.load_state_dict(as_)
    ob, ac, _, _, _ = gi.transition_batch(33, 3, A)
    x = torch.from_numpy(ob.astype(np.float32)).requires_grad_(True)
    a = torch.from_numpy(ac)
    feat = agent.critic.encoder.shared_cnn(x)
    out["nets/cnn_out"] = feat.detach().numpy()
    q1, q2 = agent.critic(x, a)
    (q1.sum() + 2.0 * q2.sum()).backward()
    out["nets/q1"], out["nets/q2"] = q1.detach().numpy(), q2.detach().numpy()
    out["nets/dx"] = x.grad.numpy().copy()
    cn = list(so.critic_shapes(OBS, A, hp).keys())
    out["nets/critic_grad_ck"] = so.checksum({n: p.grad for n, p in agent.critic.named_parameters()}, cn)
    noise = torch.from_numpy(np.random.RandomState(34).standard_normal((3, A)).astype(np.float32))
    torch.randn_like = lambda t, *a_, **k: noise.clone()
    try:
        mu, pi, log_pi, log_std = agent.actor(x.detach())
    finally:
        torch.randn_like = real_randn_like
    out["nets/mu"], out["nets/pi"] = mu.detach().numpy(), pi.detach().numpy()
    out["nets/log_pi"], out["nets/log_std"] = log_pi.detach().numpy(), log_std.detach().numpy()
    np.savez_compressed(os.path.join(HERE, "sac_golden.npz"), **out)
    print("wrote sac_golden.npz", sum(v.nbytes for v in out.values()) // 1024, "KiB")


if __name__ == "__main__":
    main()
