"""oracle/sac_oracle.py -- CPU (torch float32) restatement of the network stack and the
SAC / RAD / Non-Naive-RAD / SVEA / DrQ / AugCL update() of the reference.

TEST INFRASTRUCTURE ONLY (tests/, __graft_entry__.smoke(), bench.py's cpu_baseline leg).
Nothing under augcl_amd/ imports this file.

Written functionally (explicit parameter dicts, explicit noise / augmentation parameters,
hand-written Adam) so it shares no structure with the product's nn.Module code, and pinned
against outputs of the reference itself run in the build container
(tests/golden/make_golden_sac.py -> tests/golden/sac_golden.npz): losses, Q values and
parameter checksums of k update steps agree to <= 1e-5 (tests/test_oracle_golden_sac.py).

Reference lines restated:
  src/algorithms/modules.py:19-32   gaussian_logprob, squash
  src/algorithms/modules.py:145-165 SharedCNN (x/255, conv3x3 s2, then [ReLU, conv3x3 s1]*)
  src/algorithms/modules.py:101-111 RLProjection (Linear, LayerNorm, Tanh)
  src/algorithms/modules.py:184-197 Encoder (detach between CNN and projection)
  src/algorithms/modules.py:200-276 Actor, QFunction, Critic
  src/algorithms/sac.py:94-164      calculate_target_Q, update_critic, update_actor_and_alpha,
                                    soft_update_critic_target, update
  src/algorithms/rad.py:24-36, non_naive_rad.py:8-18, svea.py:15-76, drq.py:15-86, augcl.py:16-89
  src/utils.py:76-78                soft_update_params
Hyper-parameter defaults: src/arguments.py:22-67.
"""
import copy
import math

import numpy as np
import torch
import torch.nn.functional as F

DEFAULTS = dict(discount=0.99, critic_tau=0.01, encoder_tau=0.05, actor_update_freq=2,
                critic_target_update_freq=2, num_shared_layers=11, num_head_layers=0, num_filters=32,
                projection_dim=100, hidden_dim=1024, actor_log_std_min=-10.0, actor_log_std_max=2.0,
                init_temperature=0.1, actor_lr=1e-3, actor_beta=0.9, critic_lr=1e-3, critic_beta=0.9,
                alpha_lr=1e-4, alpha_beta=0.5, svea_alpha=0.5, svea_beta=0.5, drq_k=1, drq_m=1)


# ------------------------------------------------------------------ parameter inventory
def conv_layer_ids(n_shared):
    """nn.Sequential indices of the conv layers in SharedCNN.layers (modules.py:151-158)."""
    return [2 + 2 * i for i in range(n_shared)]


def cnn_out_hw(h, n_layers):
    h = (h - 3) // 2 + 1
    for _ in range(n_layers - 1):
        h -= 2
    return h


def critic_shapes(obs_shape, A, hp):
    C, H, W = obs_shape
    nf = hp["num_filters"]
    shapes = {}
    for li, idx in enumerate(conv_layer_ids(hp["num_shared_layers"])):
        shapes[f"encoder.shared_cnn.layers.{idx}.weight"] = (nf, C if li == 0 else nf, 3, 3)
        shapes[f"encoder.shared_cnn.layers.{idx}.bias"] = (nf,)
    for li in range(hp["num_head_layers"]):
        shapes[f"encoder.head_cnn.layers.{2 * li + 1}.weight"] = (nf, nf, 3, 3)
        shapes[f"encoder.head_cnn.layers.{2 * li + 1}.bias"] = (nf,)
    hw = cnn_out_hw(H, hp["num_shared_layers"]) - 2 * hp["num_head_layers"]
    flat = nf * hw * hw
    pd, hd = hp["projection_dim"], hp["hidden_dim"]
    shapes["encoder.projection.projection.0.weight"] = (pd, flat)
    shapes["encoder.projection.projection.0.bias"] = (pd,)
    shapes["encoder.projection.projection.1.weight"] = (pd,)
    shapes["encoder.projection.projection.1.bias"] = (pd,)
    for q in ("Q1", "Q2"):
        shapes[f"{q}.trunk.0.weight"] = (hd, pd + A)
        shapes[f"{q}.trunk.0.bias"] = (hd,)
        shapes[f"{q}.trunk.2.weight"] = (hd, hd)
        shapes[f"{q}.trunk.2.bias"] = (hd,)
        shapes[f"{q}.trunk.4.weight"] = (1, hd)
        shapes[f"{q}.trunk.4.bias"] = (1,)
    return shapes


def actor_shapes(obs_shape, A, hp):
    cs = critic_shapes(obs_shape, A, hp)
    shapes = {k: v for k, v in cs.items() if k.startswith("encoder.")}
    pd, hd = hp["projection_dim"], hp["hidden_dim"]
    shapes["mlp.0.weight"] = (hd, pd)
    shapes["mlp.0.bias"] = (hd,)
    shapes["mlp.2.weight"] = (hd, hd)
    shapes["mlp.2.bias"] = (hd,)
    shapes["mlp.4.weight"] = (2 * A, hd)
    shapes["mlp.4.bias"] = (2 * A,)
    return shapes


def is_cnn(name):
    return name.startswith("encoder.shared_cnn.") or name.startswith("encoder.head_cnn.")


def synth_state(seed, shapes, gate_safe=True):
    """Deterministic, box-independent weights (np.random.RandomState; torch's own init is LAPACK-dependent).

    gate_safe=True builds a WELL-CONDITIONED conv stack for parity tests: every conv channel gets a bias of
    +-2.5 (random sign) and the weights are scaled so that |sum w*a| stays ~0.3, i.e. every ReLU input sits
    >= ~1 away from zero -- half the gates firmly open, half firmly closed.  ReLU's gradient is discontinuous
    at 0: with He-scaled weights some of the ~10^7 pre-activations of an update land within f32 rounding of 0
    and gate differently under a different summation order (MFMA vs MKL-DNN; the reference on a GPU would too),
    which moves that update by ~1e-3 relative however correct both implementations are.  Keeping the gates clear
    of zero makes 1e-5 parity a property of the arithmetic instead of luck.  (gate_safe=False: He-like scale.)"""
    rs = np.random.RandomState(seed)
    out = {}
    for name in sorted(shapes):
        shp = shapes[name]
        cnn = is_cnn(name)
        if name.endswith("projection.1.weight"):
            v = 1.0 + 0.1 * rs.standard_normal(shp)
        elif name.endswith(".bias"):
            if cnn and gate_safe:
                v = 2.5 * np.where(rs.random_sample(shp) < 0.5, -1.0, 1.0) + 0.2 * rs.standard_normal(shp)
            else:
                v = 0.05 * rs.standard_normal(shp)
        else:
            fan_in = int(np.prod(shp[1:]))
            if cnn and gate_safe:
                # inputs: x/255 (rms ~0.58, all channels) for the first layer, else ~half the channels live at ~2.5-3
                rms_in = 0.58 if shp[1] != shp[0] else 3.0 * math.sqrt(0.5)
                v = rs.standard_normal(shp) * (0.3 / (math.sqrt(fan_in) * rms_in))
            else:
                v = rs.standard_normal(shp) * math.sqrt(2.0 / fan_in)
        out[name] = torch.from_numpy(np.ascontiguousarray(v, dtype=np.float32))
    return out


def _orthogonal(rs, rows, cols, gain=1.0):
    """nn.init.orthogonal_ restated without LAPACK (modules.py:52-66 calls it for every Linear and for the centre
    tap of every Conv2d): a (rows, cols) matrix with orthonormal rows (rows <= cols) or columns (rows > cols) times
    gain, from a seeded normal matrix by modified Gram-Schmidt with one re-orthogonalisation pass, float64."""
    n, k = max(rows, cols), min(rows, cols)
    a = rs.standard_normal((n, k))
    q = np.zeros((n, k))
    for j in range(k):
        v = a[:, j].copy()
        for _ in range(2):
            v -= q[:, :j] @ (q[:, :j].T @ v)
        q[:, j] = v / np.linalg.norm(v)
    return gain * (q if rows > cols else q.T)


def real_init_state(seed, shapes):
    """The reference's ACTUAL initial state (weight_init, modules.py:52-66), seeded and box-independent: Linear =
    orthogonal weight, zero bias; Conv2d = all taps zero except the centre one, which is an orthogonal (out, in) matrix
    times the ReLU gain sqrt(2), zero bias; LayerNorm = (1, 0)."""
    rs = np.random.RandomState(seed)
    out = {}
    for name in sorted(shapes):
        shp = shapes[name]
        if name.endswith("projection.1.weight"):
            v = np.ones(shp)
        elif name.endswith(".bias"):
            v = np.zeros(shp)
        elif len(shp) == 4:
            v = np.zeros(shp)
            v[:, :, shp[2] // 2, shp[3] // 2] = _orthogonal(rs, shp[0], shp[1], math.sqrt(2.0))
        else:
            v = _orthogonal(rs, shp[0], shp[1])
        out[name] = torch.from_numpy(np.ascontiguousarray(v, dtype=np.float32))
    return out


# ------------------------------------------------------------------ networks (functional)
class _GatedReLU(torch.autograd.Function):
    """ReLU whose gate is given from outside: y = z * gate, dz = dy * gate (see `_relu`)."""

    @staticmethod
    def forward(ctx, z, gate):
        ctx.save_for_backward(gate)
        return z * gate

    @staticmethod
    def backward(ctx, g):
        (gate,) = ctx.saved_tensors
        return g * gate, None


def _relu(h, gates, key, i):
    """torch.relu, or -- when `gates` (a dict of lists of 0/1 tensors) has an entry for this ReLU -- the same function
    with the on/off decision taken from outside.  ReLU's derivative jumps at 0: a pre-activation within float32
    rounding of zero can gate differently in two correct float32 evaluations, and one such flip moves the gradients
    below it by ~1/sqrt(#samples).  The large-batch parity tests hand the float64 referee the gates of the
    implementation under test, so that what is left to compare is arithmetic (tests/test_referee_gpu.py)."""
    if gates is None or key not in gates:
        return torch.relu(h)
    g = gates[key][i].to(device=h.device, dtype=h.dtype)
    if "audit" in gates:  # how the imposed gates sit on THIS evaluation's pre-activations: (key, i, #differing, #gates,
        with torch.no_grad():  # max |pre-activation| / rms at a differing gate) -- a wrong gate far from 0 would show here
            diff = (g > 0) != (h > 0)
            n = int(diff.sum())
            worst = float(h[diff].abs().max() / h.square().mean().sqrt().clamp_min(1e-300)) if n else 0.0
            gates["audit"].append((key, i, n, h.numel(), worst))
    return _GatedReLU.apply(h, g)


def cnn_forward(p, x, hp, gates=None):
    h = x / 255.0
    ids = conv_layer_ids(hp["num_shared_layers"])
    for li, idx in enumerate(ids):
        if li > 0:
            h = _relu(h, gates, "cnn", li - 1)
        h = F.conv2d(h, p[f"encoder.shared_cnn.layers.{idx}.weight"], p[f"encoder.shared_cnn.layers.{idx}.bias"],
                     stride=2 if li == 0 else 1)
    for li in range(hp["num_head_layers"]):
        h = torch.relu(h)
        h = F.conv2d(h, p[f"encoder.head_cnn.layers.{2 * li + 1}.weight"], p[f"encoder.head_cnn.layers.{2 * li + 1}.bias"])
    return h.flatten(1)


def encode(p, x, hp, detach=False, gates=None):
    h = cnn_forward(p, x, hp, gates)
    if detach:
        h = h.detach()
    h = F.linear(h, p["encoder.projection.projection.0.weight"], p["encoder.projection.projection.0.bias"])
    h = F.layer_norm(h, (h.shape[-1],), p["encoder.projection.projection.1.weight"],
                     p["encoder.projection.projection.1.bias"], 1e-5)
    return torch.tanh(h)


def actor_forward(p, x, hp, noise=None, detach=False, gates=None):
    """returns mu, pi, log_pi, log_std (modules.py:216-244). noise=None => compute_pi=False."""
    h = encode(p, x, hp, detach, gates)
    h = _relu(F.linear(h, p["mlp.0.weight"], p["mlp.0.bias"]), gates, "mlp", 0)
    h = _relu(F.linear(h, p["mlp.2.weight"], p["mlp.2.bias"]), gates, "mlp", 1)
    mu, log_std = F.linear(h, p["mlp.4.weight"], p["mlp.4.bias"]).chunk(2, dim=-1)
    lo, hi = hp["actor_log_std_min"], hp["actor_log_std_max"]
    log_std = lo + 0.5 * (hi - lo) * (torch.tanh(log_std) + 1)
    if noise is None:
        return torch.tanh(mu), None, None, log_std
    pi = mu + noise * log_std.exp()
    log_pi = (-0.5 * noise.pow(2) - log_std).sum(-1, keepdim=True) - 0.5 * np.log(2 * np.pi) * noise.size(-1)
    mu, pi = torch.tanh(mu), torch.tanh(pi)
    log_pi = log_pi - torch.log(F.relu(1 - pi.pow(2)) + 1e-6).sum(-1, keepdim=True)
    return mu, pi, log_pi, log_std


def q_forward(p, q, h, a, gates=None):
    z = torch.cat([h, a], dim=1)
    z = _relu(F.linear(z, p[f"{q}.trunk.0.weight"], p[f"{q}.trunk.0.bias"]), gates, q, 0)
    z = _relu(F.linear(z, p[f"{q}.trunk.2.weight"], p[f"{q}.trunk.2.bias"]), gates, q, 1)
    return F.linear(z, p[f"{q}.trunk.4.weight"], p[f"{q}.trunk.4.bias"])


def critic_forward(p, x, a, hp, detach=False, gates=None):
    h = encode(p, x, hp, detach, gates)
    return q_forward(p, "Q1", h, a, gates), q_forward(p, "Q2", h, a, gates)


# ------------------------------------------------------------------ Adam (torch.optim.Adam, no amsgrad / wd)
class Adam:
    """One instance per reference optimizer; params is an ORDERED list of leaf tensors."""

    def __init__(self, params, lr, beta1, beta2=0.999, eps=1e-8):
        self.params, self.lr, self.b1, self.b2, self.eps = list(params), lr, beta1, beta2, eps
        self.m = [torch.zeros_like(t) for t in self.params]
        self.v = [torch.zeros_like(t) for t in self.params]
        self.t = [0 for _ in self.params]

    def zero_grad(self):
        for t in self.params:
            t.grad = None

    def step(self):
        with torch.no_grad():
            for i, t in enumerate(self.params):
                if t.grad is None:  # e.g. the shared CNN inside actor.parameters() (sac.py:119-133)
                    continue
                g = t.grad
                self.t[i] += 1
                k = self.t[i]
                self.m[i].mul_(self.b1).add_(g, alpha=1 - self.b1)
                self.v[i].mul_(self.b2).addcmul_(g, g, value=1 - self.b2)
                bc1 = 1 - self.b1 ** k
                bc2 = 1 - self.b2 ** k
                denom = (self.v[i].sqrt() / math.sqrt(bc2)).add_(self.eps)
                t.addcdiv_(self.m[i], denom, value=-(self.lr / bc1))


def soft_update(src, dst, names, tau):
    """utils.soft_update_params: target <- tau*p + (1-tau)*target (src/utils.py:76-78)."""
    with torch.no_grad():
        for n in names:
            dst[n].copy_(tau * src[n] + (1 - tau) * dst[n])


# ------------------------------------------------------------------ agent
class OracleAgent:
    """algo in {sac, rad, non_naive_rad, svea, drq, augcl}.  All randomness is passed in:
    `aug(x)` callables for the augmentations and explicit actor noise tensors."""

    def __init__(self, algo, obs_shape, A, actor_state, critic_state, hp=None, dtype=torch.float32, device="cpu"):
        """dtype=torch.float64 turns the same restatement into the REFEREE of the large-batch parity tests
        (tests/test_referee_gpu.py): both this oracle in float32 and the HIP path are measured against it.  `device`
        only says where torch evaluates these plain ATen ops (the referee may borrow the GPU's float64 units to stay
        inside the test budget at B=1024); nothing of augcl_amd is involved either way."""
        self.algo = algo
        self.hp = dict(DEFAULTS, **(hp or {}))
        self.A = A
        self.dtype = dtype
        self.rec = {}  # gradients / intermediate values of the last update(), for the parity tests
        mk = lambda d: {k: v.to(device=device, dtype=dtype).clone().requires_grad_(True) for k, v in d.items()}  # noqa: E731
        self.critic = mk(critic_state)
        self.actor = mk({k: v for k, v in actor_state.items() if not is_cnn(k)})
        for k in self.critic:  # actor and critic share the SAME CNN tensors (sac.py:17-32)
            if is_cnn(k):
                self.actor[k] = self.critic[k]
        self.critic_target = {k: v.detach().clone() for k, v in self.critic.items()}  # deepcopy (sac.py:42)
        self.log_alpha = torch.tensor(np.log(self.hp["init_temperature"]), dtype=torch.float64, device=device,
                                      requires_grad=True)
        self.target_entropy = -float(A)
        hp_ = self.hp
        self.actor_names = list(actor_shapes(obs_shape, A, hp_).keys())  # nn.Module.parameters() order
        self.critic_names = list(critic_shapes(obs_shape, A, hp_).keys())
        self.actor_opt = Adam([self.actor[n] for n in self.actor_names], hp_["actor_lr"], hp_["actor_beta"])
        self.critic_opt = Adam([self.critic[n] for n in self.critic_names], hp_["critic_lr"], hp_["critic_beta"])
        self.alpha_opt = Adam([self.log_alpha], hp_["alpha_lr"], hp_["alpha_beta"])
        if algo == "augcl":
            # supported flow = curriculum hand-off (augcl.py:16-27): critic_weak is the previous agent's
            # critic (with ITS optimizer), the strong critic/actor/target start as copies of it.
            self.critic_weak = {k: v.detach().clone().requires_grad_(True) for k, v in self.critic.items()}
            self.critic_weak_opt = Adam([self.critic_weak[n] for n in self.critic_names], hp_["critic_lr"],
                                        hp_["critic_beta"])

    @property
    def alpha(self):
        return self.log_alpha.exp()

    # -- pieces -----------------------------------------------------------------------------
    def target_q(self, next_obs, reward, not_done, noise):
        with torch.no_grad():
            _, pa, log_pi, _ = actor_forward(self.actor, next_obs, self.hp, noise)
            q1, q2 = critic_forward(self.critic_target, next_obs, pa, self.hp)
            v = torch.min(q1, q2) - self.alpha.detach() * log_pi
            return reward + not_done * self.hp["discount"] * v

    @staticmethod
    def _mse2(q1, q2, y):
        return F.mse_loss(q1, y) + F.mse_loss(q2, y)

    def _critic_step(self, opt, params, loss, tag):
        opt.zero_grad()
        loss.backward()
        self.rec[tag] = {n: params[n].grad.detach().clone() for n in self.critic_names}
        opt.step()

    def update_actor_and_alpha(self, obs, noise, log, forced=None):
        critic = self.critic
        forced = forced or {}
        _, pi, log_pi, log_std = actor_forward(self.actor, obs, self.hp, noise, detach=True, gates=forced.get("actor"))
        q1, q2 = critic_forward(critic, obs, pi, self.hp, detach=True, gates=forced.get("actor_q"))
        actor_loss = (self.alpha.detach() * log_pi - torch.min(q1, q2)).mean()
        ent = 0.5 * log_std.shape[1] * (1.0 + np.log(2 * np.pi)) + log_std.sum(dim=-1)
        log["train_actor/loss"] = float(actor_loss.detach())
        log["train_actor/mean_entropy"] = float(ent.mean().detach())
        self.actor_opt.zero_grad()
        for t in critic.values():
            t.grad = None
        actor_loss.backward()
        self.rec["actor_grads"] = {n: self.actor[n].grad.detach().clone() for n in self.actor_names
                                   if self.actor[n].grad is not None}
        # actor_optimizer only owns actor.parameters(); grads that landed on critic-only tensors are dropped
        self.actor_opt.step()
        self.alpha_opt.zero_grad()
        alpha_loss = (self.alpha * (-log_pi - self.target_entropy).detach()).mean()
        log["train_alpha/loss"] = float(alpha_loss.detach())
        log["train_alpha/value"] = float(self.alpha.detach())
        alpha_loss.backward()
        self.rec["log_alpha_grad"] = self.log_alpha.grad.detach().clone()
        self.alpha_opt.step()

    def soft_update_target(self, src=None):
        src = self.critic if src is None else src
        q = [n for n in self.critic_names if n.startswith("Q1.") or n.startswith("Q2.")]
        e = [n for n in self.critic_names if n.startswith("encoder.")]
        soft_update(src, self.critic_target, q, self.hp["critic_tau"])
        soft_update(src, self.critic_target, e, self.hp["encoder_tau"])

    # -- update() ---------------------------------------------------------------------------
    def update(self, batch, step, noise_target, noise_actor=None, aug=None, shift=None, shift2=None, forced=None,
               after_critic=None):
        """batch = (obs, action, reward, next_obs, not_done) tensors of self.dtype.
        forced: ReLU gates of the gradient-carrying passes taken from outside (see `_relu`): {"critic": [one dict
        {"cnn": [...], "Q1": [...], "Q2": [...]} per critic pass, in order], "weak": [...] (AugCL's weak critic),
        "actor": {"mlp": [...]}, "actor_q": {"Q1": [...], "Q2": [...]}}.
        after_critic(agent): called between the critic step(s) and the actor step -- the parity tests use it to hand the
        referee the critic parameters of the implementation under test, so that the actor phase is compared from an
        identical state (Adam's FIRST step is lr*sign(g): an element whose gradient is ~0 lands 2*lr apart in two correct
        evaluations, which would otherwise leak into every quantity computed afterwards).
        aug(x): the RAD chain; shift(x)/shift2(x): random_shift with the caller's offsets (shift2 is DrQ's draw
        for next_obs).  DrQ with K or M > 1 (drq.py:15-65): shift / shift2 are LISTS of K / M callables and
        noise_target a list of M noise tensors (one actor call per next_obs copy)."""
        obs, action, reward, next_obs, not_done = batch
        log = {}
        self.rec = {}
        algo = self.algo
        forced = forced or {}
        fc = lambda key, i: (forced[key][i] if key in forced else None)  # noqa: E731
        if algo in ("sac", "rad", "non_naive_rad"):
            if algo in ("rad", "non_naive_rad"):
                obs = aug(obs)
            if algo == "rad":
                next_obs = aug(next_obs)
            y = self.target_q(next_obs, reward, not_done, noise_target)
            q1, q2 = critic_forward(self.critic, obs, action, self.hp, gates=fc("critic", 0))
            loss = self._mse2(q1, q2, y)
            log["train_critic/loss"] = float(loss.detach())
            self._critic_step(self.critic_opt, self.critic, loss, "critic_grads")
            actor_obs = obs
        elif algo == "svea":
            obs = shift(obs)
            y = self.target_q(next_obs, reward, not_done, noise_target)
            q1, q2 = critic_forward(self.critic, obs, action, self.hp, gates=fc("critic", 0))
            l_shift = self.hp["svea_alpha"] * self._mse2(q1, q2, y)
            obs_aug = aug(obs.clone())
            q1a, q2a = critic_forward(self.critic, obs_aug, action, self.hp, gates=fc("critic", 1))
            l_aug = self.hp["svea_beta"] * self._mse2(q1a, q2a, y)
            loss = l_shift + l_aug
            log.update({"train_critic/shift_loss": float(l_shift.detach()), "train_critic/aug_loss": float(l_aug.detach()),
                        "train_critic/loss": float(loss.detach())})
            self._critic_step(self.critic_opt, self.critic, loss, "critic_grads")
            actor_obs = obs
        elif algo == "drq":
            K, M = self.hp["drq_k"], self.hp["drq_m"]
            shifts = list(shift) if isinstance(shift, (list, tuple)) else [shift]
            shifts2 = list(shift2) if isinstance(shift2, (list, tuple)) else [shift2]
            noises = list(noise_target) if isinstance(noise_target, (list, tuple)) else [noise_target]
            assert len(shifts) == K and len(shifts2) == M and len(noises) == M
            obs_list = [f(obs) for f in shifts]            # drq.py:69-70: all obs copies first, then next_obs copies
            next_list = [f(next_obs) for f in shifts2]
            y = 0
            running = []
            for no, nz in zip(next_list, noises):           # drq.py:23-30 (running SUM is what gets logged)
                y = y + self.target_q(no, reward, not_done, nz)
                running.append(y)
            y = y / M
            loss = 0
            q1s, q2s = [], []
            for ki, o in enumerate(obs_list):
                q1, q2 = critic_forward(self.critic, o, action, self.hp, gates=fc("critic", ki))
                loss = loss + self._mse2(q1, q2, y)
                q1s.append(q1.detach())
                q2s.append(q2.detach())
            var = lambda ts: float(torch.cat(ts, dim=-1).var(dim=-1, unbiased=False).mean()) if len(ts) > 1 else 0.0  # noqa: E731
            log["train_critic/loss"] = float(loss.detach())
            log[f"train_critic/target_Q_variance_m={M}"] = var(running)   # np.var: population variance (drq.py:36-51)
            log[f"train_critic/Q1_variance_k={K}"] = var(q1s)
            log[f"train_critic/Q2_variance_k={K}"] = var(q2s)
            self._critic_step(self.critic_opt, self.critic, loss, "critic_grads")
            actor_obs = obs_list[0]
        elif algo == "augcl":
            obs_shift = shift(obs)
            obs_aug = aug(obs)  # on the UN-shifted obs (augcl.py:72)
            y = self.target_q(next_obs, reward, not_done, noise_target)
            q1, q2 = critic_forward(self.critic_weak, obs_shift, action, self.hp, gates=fc("weak", 0))
            l_weak = self._mse2(q1, q2, y)
            q1, q2 = critic_forward(self.critic, obs_aug, action, self.hp, gates=fc("critic", 0))
            l_strong = self._mse2(q1, q2, y)
            log["train_critic_weak/loss"] = float(l_weak.detach())
            log["train_critic_strong/loss"] = float(l_strong.detach())
            self._critic_step(self.critic_weak_opt, self.critic_weak, l_weak, "weak_grads")
            self._critic_step(self.critic_opt, self.critic, l_strong, "critic_grads")
            actor_obs = obs_aug
        else:
            raise KeyError(algo)
        self.rec["target_Q"] = y.detach().clone()
        if after_critic is not None:
            after_critic(self)
        if step % self.hp["actor_update_freq"] == 0:
            self.update_actor_and_alpha(actor_obs, noise_actor, log, forced)
        if step % self.hp["critic_target_update_freq"] == 0:
            self.soft_update_target(self.critic_weak if algo == "augcl" else None)
        return log


def checksum(params, names):
    """Small per-parameter fingerprint (sum, abs-sum, first 4 values) for golden comparison."""
    out = []
    for n in names:
        t = params[n].detach().double().flatten()
        out.append([float(t.sum()), float(t.abs().sum())] + [float(v) for v in t[:4]] + [0.0] * max(0, 4 - t.numel()))
    return np.array(out, dtype=np.float64)
