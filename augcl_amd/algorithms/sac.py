"""SAC base agent (drop-in for src/algorithms/sac.py:9-164).

Same attribute surface as the reference (actor, critic, critic_target, log_alpha, alpha,
*_optimizer, train/eval, select_action/sample_action, update(replay_buffer, L, step)), different
engine room: parameters of each optimiser group live in one flat buffer (flat.FlatGroup), the
optimiser step and the target EMA are single fused HIP launches, the shared CNN runs as one HIP
autograd node, and gradients can be exchanged data-parallel through one all-reduce per group.
"""
from copy import deepcopy

import numpy as np
import torch
import torch.nn.functional as F

from .. import augmentations
from .. import dist as apdist
from .. import flat, head_ops
from . import modules as m


def _device():
    if not torch.cuda.is_available():
        raise RuntimeError("augcl_amd agents need an MI355X: there is no CPU path")
    return torch.device("cuda", torch.cuda.current_device())


class SAC(object):
    def __init__(self, obs_shape, action_shape, args):
        self.discount = args.discount
        self.critic_tau = args.critic_tau
        self.encoder_tau = args.encoder_tau
        self.actor_update_freq = args.actor_update_freq
        self.critic_target_update_freq = args.critic_target_update_freq
        dev = self.device = _device()

        shared_cnn = m.SharedCNN(obs_shape, args.num_shared_layers, args.num_filters)
        head_cnn = m.HeadCNN(shared_cnn.out_shape, args.num_head_layers, args.num_filters)
        actor_encoder = m.Encoder(shared_cnn, head_cnn, m.RLProjection(head_cnn.out_shape, args.projection_dim))
        critic_encoder = m.Encoder(shared_cnn, head_cnn, m.RLProjection(head_cnn.out_shape, args.projection_dim))
        self.actor = m.Actor(actor_encoder, action_shape, args.hidden_dim, args.actor_log_std_min,
                             args.actor_log_std_max).to(dev)
        self.critic = m.Critic(critic_encoder, action_shape, args.hidden_dim).to(dev)
        self.critic_target = deepcopy(self.critic)

        self.log_alpha = torch.tensor(np.log(args.init_temperature), device=dev)  # float64 scalar, as sac.py:44
        self.log_alpha.requires_grad = True
        self.target_entropy = -np.prod(action_shape)

        self._hp = dict(actor_lr=args.actor_lr, actor_beta=args.actor_beta, critic_lr=args.critic_lr,
                        critic_beta=args.critic_beta, alpha_lr=args.alpha_lr, alpha_beta=args.alpha_beta)
        self._build_groups()
        self._graph = None
        self._act = None
        self.train()
        self.critic_target.train()

    def __getstate__(self):
        """torch.save(agent) (src/train.py:229): captured graphs are process-local, the modes they belong to survive."""
        st = dict(self.__dict__)
        if st.get("_graph") is not None:
            st["_graph"] = dict(st["_graph"], graphs={}, static=None, pool=None, seen=0)
        if st.get("_act") is not None:
            st["_act"] = dict(x=None, graphs={})
        return st

    # ---- flat groups / optimisers ------------------------------------------------------------
    def _build_groups(self):
        hp = self._hp
        reducer = apdist.default_reducer()
        # critic group: CNN + critic projection + Q1 + Q2, in nn.Module.parameters() order
        self.critic_group = flat.FlatGroup(flat.critic_order(self.critic.named_parameters()))
        # actor group: what the actor owns exclusively.  The reference's actor optimiser also lists the shared
        # CNN tensors, but they never receive a gradient there (detach=True, sac.py:120) so it never moves them.
        shared = {id(p) for p in self.critic.parameters()}
        self.actor_group = flat.FlatGroup([(n, p) for n, p in self.actor.named_parameters() if id(p) not in shared])
        self.target_group = flat.FlatGroup(flat.critic_order(self.critic_target.named_parameters()))
        assert self.target_group.names == self.critic_group.names
        self.critic_optimizer = flat.FlatAdam(self.critic_group, hp["critic_lr"], (hp["critic_beta"], 0.999), reducer=reducer)
        self.actor_optimizer = flat.FlatAdam(self.actor_group, hp["actor_lr"], (hp["actor_beta"], 0.999), reducer=reducer)
        self.log_alpha_optimizer = flat.ScalarAdam(self.log_alpha, hp["alpha_lr"], (hp["alpha_beta"], 0.999))
        self._reducer = reducer
        for grp in (self.critic_group, self.target_group):
            grp.mark_cut(lambda n: n.startswith("encoder.shared_cnn.") or n.startswith("encoder.head_cnn."))
        if reducer is not None:  # replicas must start identical
            for g in (self.critic_group, self.actor_group, self.target_group):
                reducer.broadcast_(g.param)

    def train(self, training=True):
        self.training = training
        self.actor.train(training)
        self.critic.train(training)

    def eval(self):
        self.train(False)

    @property
    def alpha(self):
        return self.log_alpha.exp()

    # ---- acting ------------------------------------------------------------------------------
    def _obs_to_input(self, obs):
        _obs = np.array(obs) if not isinstance(obs, (np.ndarray, torch.Tensor)) else obs
        # upload in the observation's own dtype (uint8 frames: 63 KB), convert on the device: a host-side uint8 -> float32
        # pass plus a 254 KB pageable copy measured milliseconds per call
        _obs = torch.as_tensor(_obs).to(self.device).to(torch.float32)
        return _obs.unsqueeze(0).contiguous()

    def select_action(self, obs):
        """sac.py:73-78"""
        if self._act is not None:
            return self._act_replay(obs, sample=False)
        _obs = self._obs_to_input(obs)
        with torch.no_grad():
            mu, _, _, _ = self.actor(_obs, compute_pi=False, compute_log_pi=False)
        return mu.cpu().data.numpy().flatten()

    def sample_action(self, obs):
        """sac.py:87-92"""
        if self._act is not None:
            return self._act_replay(obs, sample=True)
        _obs = self._obs_to_input(obs)
        with torch.no_grad():
            mu, pi, _, _ = self.actor(_obs, compute_log_pi=False)
        return pi.cpu().data.numpy().flatten()

    def enable_action_graphs(self):
        """Batch-1 acting (one call per environment step, sac.py:73-92) as a captured hipGraph: the observation is copied
        into a static device tensor, ONE graph launch replays the conv stack, projection, policy MLP and head (~20
        kernels whose CPU launch cost dominates at batch 1), the action comes back through a pinned buffer.  The
        graphs read the live parameter buffers, so training in between needs no re-capture."""
        if self._act is None:
            self._act = dict(x=None, graphs={})

    def _act_replay(self, obs, sample):
        A = self._act
        src = torch.as_tensor(np.ascontiguousarray(np.asarray(obs)))
        if A["x"] is None or tuple(A["x"].shape[1:]) != tuple(src.shape) or A["x"].dtype != src.dtype:
            A.update(x=torch.zeros((1, *src.shape), dtype=src.dtype, device=self.device), graphs={})
        A["x"][0].copy_(src)  # one H2D copy in the observation's own dtype (the .float() is part of the graph)
        if sample not in A["graphs"]:
            def run():
                with torch.no_grad():
                    mu, pi, _, _ = self.actor(A["x"].to(torch.float32), compute_pi=sample, compute_log_pi=False)
                return pi if sample else mu
            for _ in range(2):  # eager warm-up: lazy initialisation (GEMM heuristics) must not happen inside a capture
                run()
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                out = run()
            A["graphs"][sample] = (g, out)
        g, out = A["graphs"][sample]
        g.replay()
        return out.cpu().numpy().flatten()

    # ---- learning ----------------------------------------------------------------------------
    def calculate_target_Q(self, next_obs, reward, not_done):
        """sac.py:94-101"""
        with torch.no_grad():
            fa = ft = None
            ea, et = self.actor.encoder, self.critic_target.encoder
            if len(list(ea.head_cnn.parameters())) == 0 and len(list(et.head_cnn.parameters())) == 0:
                # the actor's CNN and the target critic's CNN see the same next_obs: one launch per layer for both
                both = ea.shared_cnn.forward_with(et.shared_cnn, next_obs)
                if both is not None:
                    fa, ft = ea.head_cnn(both[0]), et.head_cnn(both[1])
            _, policy_action, log_pi, _ = self.actor(next_obs, features=fa)
            target_Q1, target_Q2 = self.critic_target(next_obs, policy_action, features=ft)
            if head_ops._ok(target_Q1, log_pi, reward, not_done):
                return head_ops.td_target(target_Q1, target_Q2, log_pi, reward, not_done, None, self.discount,
                                          log_alpha=self.log_alpha)
            target_V = torch.min(target_Q1, target_Q2) - self.alpha.detach() * log_pi
            return reward + (not_done * self.discount * target_V)

    @staticmethod
    def _q_loss(q1, q2, target_Q):
        if head_ops._ok(q1, q2, target_Q) and not target_Q.requires_grad:
            return head_ops.q_loss(q1, q2, target_Q)[0]
        return F.mse_loss(q1, target_Q) + F.mse_loss(q2, target_Q)

    # ---- logged scalars: ONE device tensor, one collective, one host sync per update() ------------------------------------------
    # The reference hands every scalar to L.log as a device tensor and its Logger calls .item() on each (src/logger.py:138-141):
    # 3-7 device syncs per update.  Here update() COLLECTS what its sub-steps log (same keys, same values, same order),
    # packs the values into one float64 vector on the device (one concatenation launch), averages that vector over the ranks with
    # ONE all-reduce under data parallelism, copies it to the host ONCE and then calls L.log(key, float, step).  Sub-steps called
    # on their own (update_critic(..., L, step) from a test) log immediately, as before.
    _log_items = None

    def _log(self, L, key, value, step):
        if L is None:
            return
        if self._log_items is not None:  # inside update(): collected, emitted together at its end
            self._log_items.append((key, value))
            return
        if self._reducer is not None and torch.is_tensor(value):
            value = self._reducer.mean_scalar(value)
        L.log(key, value, step)

    def _pack_logs(self):
        """(keys, float64 device vector of the collected TENSOR values, averaged over the ranks) -- part of a captured graph in
        hipGraph mode.  A key whose value is a python number (DrQ's variances at K = M = 1: the constant 0) carries it in `keys`
        as (name, number): no host-to-device copy, which a capturing stream would refuse.  float32 -> float64 is exact."""
        items, self._log_items = self._log_items, None
        if not items:
            return (), None
        keys = tuple(k if torch.is_tensor(v) else (k, float(v)) for k, v in items)
        vals = [v.detach().reshape(1) for _, v in items if torch.is_tensor(v)]
        if not vals:
            return keys, None
        if all(v.dtype == torch.float32 for v in vals):
            vec = torch.cat(vals).to(torch.float64)           # one concatenation launch + one widening
        else:                                                 # (the temperature's loss / value are float64 like log_alpha)
            vec = torch.cat([v.to(torch.float64) for v in vals])
        if self._reducer is not None and self._reducer.world > 1:
            self._reducer.allreduce_mean_(vec)                # ONE collective for all of the update's scalars
        return keys, vec

    @staticmethod
    def _emit_logs(L, keys, vec, step):
        if L is None or not keys:
            return
        host = iter(vec.tolist()) if vec is not None else iter(())  # ONE device-to-host copy (and sync) per update()
        for k in keys:
            if isinstance(k, tuple):
                L.log(k[0], k[1], step)
            else:
                L.log(k, next(host), step)

    def _cuts(self, critic, begin=True):
        """Data parallel only: collect (begin) / hand over the CNN features of a critic's gradient-carrying passes, the
        cut that lets the heads' gradient exchange overlap the conv stack's backward (flat.FlatAdam.backward_begin)."""
        enc = critic.encoder
        if begin:
            enc.cuts = [] if (self._reducer is not None and self._reducer.world > 1) else None
            return None
        cuts, enc.cuts = enc.cuts, None
        return cuts

    def update_critic(self, obs, action, reward, next_obs, not_done, L=None, step=None):
        """sac.py:103-117"""
        target_Q = self.calculate_target_Q(next_obs=next_obs, reward=reward, not_done=not_done)
        self._cuts(self.critic)
        current_Q1, current_Q2 = self.critic(obs, action)
        critic_loss = self._q_loss(current_Q1, current_Q2, target_Q)
        self._log(L, "train_critic/loss", critic_loss, step)
        self.critic_optimizer.backward_step(critic_loss, self._cuts(self.critic, False))  # zero_grad / backward / step

    def update_actor_and_alpha(self, obs, L=None, step=None, update_alpha=True, critic=None):
        """sac.py:119-146.  The actor's and the critic's encoders share the CNN and see the same obs
        with detach=True, so the CNN runs once (the reference runs it twice to the same values)."""
        critic = self.critic if critic is None else critic
        with torch.no_grad():
            feat = self.actor.encoder.features(obs)
        same_cnn = critic.encoder.shared_cnn is self.actor.encoder.shared_cnn
        _, pi, log_pi, log_std = self.actor(obs, detach=True, features=feat)
        actor_Q1, actor_Q2 = critic(obs, pi, detach=True, features=feat if same_cnn else None, freeze_q=True)
        fused = head_ops._ok(log_pi, actor_Q1, actor_Q2)
        if fused:  # alpha * log_pi - min(Q1, Q2), mean: one launch forward, one backward
            actor_loss = head_ops.actor_loss(log_pi, actor_Q1, actor_Q2, self.log_alpha)
        else:
            actor_loss = (self.alpha.detach() * log_pi - torch.min(actor_Q1, actor_Q2)).mean()
        if L is not None:
            entropy = 0.5 * log_std.shape[1] * (1.0 + np.log(2 * np.pi)) + log_std.sum(dim=-1)
            self._log(L, "train_actor/loss", actor_loss, step)
            self._log(L, "train_actor/mean_entropy", entropy.mean(), step)
        # zero_grad / backward / step (sac.py:133-135); the critic's own weight gradients, which the reference
        # computes here and discards at the next critic zero_grad, are never built
        self.actor_optimizer.backward_step(actor_loss)
        if update_alpha:
            if fused:  # temperature loss, its gradient and Adam's step on the float64 scalar: one launch (sac.py:137-146)
                alpha_loss, alpha_before = self.log_alpha_optimizer.fused_step(log_pi, self.target_entropy, self._reducer)
                if L is not None:
                    self._log(L, "train_alpha/loss", alpha_loss.clone(), step)
                    self._log(L, "train_alpha/value", alpha_before.clone(), step)
                return
            self.log_alpha_optimizer.zero_grad()
            alpha_loss = (self.alpha * (-log_pi - self.target_entropy).detach()).mean()
            self._log(L, "train_alpha/loss", alpha_loss, step)
            self._log(L, "train_alpha/value", self.alpha, step)
            alpha_loss.backward()
            if self._reducer is not None:
                self._reducer.allreduce_mean_(self.log_alpha.grad)
            self.log_alpha_optimizer.step()

    def _soft_update_from(self, src_group):
        """sac.py:148-153: Q1,Q2 with critic_tau, encoder (CNN + projection) with encoder_tau -- two fused
        segments of the flat buffers in ONE launch instead of ~40 per-tensor updates."""
        tg = self.target_group
        ea, eb = tg.segment("encoder.")
        qa, qb = tg.segment("Q")
        flat.ema2_(tg.param, src_group.param, (ea, eb, self.encoder_tau), (qa, qb, self.critic_tau))

    def soft_update_critic_target(self):
        self._soft_update_from(self.critic_group)

    # ---- update() = front end (sampling + augmentation, eager) + learn (losses / backward / optimisers) ------
    def _front_end(self, obs, action, reward, next_obs, not_done, out=None):
        """What update() does to the sampled batch before any network runs.  Returns the tuple of tensors
        `_learn` consumes; `out` (graph mode) = the static tensors a previous call returned, to be refilled."""
        return obs, action, reward, next_obs, not_done

    def _learn(self, t, L, step, do_actor, do_target):
        obs, action, reward, next_obs, not_done = t
        self.update_critic(obs, action, reward, next_obs, not_done, L, step)
        if do_actor:
            self.update_actor_and_alpha(obs, L, step)
        if do_target:
            self.soft_update_critic_target()

    def _sample(self, replay_buffer):
        return replay_buffer.sample()

    def update(self, replay_buffer, L, step):
        """sac.py:155-164"""
        if self._reducer is not None:  # data parallel: augmentation parameters are keyed by the GLOBAL sample index
            augmentations.set_sample_base(apdist.sample_base(self._reducer, getattr(replay_buffer, "batch_size", 0)))
        batch = self._sample(replay_buffer)
        if self._reducer is not None:
            augmentations.set_sample_base(apdist.sample_base(self._reducer, batch[1].shape[0]))
        do_actor = step % self.actor_update_freq == 0
        do_target = step % self.critic_target_update_freq == 0
        if self._graph is not None:
            return self._graph_update(batch, L, step, do_actor, do_target)
        keys, vec = self._learn_logged(self._front_end(*batch), L, step, do_actor, do_target)
        self._emit_logs(L, keys, vec, step)

    def _learn_logged(self, t, L, step, do_actor, do_target):
        """_learn with its logged scalars collected and packed: returns (keys, device vector) -- ((), None) without a logger"""
        self._log_items = [] if L is not None else None
        try:
            self._learn(t, L, step, do_actor, do_target)
        except BaseException:
            self._log_items = None
            raise
        return self._pack_logs() if L is not None else ((), None)

    # ---- hipGraph mode ---------------------------------------------------------------------------------------
    def _optimizers(self):
        return [self.critic_optimizer, self.actor_optimizer]

    def _cnn_modules(self):
        from .modules import HeadCNN, SharedCNN
        nets = [getattr(self, n, None) for n in ("actor", "critic", "critic_target", "critic_weak")]
        return [m for net in nets if net is not None for m in net.modules() if isinstance(m, (SharedCNN, HeadCNN))]

    def set_precision(self, name):
        """Arithmetic of THIS agent's conv stacks (not in the reference): "f32" = parity path (f16 pairs), "f32_x3" = the same
        tensors as range-free bf16 triples, "f32_mfma" = the same tensors on the native f32 MFMA pipe, "bf16" = performance
        mode, None = follow encoder_ops' process default.  Captured hipGraphs bake the kernels in, so they are dropped and
        recaptured."""
        from .. import encoder_ops
        if name is not None:
            encoder_ops._check(name)
        for m in self._cnn_modules():
            m.precision = name
        if getattr(self, "_graph", None) is not None:
            self._graph["graphs"].clear()
        if getattr(self, "_act", None) is not None:  # the acting graphs replay the old precision's kernels as well
            self._act["graphs"] = {}
        return self

    def enable_hip_graphs(self, warmup=3):
        """Capture `_learn` (every network pass, backward, the all-reduce, Adam and the target EMA) in hipGraphs --
        one per (actor step?, target step?, logging?) variant -- and replay them: one launch per update() instead
        of ~300.  The front end (replay sampling, augmentation with its per-call Philox offsets) stays eager and
        writes into the graphs' static input tensors.  The first `warmup` updates run eagerly (lazy workspaces,
        hipBLASLt heuristics).  Step counts live on the device (FlatAdam.make_capturable, Adam(capturable=True))."""
        if self._graph is not None:
            return
        for opt in self._optimizers():
            opt.make_capturable()
        self._graph = dict(warmup=int(warmup), seen=0, graphs={}, static=None, pool=None, shapes=None)

    def _graph_update(self, batch, L, step, do_actor, do_target):
        G = self._graph
        if G["seen"] < G["warmup"]:
            G["seen"] += 1
            keys, vec = self._learn_logged(self._front_end(*batch), L, step, do_actor, do_target)
            return self._emit_logs(L, keys, vec, step)
        if G["static"] is not None and tuple(batch[1].shape) != G["shapes"]:
            G.update(graphs={}, static=None, pool=None)  # another batch size: drop the graphs (and their memory pool)
        t = self._front_end(*batch, out=G["static"])
        G["shapes"] = tuple(batch[1].shape)
        if G["static"] is None:
            G["static"] = tuple(x.clone() if torch.is_tensor(x) else x for x in t)
        pairs = [(dst, src) for dst, src in zip(G["static"], t) if torch.is_tensor(src) and dst.data_ptr() != src.data_ptr()]
        small = [(d, s_) for d, s_ in pairs if d.numel() <= (1 << 18)]
        if len(small) > 1:  # refill the graphs' static inputs: the (B, 1)-sized ones in one multi-tensor launch
            torch._foreach_copy_([d for d, _ in small], [s_ for _, s_ in small])
            pairs = [p for p in pairs if p[0].numel() > (1 << 18)]
        for d, s_ in pairs:
            d.copy_(s_)
        key = (bool(do_actor), bool(do_target), L is not None)
        if key not in G["graphs"]:
            g = torch.cuda.CUDAGraph()
            torch.cuda.synchronize()
            with torch.cuda.graph(g, pool=G["pool"]):  # the packed log vector (and its all-reduce) is part of the graph
                packed = self._learn_logged(G["static"], _CAPTURE_LOGGER if L is not None else None, step, do_actor, do_target)
            if G["pool"] is None:
                G["pool"] = g.pool()
            G["graphs"][key] = (g, packed)
        g, (keys, vec) = G["graphs"][key]
        g.replay()
        self._emit_logs(L, keys, vec, step)


class _CaptureLogger:
    """Stands in for the Logger while a graph is captured (the scalars are collected by the agent, nothing is logged here)."""

    def log(self, key, value, step, n=1):
        raise RuntimeError("nothing logs directly inside a captured update()")


_CAPTURE_LOGGER = _CaptureLogger()
