"""SVEA (src/algorithms/svea.py:9-76): random_shift on obs, then two critic passes -- the shifted
batch and its strongly augmented copy -- against one un-augmented target, weighted alpha / beta."""
import inspect

import torch

from .. import augmentations, head_ops
from .rad import RAD


def _accepts(fn, name):
    key = (getattr(fn, "__func__", fn), name)
    hit = _accepts.cache.get(key)
    if hit is None:
        try:
            params = inspect.signature(fn).parameters
            ok = name in params or any(p.kind is inspect.Parameter.VAR_KEYWORD for p in params.values())
        except (TypeError, ValueError):
            ok = False
        _accepts.cache[key] = hit = (name, ok)
    return hit[1]


_accepts.cache = {}


class SVEA(RAD):
    def __init__(self, obs_shape, action_shape, args):
        super().__init__(obs_shape, action_shape, args)
        self.svea_alpha = args.svea_alpha
        self.svea_beta = args.svea_beta

    def update_critic(self, obs, action, reward, next_obs, not_done, L=None, step=None, obs_aug=None, obs_both=None):
        """svea.py:15-64.  The reference runs the critic twice (shifted batch, augmented batch) and adds the two
        weighted losses; here both halves go through the critic as ONE 2B batch -- every per-sample quantity is
        unchanged (no batch statistics anywhere on this path: LayerNorm is per sample), each half's MSE mean is taken
        over its own B rows, and the network sees half as many, larger launches."""
        target_Q = self.calculate_target_Q(next_obs=next_obs, reward=reward, not_done=not_done)
        B = obs.shape[0]
        if obs_both is None:
            if obs_aug is None:
                # apply_aug never writes its input except color_jitter (in place, like the reference's):
                # only then is the reference's obs.clone() needed to keep `obs` intact for the actor update
                obs_aug = self.apply_aug(obs.clone() if self._aug_mutates() else obs)
            obs_both = torch.cat([obs, obs_aug], dim=0)
        self._cuts(self.critic)
        q1, q2 = self.critic(obs_both, action)  # B action rows for the 2B observations: the critic repeats them
        if head_ops._ok(q1, q2, target_Q):
            # one launch: per-half mse(Q1,y) + mse(Q2,y) and their weighted sum (and one launch backward)
            critic_loss, segs = head_ops.q_loss(q1, q2, target_Q, (self.svea_alpha, self.svea_beta))
            critic_loss_shift, critic_loss_aug = (self.svea_alpha * segs[0], self.svea_beta * segs[1]) if L is not None else (None, None)
        else:
            y2 = torch.cat([target_Q, target_Q], dim=0)
            se = (q1 - y2).square() + (q2 - y2).square()  # per-sample (Q1-y)^2 + (Q2-y)^2
            critic_loss_shift = self.svea_alpha * se[:B].mean()  # = alpha * (mse(Q1,y) + mse(Q2,y)) on the shifted half
            critic_loss_aug = self.svea_beta * se[B:].mean()
            critic_loss = critic_loss_shift + critic_loss_aug
        if L is not None:
            self._log(L, "train_critic/shift_loss", critic_loss_shift, step)
            self._log(L, "train_critic/aug_loss", critic_loss_aug, step)
            self._log(L, "train_critic/loss", critic_loss, step)
        self.critic_optimizer.backward_step(critic_loss, self._cuts(self.critic, False))  # zero_grad / backward / step

    def _sample(self, replay_buffer):
        """With the device-resident replay buffer and the fused front end, sampling and augmentation are ONE pass over
        the uint8 frame ring (ReplayBuffer.sample_shift_conv): the first tensor of the batch is then already the
        (2B,C,H,W) stack `_front_end` would build."""
        fused = getattr(replay_buffer, "sample_shift_conv", None)
        # the fused kernel exists for square 84 / 100 frames (augcl_replay_shift_conv_u8_f32); any other ring geometry takes
        # sample() + the un-fused front end, as it did before the fused path existed
        hw = (getattr(replay_buffer, "h", None), getattr(replay_buffer, "w", None))
        if fused is None or not self._fused_front_end(None, hw=hw):
            return replay_buffer.sample()
        # next_obs is never augmented here (svea.py:66-68): it can stay uint8 all the way into the target pass's first conv layer
        u8 = torch.uint8 if (self._u8_next_obs() and hw == (84, 84) and _accepts(fused, "next_dtype")) else None
        static = self._graph["static"] if self._graph is not None and self._graph["seen"] >= self._graph["warmup"] else None
        if static is not None and static[0].shape[0] != 2 * replay_buffer.batch_size:
            static = None  # another batch size: the graphs are about to be re-captured
        if static is not None and (static[3].dtype == torch.uint8) != (u8 is not None):
            static = None  # the switch was flipped after the capture: hand over fresh tensors, _graph_update copies them in
        if static is None:
            return fused(next_dtype=u8) if u8 is not None else fused()
        # the graph's static obs_both / next_obs tensors are written in place: nothing to copy before the replay.  Whether
        # the buffer's sample_shift_conv takes out_next is read off its signature ONCE -- catching TypeError around the
        # call would also swallow one raised inside it, after the index draw had advanced the RNG streams
        if _accepts(fused, "out_next"):
            return fused(out=static[0], out_next=static[3])
        return fused(out=static[0])

    def _u8_next_obs(self):
        """uint8 next_obs from the replay ring straight into the target pass (precision "f32", no head conv layers: the no-grad
        stack's first layer reads uint8 observations, ABI v5).  `u8_next_obs = False` on the agent switches it off."""
        from .. import encoder_hip, encoder_ops
        cnn = self.critic_target.encoder.shared_cnn
        key = (cnn.precision, encoder_ops.PRECISION, encoder_hip._U8_FIRST, getattr(self, "u8_next_obs", True))
        hit = self.__dict__.get("_u8_next_hit")
        if hit is None or hit[0] != key:  # (asked once per update(): the answer only moves with the four switches above)
            heads = len(list(self.critic_target.encoder.head_cnn.parameters())) + len(list(self.actor.encoder.head_cnn.parameters()))
            ok = bool(key[3]) and key[2] and (key[0] or key[1]) == "f32" and heads == 0 and cnn.num_layers >= 2
            self._u8_next_hit = hit = (key, ok)
        return hit[1]

    def _front_end(self, obs, action, reward, next_obs, not_done, out=None):
        """svea.py:66-68.  Returns (obs_both | obs, action, reward, next_obs, not_done): with the fused front end
        the first tensor is the (2B,C,H,W) stack [random_shift(obs); random_conv(random_shift(obs))]."""
        if obs.shape[0] == 2 * action.shape[0]:  # already [shifted; shifted + conv] straight from the replay ring
            return obs, action, reward, next_obs, not_done
        if self._fused_front_end(obs):
            # one pass over the (uint8 or f32) batch: random_shift AND random_conv(random_shift) together, written
            # as the two halves of one (2B,C,H,W) buffer so the critic consumes them without a concatenation copy
            first = augmentations.shift_conv(obs, stacked=True, out=None if out is None else out[0])
        else:
            first = augmentations.random_shift(obs, out=None if out is None else out[0])
        return first, action, reward, next_obs, not_done

    def _learn(self, t, L, step, do_actor, do_target):
        first, action, reward, next_obs, not_done = t
        B = action.shape[0]
        obs_both = first if first.shape[0] == 2 * B else None
        obs = first[:B]
        self.update_critic(obs, action, reward, next_obs, not_done, L, step, obs_both=obs_both)
        if do_actor:
            self.update_actor_and_alpha(obs, L, step)
        if do_target:
            self.soft_update_critic_target()

    def _fused_front_end(self, obs, hw=None):
        a = self.aug_funcs
        if hw is None and obs is not None:
            hw = (obs.shape[-2], obs.shape[-1])
        return (list(a) == ["conv"] and a["conv"]["func"] is augmentations.random_conv and not a["conv"]["params"]
                and hw is not None and hw[1] in (84, 100) and hw[0] == hw[1]
                and getattr(self, "fuse_front_end", True))
