"""SVEA (src/algorithms/svea.py:9-76): random_shift on obs, then two critic passes -- the shifted
batch and its strongly augmented copy -- against one un-augmented target, weighted alpha / beta."""
import torch

from .. import augmentations
from .rad import RAD


class SVEA(RAD):
    def __init__(self, obs_shape, action_shape, args):
        super().__init__(obs_shape, action_shape, args)
        self.svea_alpha = args.svea_alpha
        self.svea_beta = args.svea_beta

    def update_critic(self, obs, action, reward, next_obs, not_done, L=None, step=None, obs_aug=None):
        target_Q = self.calculate_target_Q(next_obs=next_obs, reward=reward, not_done=not_done)
        current_Q1, current_Q2 = self.critic(obs, action)
        critic_loss_shift = self.svea_alpha * self._q_loss(current_Q1, current_Q2, target_Q)
        if obs_aug is None:
            # apply_aug never writes its input except color_jitter (in place, like the reference's):
            # only then is the reference's obs.clone() needed to keep `obs` intact for the actor update
            obs_aug = self.apply_aug(obs.clone() if self._aug_mutates() else obs)
        current_Q1_aug, current_Q2_aug = self.critic(obs_aug, action)
        critic_loss_aug = self.svea_beta * self._q_loss(current_Q1_aug, current_Q2_aug, target_Q)
        critic_loss = critic_loss_shift + critic_loss_aug
        if L is not None:
            self._log(L, "train_critic/shift_loss", critic_loss_shift, step)
            self._log(L, "train_critic/aug_loss", critic_loss_aug, step)
            self._log(L, "train_critic/loss", critic_loss, step)
        self.critic_optimizer.zero_grad()
        critic_loss.backward()
        self.critic_optimizer.step()

    def update(self, replay_buffer, L, step):
        obs, action, reward, next_obs, not_done = replay_buffer.sample()
        obs_aug = None
        if self._fused_front_end(obs):
            # one pass over the (uint8 or f32) batch: random_shift AND random_conv(random_shift) together
            obs, obs_aug = augmentations.shift_conv(obs)
        else:
            obs = augmentations.random_shift(obs)
        self.update_critic(obs, action, reward, next_obs, not_done, L, step, obs_aug=obs_aug)
        if step % self.actor_update_freq == 0:
            self.update_actor_and_alpha(obs, L, step)
        if step % self.critic_target_update_freq == 0:
            self.soft_update_critic_target()

    def _fused_front_end(self, obs):
        a = self.aug_funcs
        return (list(a) == ["conv"] and a["conv"]["func"] is augmentations.random_conv and not a["conv"]["params"]
                and obs.shape[-1] in (84, 100) and obs.shape[-2] == obs.shape[-1]
                and getattr(self, "fuse_front_end", True))
