"""AugCL (src/algorithms/augcl.py:9-89): a weak critic trained on shifted obs and a strong critic
trained on strongly augmented obs share one target, which tracks the WEAK critic; the actor
learns from the strong critic.  Entered through the curriculum hand-off `load_pretrained_agent`."""
from copy import deepcopy

from .. import augmentations, flat
from ..augmentations import random_shift  # noqa: F401  (imported by name in the reference, augcl.py:3)
from .rad import RAD
from .sac import SAC


class AugCL(RAD):
    def __init__(self, obs_shape, action_shape, args):
        super().__init__(obs_shape, action_shape, args)
        # as in the reference (augcl.py:13-14) the pre-hand-off weak critic is a detached copy whose
        # optimiser owns orphan tensors; it only becomes meaningful after load_pretrained_agent()
        self.critic_weak = deepcopy(self.critic)
        self._weak_group = flat.FlatGroup(self.critic_weak.named_parameters())
        self.critic_weak_optimizer = flat.FlatAdam(flat.FlatGroup(deepcopy(self.critic).named_parameters()),
                                                   self._hp["critic_lr"], (self._hp["critic_beta"], 0.999),
                                                   reducer=self._reducer)

    def load_pretrained_agent(self, pretrained_agent: SAC):
        """augcl.py:16-27: copy actor / critic / target values, then ALIAS the previous agent's critic and
        its optimiser (state included) as the weak critic."""
        for src, dst in ((pretrained_agent.actor_group, self.actor_group),
                         (pretrained_agent.critic_group, self.critic_group),
                         (pretrained_agent.target_group, self.target_group)):
            assert src.names == dst.names
            dst.param.copy_(src.param)
        self.critic_weak = pretrained_agent.critic
        self._weak_group = pretrained_agent.critic_group
        self.critic_weak_optimizer = pretrained_agent.critic_optimizer

    def update_critic(self, obs_aug, obs_shift, action, reward, next_obs, not_done, L=None, step=None):
        target_Q = self.calculate_target_Q(next_obs=next_obs, reward=reward, not_done=not_done)
        q1, q2 = self.critic_weak(obs_shift, action)
        critic_loss_shift = self._q_loss(q1, q2, target_Q)
        q1, q2 = self.critic(obs_aug, action)
        critic_loss_aug = self._q_loss(q1, q2, target_Q)
        if L is not None:
            self._log(L, "train_critic_weak/loss", critic_loss_shift, step)
            self._log(L, "train_critic_strong/loss", critic_loss_aug, step)
        self.critic_weak_optimizer.backward_step(critic_loss_shift)  # zero_grad / backward / step (augcl.py:50-52)
        self.critic_optimizer.backward_step(critic_loss_aug)         # (augcl.py:54-56)

    def soft_update_critic_target(self):
        self._soft_update_from(self._weak_group)  # the target tracks the weak critic (augcl.py:58-67)

    def update(self, replay_buffer, L, step):
        obs, action, reward, next_obs, not_done = replay_buffer.sample()
        obs_shift = augmentations.random_shift(obs)
        obs_aug = self.apply_aug(obs.clone() if self._aug_mutates() else obs)  # on the UN-shifted obs (augcl.py:72)
        self.update_critic(obs_shift=obs_shift, obs_aug=obs_aug, action=action, reward=reward, next_obs=next_obs,
                           not_done=not_done, L=L, step=step)
        if step % self.actor_update_freq == 0:
            self.update_actor_and_alpha(obs_aug, L, step)
        if step % self.critic_target_update_freq == 0:
            self.soft_update_critic_target()
