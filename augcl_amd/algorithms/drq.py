"""DrQ (src/algorithms/drq.py:9-86): K shifted copies of obs, M of next_obs, target averaged over M,
critic loss summed over K (K = M = 1 by default, src/arguments.py:66-67)."""
import torch

from .. import augmentations
from .sac import SAC


class DrQ(SAC):
    def __init__(self, obs_shape, action_shape, args):
        super().__init__(obs_shape, action_shape, args)
        self.k = args.drq_k
        self.m = args.drq_m

    def update_critic(self, obs_list, action, reward, next_obs_list, not_done, L=None, step=None):
        target_Q = 0
        targets = []
        with torch.no_grad():
            for next_obs in next_obs_list:
                target_Q = target_Q + self.calculate_target_Q(next_obs, reward, not_done)
                targets.append(target_Q)  # running sum, as the reference logs it (drq.py:29-30)
        target_Q = target_Q / self.m
        critic_loss = 0
        q1s, q2s = [], []
        self._cuts(self.critic)
        for obs in obs_list:
            q1, q2 = self.critic(obs, action)
            critic_loss = critic_loss + self._q_loss(q1, q2, target_Q)
            q1s.append(q1.detach())
            q2s.append(q2.detach())
        if L is not None:
            # variances stay on the device (the reference forces a D2H copy per list element, drq.py:30,38-39)
            var = lambda ts: torch.cat(ts, dim=-1).var(dim=-1, unbiased=False).mean() if len(ts) > 1 else 0  # noqa: E731
            self._log(L, "train_critic/loss", critic_loss, step)
            self._log(L, f"train_critic/target_Q_variance_m={self.m}", var(targets), step)
            self._log(L, f"train_critic/Q1_variance_k={self.k}", var(q1s), step)
            self._log(L, f"train_critic/Q2_variance_k={self.k}", var(q2s), step)
        self.critic_optimizer.backward_step(critic_loss, self._cuts(self.critic, False))  # zero_grad / backward / step

    def _front_end(self, obs, action, reward, next_obs, not_done, out=None):
        """drq.py:69-74: K shifted copies of obs, M of next_obs"""
        obs_list = [augmentations.random_shift(obs) for _ in range(self.k)]
        next_obs_list = [augmentations.random_shift(next_obs) for _ in range(self.m)]
        return (action, reward, not_done, *obs_list, *next_obs_list)

    def _learn(self, t, L, step, do_actor, do_target):
        action, reward, not_done = t[:3]
        obs_list, next_obs_list = list(t[3:3 + self.k]), list(t[3 + self.k:])
        self.update_critic(obs_list=obs_list, action=action, reward=reward, next_obs_list=next_obs_list,
                           not_done=not_done, L=L, step=step)
        if do_actor:
            self.update_actor_and_alpha(obs_list[0], L, step)
        if do_target:
            self.soft_update_critic_target()
