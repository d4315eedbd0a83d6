"""Non-Naive RAD: only obs is augmented, the target sees the clean next_obs
(src/algorithms/non_naive_rad.py:8-18).  `--data_aug shift` is BASELINE config 1."""
from .rad import RAD


class NonNaiveRAD(RAD):
    def update(self, replay_buffer, L, step):
        obs, action, reward, next_obs, not_done = replay_buffer.sample()
        obs = self.apply_aug(obs)
        self.update_critic(obs, action, reward, next_obs, not_done, L, step)
        if step % self.actor_update_freq == 0:
            self.update_actor_and_alpha(obs, L, step)
        if step % self.critic_target_update_freq == 0:
            self.soft_update_critic_target()
