"""Non-Naive RAD: only obs is augmented, the target sees the clean next_obs
(src/algorithms/non_naive_rad.py:8-18).  `--data_aug shift` is BASELINE config 1."""
from .rad import RAD


class NonNaiveRAD(RAD):
    def _front_end(self, obs, action, reward, next_obs, not_done, out=None):
        """non_naive_rad.py:9-12: the augmentation chain on obs only"""
        return self.apply_aug(obs), action, reward, next_obs, not_done
