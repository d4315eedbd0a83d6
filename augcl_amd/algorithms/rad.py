"""RAD: augment obs and next_obs with a `--data_aug a-b-c` chain (src/algorithms/rad.py:7-36)."""
import json

from .. import augmentations
from .sac import SAC


class RAD(SAC):
    def __init__(self, obs_shape, action_shape, args):
        super().__init__(obs_shape, action_shape, args)
        aug_params = json.loads(args.aug_params) if args.aug_params else {}
        self.aug_funcs = dict()
        for key in args.data_aug.split("-"):
            if key not in augmentations.aug_to_func:
                raise KeyError(f"augmentation {key!r} is not on the accelerated path; have "
                               f"{sorted(augmentations.aug_to_func)}")
            self.aug_funcs[key] = dict(func=augmentations.aug_to_func[key], params=aug_params.get(key, {}))

    def apply_aug(self, x):
        for aug in self.aug_funcs.values():
            x = aug["func"](x, **aug["params"])
        return x

    def _aug_mutates(self):
        return any(k == "color_jitter" for k in self.aug_funcs)

    def _front_end(self, obs, action, reward, next_obs, not_done, out=None):
        """rad.py:31-35: the augmentation chain on obs and on next_obs"""
        return self.apply_aug(obs), action, reward, self.apply_aug(next_obs), not_done
