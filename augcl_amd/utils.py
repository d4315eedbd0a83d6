"""Host-side helpers the hot path needs from the reference's src/utils.py (same names)."""
import torch

from . import _lib


class eval_mode(object):
    """src/utils.py:60-73"""

    def __init__(self, *models):
        self.models = models

    def __enter__(self):
        self.prev_states = []
        for model in self.models:
            self.prev_states.append(model.training)
            model.train(False)

    def __exit__(self, *args):
        for model, state in zip(self.models, self.prev_states):
            model.train(state)
        return False


def soft_update_params(net, target_net, tau):
    """src/utils.py:76-78 for arbitrary module pairs: one fused EMA launch per parameter tensor.
    (Agents built by this package use flat.ema_ on whole segments instead: 2 launches total.)"""
    for param, target_param in zip(net.parameters(), target_net.parameters()):
        if tau == 1:
            target_param.data.copy_(param.data)
        else:
            _lib.call("augcl_ema_f32", target_param.data.data_ptr(), param.data.data_ptr(), param.numel(), float(tau),
                      _lib.current_stream(param.device))


class dotdict(dict):
    __getattr__ = dict.__getitem__
    __setattr__ = dict.__setitem__
    __delattr__ = dict.__delitem__


class LazyFrames(object):
    """Minimal stand-in with the array protocol of src/utils.py:283-322 (select_action input)."""

    def __init__(self, frames):
        self._frames = frames

    def __array__(self, dtype=None, copy=None):
        import numpy as np
        out = np.concatenate(self._frames, axis=0)
        return out.astype(dtype) if dtype is not None else out
