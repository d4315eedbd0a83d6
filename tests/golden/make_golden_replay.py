#!/usr/bin/env python3
"""Golden vectors for the replay path from the UNMODIFIED reference ReplayBuffer / LazyFrames
(src/utils.py:136-322).  Run: PYTHONDONTWRITEBYTECODE=1 python3 -O tests/golden/make_golden_replay.py
A seeded episodic stream (tests/golden/inputs.py: replay_stream) is pushed through the reference buffer
(capacity 32, wraps once), then np.random.seed(s); sample() -> stored outputs (numbers only)."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import inputs as gi  # noqa: E402
import make_golden as mg  # noqa: E402


def main():
    mg._install_stubs()
    import utils as ref_utils

    class _Numpy1Array:
        """Version skew only: the reference pins numpy 1.19 where np.array(x, copy=False) means "copy if needed";
        numpy 2 raises instead.  The harness gives the reference's module numpy-1 semantics for that one call."""

        def __getattr__(self, name):
            return getattr(np, name)

        @staticmethod
        def array(obj, *a, copy=True, **k):
            return np.asarray(obj, *a, **k) if copy is False else np.array(obj, *a, copy=copy, **k)

    ref_utils.np = _Numpy1Array()
    A, cap, B = 2, 32, 3
    buf = ref_utils.ReplayBuffer((9, 84, 84), (A,), capacity=cap, batch_size=B)
    for obs_frames, action, reward, next_frames, done in gi.replay_stream(seed=21, n=45, A=A):
        buf.add(ref_utils.LazyFrames(obs_frames), action, reward, ref_utils.LazyFrames(next_frames), done)
    out = {"idx_after": np.array([buf.idx]), "full": np.array([buf.full])}
    for k, s in enumerate((5, 6)):
        np.random.seed(s)
        obs, act, rew, nobs, nd = buf.sample()
        np.random.seed(s)
        idxs = np.random.randint(0, cap, size=B)
        out[f"s{k}/idxs"] = idxs
        out[f"s{k}/obs"] = obs.numpy().astype(np.uint8)
        assert np.array_equal(out[f"s{k}/obs"].astype(np.float32), obs.numpy())
        out[f"s{k}/next_obs"] = nobs.numpy().astype(np.uint8)
        out[f"s{k}/action"], out[f"s{k}/reward"], out[f"s{k}/not_done"] = act.numpy(), rew.numpy(), nd.numpy()
    np.savez_compressed(os.path.join(HERE, "replay_golden.npz"), **out)
    print("wrote replay_golden.npz", {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
