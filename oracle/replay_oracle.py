"""oracle/replay_oracle.py -- numpy restatement of the reference's replay sampling
(src/utils.py:136-151, 216-240, 272-280, 283-299).  TEST INFRASTRUCTURE ONLY.
Pinned against outputs of the reference's own ReplayBuffer (tests/golden/make_golden_replay.py)."""
import numpy as np


class ReplayOracle:
    def __init__(self, action_shape, capacity):
        self.capacity = capacity
        self.obses = [None] * capacity  # (list of frame arrays, list of frame arrays): frames are shared objects
        self.actions = np.empty((capacity, *action_shape), dtype=np.float32)
        self.rewards = np.empty((capacity, 1), dtype=np.float32)
        self.not_dones = np.empty((capacity, 1), dtype=np.float32)
        self.idx, self.full = 0, False

    def add(self, obs_frames, action, reward, next_frames, done):
        self.obses[self.idx] = (list(obs_frames), list(next_frames))
        self.actions[self.idx] = action
        self.rewards[self.idx] = reward
        self.not_dones[self.idx] = not done
        self.idx = (self.idx + 1) % self.capacity
        self.full = self.full or self.idx == 0

    def draw(self, n):
        return np.random.randint(0, self.capacity if self.full else self.idx, size=n)

    def sample(self, idxs):
        obs = np.stack([np.concatenate(self.obses[i][0], axis=0) for i in idxs]).astype(np.float32)
        nobs = np.stack([np.concatenate(self.obses[i][1], axis=0) for i in idxs]).astype(np.float32)
        return obs, self.actions[idxs], self.rewards[idxs], nobs, self.not_dones[idxs]
