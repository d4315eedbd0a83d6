"""Device-resident replay buffer: drop-in for the sampling side of
`utils.ReplayBuffer` (src/utils.py:136-280) and `LazyFrames` (src/utils.py:283-322).

Same constructor / `add` / `sample` surface and the SAME index stream (`np.random.randint(0, capacity if full
else idx, size=n)` on numpy's global generator, src/utils.py:229-232), so a run seeded like the reference samples
the same transitions.  Different storage: every RGB frame is kept ONCE in a uint8 ring in HBM (frame stacks of
consecutive steps share frames exactly like the reference's LazyFrames share numpy arrays); `add` uploads only
frames it has not seen (normally one 21 KB frame per environment step) and `sample` is one gather launch per
tensor that writes the stacked (B, 3*stack, H, W) batch -- float32 like `tensor_buffer_samples`' `.float()`, or
uint8 for the fused uint8 front end (`augmentations.shift_conv`) -- straight into HBM.  No per-sample host loop,
no 2 x B x 63.5 KB PCIe transfer per update.

`save / requeue_save / load` keep the reference's incremental "{start}_{end}.tar" chunk files and bookkeeping
(src/utils.py:156-204); a chunk holds plain tensors (the frames its transitions reference, once each, plus their
index tables) instead of pickled LazyFrames objects, so it opens with `torch.load(weights_only=True)`.
`state_dict()/load_state_dict()` carry the whole ring.
"""
import os

import numpy as np
import torch

from . import _lib


class LazyFrames(object):
    """Frame stack that shares its frames with its neighbours (src/utils.py:283-322)."""

    def __init__(self, frames, extremely_lazy=True):
        self._frames = list(frames)

    @property
    def frames(self):
        return self._frames

    def __array__(self, dtype=None, copy=None):
        out = np.concatenate(self._frames, axis=0)
        return out.astype(dtype) if dtype is not None else out

    def __len__(self):
        return len(self._frames)

    def __getitem__(self, i):
        return np.concatenate(self._frames, axis=0)[i]


class ReplayBuffer(object):
    def __init__(self, obs_shape, action_shape, capacity, batch_size, prefill=True, device=None, frame_slack=None,
                 sample_dtype=torch.float32, index_stream="numpy"):
        c, h, w = obs_shape
        assert c % 3 == 0, "observations are stacks of RGB frames"
        if device is None:
            if not torch.cuda.is_available():
                raise _lib.AugclError("augcl_amd.replay.ReplayBuffer keeps its frames in HBM: a GPU is required")
            device = torch.device("cuda", torch.cuda.current_device())
        self.device = torch.device(device)
        self.capacity, self.batch_size = int(capacity), int(batch_size)
        self.stack, self.h, self.w = c // 3, h, w
        self.frame_bytes = 3 * h * w
        assert self.frame_bytes % 16 == 0, "frame size must keep the ring 16-byte aligned"
        # one new frame per step plus a fresh stack per episode: capacity + slack slots keep every frame referenced
        # by a live transition resident (checked at sample time)
        self.n_slots = self.capacity + (frame_slack if frame_slack is not None else self.capacity // 50 + 4 * self.stack + 16)
        self.frames = torch.empty((self.n_slots, self.frame_bytes), dtype=torch.uint8, device=self.device)
        self.frame_seq = 0  # frames ever stored; slot = seq % n_slots
        self._ids = {}      # id(frame ndarray) -> (seq, weakref-free strong ref to keep the id unique)
        self._ids_order = []
        self.obs_seq = np.zeros((self.capacity, self.stack), dtype=np.int64)
        self.next_seq = np.zeros((self.capacity, self.stack), dtype=np.int64)
        assert len(action_shape) == 1, "flat action vectors"
        A = self.A = int(action_shape[0])
        # device-resident tables sample() gathers from in ONE launch: per transition the ring slots of its two frame
        # stacks, and action | reward | not_done as one row
        self.scal = torch.zeros((self.capacity, A + 2), dtype=torch.float32, device=self.device)
        self.actions, self.rewards, self.not_dones = self.scal[:, :A], self.scal[:, A:A + 1], self.scal[:, A + 1:A + 2]
        self.slot_tab = torch.zeros((self.capacity, 2 * self.stack), dtype=torch.int32, device=self.device)
        pin = (lambda t: t.pin_memory()) if self.device.type == "cuda" else (lambda t: t)
        self._scal = pin(torch.empty((A + 2,), dtype=torch.float32))
        self._slots = pin(torch.empty((2 * self.stack,), dtype=torch.int32))
        self.sample_dtype = sample_dtype
        # where sample()'s transition indices come from: "numpy" = np.random.randint on the GLOBAL generator like the
        # reference (src/utils.py:229-232; one H2D copy of the indices per call), "device" = drawn by a Philox launch keyed
        # like the augmentations (augcl_philox_replay_indices): sample() / sample_shift_conv() are then launches only
        if index_stream not in ("numpy", "device"):
            raise ValueError("index_stream must be 'numpy' or 'device'")
        self.index_stream = index_stream
        self._idx_dev = None
        self.idx = 0
        self._adds = 0
        self.full = False
        self.last_save = 0
        self.last_requeue_save = 0
        self.refilled = False

    # ---- storing --------------------------------------------------------------------------------
    def _frames_of(self, obs):
        if hasattr(obs, "frames"):
            fr = [np.asarray(f) for f in obs.frames]
        else:
            a = np.asarray(obs)
            fr = [a[3 * k:3 * k + 3] for k in range(self.stack)]  # views: identity is lost, content is compared
        assert len(fr) == self.stack and all(f.shape == (3, self.h, self.w) for f in fr), "unexpected observation shape"
        return fr

    def _store_frame(self, f, allow_content_match):
        key = id(f)
        hit = self._ids.get(key)
        if hit is not None and hit[1] is f:
            return hit[0]
        if allow_content_match:  # ndarray observations: the previous step's frames are the only candidates
            for k2 in self._ids_order[-2 * self.stack:]:
                seq, ref = self._ids[k2]
                if ref.shape == f.shape and np.array_equal(ref, f):
                    return seq
        seq = self.frame_seq
        self.frame_seq += 1
        src = torch.from_numpy(np.ascontiguousarray(f, dtype=np.uint8).reshape(-1))
        self.frames[seq % self.n_slots].copy_(src, non_blocking=False)
        keep = f if not allow_content_match else np.array(f, copy=True)
        self._ids[id(keep) if allow_content_match else key] = (seq, keep)
        self._ids_order.append(id(keep) if allow_content_match else key)
        if len(self._ids_order) > 8 * self.stack:  # only recent frames can recur
            old = self._ids_order.pop(0)
            self._ids.pop(old, None)
        return seq

    def add(self, obs, action, reward, next_obs, done):
        """src/utils.py:216-227"""
        by_content = not hasattr(obs, "frames")
        self.obs_seq[self.idx] = [self._store_frame(f, by_content) for f in self._frames_of(obs)]
        self.next_seq[self.idx] = [self._store_frame(f, by_content) for f in self._frames_of(next_obs)]
        a = np.asarray(action, dtype=np.float32).reshape(-1)
        sc = self._scal.numpy()
        sc[:a.size], sc[a.size], sc[a.size + 1] = a, float(reward), float(not done)
        sl = self._slots.numpy()
        sl[:self.stack] = self.obs_seq[self.idx] % self.n_slots
        sl[self.stack:] = self.next_seq[self.idx] % self.n_slots
        self.scal[self.idx].copy_(self._scal)  # two small pinned -> HBM row copies per environment step
        self.slot_tab[self.idx].copy_(self._slots)
        self.idx = (self.idx + 1) % self.capacity
        self.full = self.full or self.idx == 0
        # every frame of every LIVE transition must still be in the ring; the oldest live transition is the next one to
        # be overwritten.  Checked here so that a ring sized for long episodes (one new frame per step) fails at the
        # step that breaks it -- short episodes / un-shared ndarray observations need a larger frame_slack -- and not
        # thousands of steps later inside sample().
        self._adds += 1
        oldest = self.idx if self.full else 0
        if min(self.obs_seq[oldest].min(), self.next_seq[oldest].min()) < self.frame_seq - self.n_slots:
            # frames stored per transition over the LIVE window (add counters do not survive load()): the oldest and newest
            # live transitions' first frames bracket the frames their episode stream consumed
            newest = (self.idx - 1) % self.capacity
            span = max(1, (self.capacity if self.full else max(self.idx, 1)))
            per = max(1.0, float(self.obs_seq[newest].max() - self.obs_seq[oldest].min() + 1) / span)
            need = int(np.ceil((per - 1.0) * self.capacity)) + 4 * self.stack + 16
            raise _lib.AugclError(
                f"replay frame ring too small ({self._adds} transitions stored, {self.frame_seq} frames): this stream stores "
                f"{per:.2f} new frames per transition, so {self.capacity} "
                f"live transitions need more than the {self.n_slots} slots of the ring; construct "
                f"ReplayBuffer(..., frame_slack={need}) or more")

    # ---- sampling -------------------------------------------------------------------------------
    def _get_idxs(self, n=None):
        """src/utils.py:229-232 (numpy's GLOBAL generator, like the reference)"""
        if n is None:
            n = self.batch_size
        return np.random.randint(0, self.capacity if self.full else self.idx, size=n)

    def _device_idxs(self, n=None):
        """B transition indices drawn on the device (no host step; the ring invariant add() maintains makes every live
        transition valid, so nothing needs checking on the host)."""
        from . import augmentations
        B = self.batch_size if n is None else int(n)
        n_valid = self.capacity if self.full else self.idx
        if n_valid < 1:
            raise IndexError("replay sample: the buffer is empty")
        if self._idx_dev is None or self._idx_dev.numel() != B:
            self._idx_dev = torch.empty(B, dtype=torch.int64, device=self.device)
        seed, off, s0 = augmentations._next_call()
        _lib.call("augcl_philox_replay_indices", seed, off, s0, n_valid, self._idx_dev.data_ptr(), B, _lib.current_stream(self.device))
        return self._idx_dev

    def _gather(self, seqs, dtype):
        live = self.frame_seq - self.n_slots
        if seqs.min() < live:
            raise _lib.AugclError("replay frame ring too small: a sampled transition's frame was overwritten "
                                  f"(slots={self.n_slots}); raise frame_slack")
        B = seqs.shape[0]
        slots = torch.from_numpy((seqs % self.n_slots).astype(np.int32)).to(self.device)
        out = torch.empty((B, 3 * self.stack, self.h, self.w), dtype=dtype, device=self.device)
        _lib.call("augcl_replay_gather", self.frames.data_ptr(), slots.data_ptr(), out.data_ptr(), B, self.stack,
                  self.frame_bytes, 1 if dtype == torch.float32 else 0, _lib.current_stream(self.device))
        return out

    def _sync_tables(self, a=0, b=None):
        """Refresh the device slot table for transitions [a, b) from the host sequence tables."""
        b = self.capacity if b is None else b
        tab = np.concatenate([self.obs_seq[a:b], self.next_seq[a:b]], axis=1) % self.n_slots
        self.slot_tab[a:b] = torch.from_numpy(tab.astype(np.int32)).to(self.device)

    def _checked_idxs(self, idxs):
        idxs = np.ascontiguousarray(idxs, dtype=np.int64)
        if idxs.size and (idxs.min() < 0 or idxs.max() >= (self.capacity if self.full else max(self.idx, 1))):
            raise IndexError("replay sample: transition index outside the filled part of the buffer")
        if idxs.size and min(self.obs_seq[idxs].min(), self.next_seq[idxs].min()) < self.frame_seq - self.n_slots:
            raise _lib.AugclError("replay frame ring too small: a sampled transition's frame was overwritten "
                                  f"(slots={self.n_slots}); raise frame_slack")
        return idxs

    def sample(self, n=None, idxs=None, dtype=None):
        """src/utils.py:272-280: returns (obs, actions, rewards, next_obs, not_dones) on the device -- one H2D copy
        of the drawn indices and one launch (augcl_replay_sample) for all five tensors."""
        if idxs is None and self.index_stream == "device":
            it = self._device_idxs(n)
        else:
            if idxs is None:
                idxs = self._get_idxs(n)
            idxs = self._checked_idxs(idxs)
            it = torch.from_numpy(idxs).to(self.device, non_blocking=True)
        dtype = dtype or self.sample_dtype
        B, A = it.shape[0], self.A
        obs = torch.empty((B, 3 * self.stack, self.h, self.w), dtype=dtype, device=self.device)
        next_obs = torch.empty_like(obs)
        sc = torch.empty(B * (A + 2), dtype=torch.float32, device=self.device)
        _lib.call("augcl_replay_sample", self.frames.data_ptr(), self.slot_tab.data_ptr(), self.scal.data_ptr(), it.data_ptr(),
                  obs.data_ptr(), next_obs.data_ptr(), sc.data_ptr(), B, self.stack, self.frame_bytes, A + 2,
                  1 if dtype == torch.float32 else 0, _lib.current_stream(self.device))
        return (obs, sc[:B * A].view(B, A), sc[B * A:B * A + B].view(B, 1), next_obs, sc[B * A + B:].view(B, 1))

    def sample_shift_conv(self, n=None, idxs=None, pad=4, *, oy=None, ox=None, weights=None, out=None, out_next=None,
                          next_dtype=torch.float32):
        """sample() + SVEA's front end (src/algorithms/svea.py:66-68 and :49) in one pass over the uint8 ring: returns
        (obs_both, actions, rewards, next_obs, not_dones) where obs_both is the (2B, C, H, W) stack
        [random_shift(obs); random_conv(random_shift(obs))] -- the sampled obs batch itself is never materialised
        (63.5 KB read per observation instead of 63.5 KB read + 254 KB written + 254 KB read again).  Bit-identical to
        `sample()` followed by `augmentations.shift_conv(obs, stacked=True)` with the same parameters / Philox call.
        out / out_next: write obs_both / next_obs into these tensors (a hipGraph's static inputs: no copy afterwards).
        next_dtype = torch.uint8: next_obs stays in the ring's own format (it is never augmented on this path, svea.py:66-68, so
        its pixels are integers): a quarter of the bytes written here and read by the encoder's first layer, which takes uint8
        observations as they are (csrc/encoder_first_u8.hip)."""
        from . import augmentations
        if idxs is None and self.index_stream == "device":
            it = self._device_idxs(n)
        else:
            if idxs is None:
                idxs = self._get_idxs(n)
            idxs = self._checked_idxs(idxs)
            it = torch.from_numpy(idxs).to(self.device, non_blocking=True)
        B, A, C = it.shape[0], self.A, 3 * self.stack
        both = out if out is not None else torch.empty((2 * B, C, self.h, self.w), dtype=torch.float32, device=self.device)
        assert both.shape == (2 * B, C, self.h, self.w) and both.dtype == torch.float32 and both.is_contiguous()
        oy = augmentations._i32(oy, B, self.device, "oy")
        ox = augmentations._i32(ox, B, self.device, "ox")
        weights = augmentations._conv_weights(weights, B, self.device)
        seed, off, s0 = augmentations._next_call()
        st = _lib.current_stream(self.device)
        _lib.call("augcl_replay_shift_conv_u8_f32", self.frames.data_ptr(), self.slot_tab.data_ptr(), it.data_ptr(),
                  2 * self.stack, both[:B].data_ptr(), both[B:].data_ptr(), B, self.stack, self.h, self.w, int(pad),
                  _lib.ptr(oy), _lib.ptr(ox), _lib.ptr(weights), seed, off, s0, st)
        if out_next is not None:
            next_dtype = out_next.dtype
        assert next_dtype in (torch.float32, torch.uint8)
        next_obs = out_next if out_next is not None else torch.empty((B, C, self.h, self.w), dtype=next_dtype, device=self.device)
        assert next_obs.shape == (B, C, self.h, self.w) and next_obs.is_contiguous()
        sc = torch.empty(B * (A + 2), dtype=torch.float32, device=self.device)
        _lib.call("augcl_replay_sample", self.frames.data_ptr(), self.slot_tab.data_ptr(), self.scal.data_ptr(), it.data_ptr(),
                  None, next_obs.data_ptr(), sc.data_ptr(), B, self.stack, self.frame_bytes, A + 2,
                  1 if next_dtype == torch.float32 else 0, st)
        return (both, sc[:B * A].view(B, A), sc[B * A:B * A + B].view(B, 1), next_obs, sc[B * A + B:].view(B, 1))

    def sample_obs(self, n=None):
        return self._gather(self.obs_seq[self._get_idxs(n)], torch.float32)

    def sample_drq(self, n=None, pad=4):
        """src/utils.py:258-270"""
        from . import augmentations
        obs, actions, rewards, next_obs, not_dones = self.sample(n)
        return augmentations.random_shift(obs, pad), actions, rewards, augmentations.random_shift(next_obs, pad), not_dones

    # ---- persistence ----------------------------------------------------------------------------
    def __save__(self, save_dir, last_save_idx):
        """src/utils.py:156-168: transitions [last_save_idx, idx) -> "<start>_<end>.tar"."""
        if self.idx == last_save_idx:
            return self.idx
        a, b = last_save_idx, self.idx
        # after the ring has wrapped (idx < last_save_idx) the reference's slices [a:b] are EMPTY and it still writes the
        # "<a>_<b>.tar" chunk (src/utils.py:159-167); same here: an empty chunk, which load() accepts
        seqs = np.unique(np.concatenate([self.obs_seq[a:b].ravel(), self.next_seq[a:b].ravel()])).astype(np.int64)
        if seqs.size and seqs.min() < self.frame_seq - self.n_slots:
            raise _lib.AugclError("replay save: a frame of an unsaved transition was already overwritten")
        slots = torch.from_numpy(seqs % self.n_slots).to(self.device)
        payload = dict(frames=self.frames[slots].cpu(), seqs=torch.from_numpy(seqs), obs_seq=torch.from_numpy(self.obs_seq[a:b].copy()),
                       next_seq=torch.from_numpy(self.next_seq[a:b].copy()), actions=self.actions[a:b].cpu(),
                       rewards=self.rewards[a:b].cpu(), not_dones=self.not_dones[a:b].cpu())
        torch.save(payload, os.path.join(save_dir, "%d_%d.tar" % (a, b)))
        return self.idx

    def save(self, save_dir):
        self.last_save = self.__save__(save_dir=save_dir, last_save_idx=self.last_save)

    def requeue_save(self, save_dir):
        self.last_requeue_save = self.__save__(save_dir=save_dir, last_save_idx=self.last_requeue_save)

    def load(self, save_dir, end_step=None, is_requeue_load=True, is_load=True):
        """src/utils.py:178-204: chunks in order of their start index, each must begin where the buffer stands."""
        if self.refilled:
            return
        chunks = sorted(os.listdir(save_dir), key=lambda x: int(x.split("_")[0]))
        remap = {}  # frame sequence number in the saved buffer -> here
        for chunk in chunks:
            start, end = [int(x) for x in chunk.split(".")[0].split("_")]
            payload = torch.load(os.path.join(save_dir, chunk), map_location="cpu", weights_only=True)
            assert self.idx == start, (f"self.idx: {self.idx} DNE start: {start} for save_dir: {save_dir}. "
                                       f"Current chunk is {chunk}. Chucks: {chunks}.")
            if end < start:  # a chunk written after the ring wrapped: empty, like the reference's (its slices are empty)
                self.idx = end
                if end == end_step:
                    break
                continue
            seqs = payload["seqs"].numpy()
            new = [i for i, s in enumerate(seqs.tolist()) if s not in remap]
            if new:
                first = self.frame_seq
                for k, i in enumerate(new):
                    remap[int(seqs[i])] = first + k
                self.frame_seq += len(new)
                slots = torch.from_numpy((np.arange(first, first + len(new)) % self.n_slots)).to(self.device)
                self.frames[slots] = payload["frames"][torch.as_tensor(new)].to(self.device)
            lut = np.vectorize(remap.__getitem__, otypes=[np.int64])
            self.obs_seq[start:end] = lut(payload["obs_seq"].numpy())
            self.next_seq[start:end] = lut(payload["next_seq"].numpy())
            self._sync_tables(start, end)
            self.actions[start:end] = payload["actions"].to(self.device)
            self.rewards[start:end] = payload["rewards"].to(self.device)
            self.not_dones[start:end] = payload["not_dones"].to(self.device)
            self.idx = end
            self._adds += end - start  # loaded transitions count as stored ones (the counters of a resumed run)
            if end == end_step:
                break
        self._ids.clear()
        self._ids_order.clear()
        if is_requeue_load:
            self.last_requeue_save = self.idx
        if is_load:
            self.last_save = self.idx

    def state_dict(self):
        return dict(frames=self.frames.cpu(), frame_seq=self.frame_seq, obs_seq=self.obs_seq.copy(),
                    next_seq=self.next_seq.copy(), actions=self.actions.cpu(), rewards=self.rewards.cpu(),
                    not_dones=self.not_dones.cpu(), idx=self.idx, full=self.full, adds=self._adds)

    def load_state_dict(self, sd):
        self.frames.copy_(sd["frames"])
        self.frame_seq, self.idx, self.full = int(sd["frame_seq"]), int(sd["idx"]), bool(sd["full"])
        # transitions ever stored: saved by state_dict(); a dict from before that field = the live transitions
        self._adds = int(sd.get("adds", self.capacity if self.full else self.idx))
        self.obs_seq[:], self.next_seq[:] = sd["obs_seq"], sd["next_seq"]
        self._sync_tables()
        self.actions.copy_(sd["actions"])
        self.rewards.copy_(sd["rewards"])
        self.not_dones.copy_(sd["not_dones"])
        self._ids.clear()
        self._ids_order.clear()
