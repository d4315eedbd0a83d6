"""Data parallelism for the update(): one process per GPU, batch sharded over ranks, every rank a full
replica; per optimiser step ONE sum all-reduce of the group's flat gradient bucket (RCCL over xGMI,
`backend="nccl"` on ROCm), the 1/world_size folded into the fused Adam kernel.  No collective exists
on the augmentation path (samples are independent; in-kernel Philox parameters are keyed by the GLOBAL
sample index so results do not depend on the sharding).

The reference has no distributed code at all: F.mse_loss / .mean() are batch means
(src/algorithms/sac.py:109,124,139) and LayerNorm is per-sample, so equal shards + gradient averaging
reproduce the full-batch gradient up to summation order.

Works on any device torch.distributed supports: the CPU/gloo tests exercise exactly this code.
"""
import os

import torch
import torch.distributed as dist

_default = None


class GradReducer:
    def __init__(self, group=None):
        assert dist.is_initialized(), "init torch.distributed first (dist.init_from_env)"
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self._scalars = None

    def allreduce_sum_(self, flat_grad):
        """In-place SUM over ranks of one flat bucket; returns the scale (1/world) the caller must apply."""
        if self.world > 1:
            dist.all_reduce(flat_grad, op=dist.ReduceOp.SUM, group=self.group)
        return 1.0 / self.world

    def allreduce_sum_async(self, flat_grad):
        """The same SUM, issued without waiting for it: the collective runs on the backend's own stream (RCCL) / thread
        (gloo) behind everything already queued on the current stream; `.wait()` on the returned handle (None for one
        rank) makes the current stream wait for the result.  What lets the all-reduce of the heads' gradients -- 97 % of
        a critic's bucket -- run under the conv stack's backward."""
        if self.world > 1:
            return dist.all_reduce(flat_grad, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        return None

    def allreduce_mean_(self, t):
        if self.world > 1:
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
            t.div_(self.world)
        return t

    def mean_scalar(self, value):
        """Mean over ranks of a logged scalar (each rank logs its shard's mean)."""
        if self.world == 1:
            return value
        v = value.detach().clone().to(torch.float64).reshape(1)
        dist.all_reduce(v, op=dist.ReduceOp.SUM, group=self.group)
        return (v / self.world).reshape(())

    def shard(self, n):
        """[begin, end) of this rank's slice of a global batch of n samples."""
        per = n // self.world
        assert per * self.world == n, f"global batch {n} not divisible by world size {self.world}"
        return self.rank * per, (self.rank + 1) * per

    def broadcast_(self, flat_param, src=0):
        """Make replicas identical at start-up."""
        if self.world > 1:
            dist.broadcast(flat_param, src=src, group=self.group)

    def __getstate__(self):  # agents are pickled whole (torch.save(agent), src/train.py:229)
        return {}

    def __setstate__(self, state):
        self.group, self.world, self.rank, self._scalars = None, 1, 0, None


def init_from_env(backend=None):
    """Initialise torch.distributed from RANK / WORLD_SIZE / MASTER_* (torchrun).  Returns the reducer
    (None for a single process)."""
    global _default
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world <= 1:
        _default = None
        return None
    if not dist.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        kw = {}
        if backend == "nccl":
            dev = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")))
            torch.cuda.set_device(dev)
            kw["device_id"] = dev
        dist.init_process_group(backend, **kw)
    _default = GradReducer()
    return _default


def seed_all(seed, rank=None):
    """utils.set_seed_everywhere (src/utils.py:90-95) for data-parallel ranks.  Replicas start identical through the
    parameter broadcast at construction, so what must DIFFER per rank is everything drawn afterwards: the actor's noise
    (torch's generators), the replay indices (numpy's) -- both seeded with seed + rank -- while the in-kernel
    augmentation streams keep ONE seed on all ranks and are told apart by the global sample index
    (`sample_base`: rank r owns samples [r*B, (r+1)*B) of the global batch)."""
    import numpy as np

    from . import augmentations
    if rank is None:
        rank = _default.rank if _default is not None else int(os.environ.get("RANK", "0"))
    torch.manual_seed(int(seed) + rank)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(int(seed) + rank)
    np.random.seed((int(seed) + rank) % (2 ** 32))
    augmentations.manual_seed(seed)


def sample_base(reducer, local_batch):
    """Global index of this rank's first sample: what augmentations.set_sample_base() gets before every front end."""
    return 0 if reducer is None else reducer.rank * int(local_batch)


def set_default_reducer(r):
    global _default
    _default = r


def default_reducer():
    return _default
