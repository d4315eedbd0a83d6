"""Drop-in for the hot-path operators of the reference's `src/augmentations.py`.

Same names, argument meaning and return conventions as the reference:

    aug_to_func["shift" | "conv" | "overlay" | "color_jitter" | "identity"](x, **params)

x is a contiguous (B, C, H, W) float32 tensor holding 0..255 values on the GPU
(what `ReplayBuffer.sample()` returns, src/utils.py:206-214); a uint8 tensor is also
accepted where a fused `.float()` exists.  Every op returns a NEW tensor except
`color_jitter`, which (like the reference, augmentations.py:236-237) mutates its
argument and returns a view of it.

All arithmetic runs in hand-written HIP kernels behind the C ABI (include/augcl.h).
Random parameters are either drawn INSIDE the kernels from a counter-based Philox
stream keyed by (seed, call counter, global sample index) -- no host round trip, no
extra kernel, identical results however a batch is sharded over ranks -- or passed
explicitly through the keyword-only arguments (what the parity tests do).
"""
import numpy as np
import torch

from . import _lib

# ---------------------------------------------------------------- RNG-internal state
_rng = {"seed": 0x5EED0A06C1, "offset": 0, "sample0": 0}


def manual_seed(seed):
    """Seed the in-kernel Philox streams (the analogue of utils.set_seed_everywhere,
    src/utils.py:90-95, for parameters that never touch the host)."""
    _rng["seed"] = int(seed) & 0xFFFFFFFFFFFFFFFF
    _rng["offset"] = 0


def set_sample_base(sample0):
    """Global index of this rank's first sample (data parallel: rank * local_batch)."""
    _rng["sample0"] = int(sample0)


def _next_call():
    off = _rng["offset"]
    _rng["offset"] = off + 1
    return _rng["seed"], off, _rng["sample0"]


def _check_bchw(x, name):
    if x.dim() != 4:
        raise ValueError(f"{name}: expected a (B,C,H,W) tensor, got shape {tuple(x.shape)}")
    _lib.require_cuda(x, name)


def _i32(t, n, device, name):
    if t is None:
        return None
    t = torch.as_tensor(t, device=device).to(torch.int32).contiguous()
    if t.numel() != n:
        raise ValueError(f"{name}: expected {n} values, got {t.numel()}")
    return t


# ---------------------------------------------------------------- operators
def identity(x):
    """src/augmentations.py:374-375"""
    return x


def random_shift(imgs, pad=4, *, oy=None, ox=None, out=None):
    """src/augmentations.py:378-382: replicate-pad by `pad`, then a random (h,w) crop with one
    offset per sample (shared by its C channels).  oy/ox (ints in [0,2*pad]) override the RNG."""
    _check_bchw(imgs, "random_shift")
    B, C, H, W = imgs.shape
    oy = _i32(oy, B, imgs.device, "oy")
    ox = _i32(ox, B, imgs.device, "ox")
    if (oy is None) != (ox is None):
        raise ValueError("random_shift: give both oy and ox or neither")
    seed, off, s0 = _next_call()
    st = _lib.current_stream(imgs.device)
    if out is None:
        out = torch.empty((B, C, H, W), dtype=torch.float32, device=imgs.device)
    if imgs.dtype == torch.float32:
        fn = "augcl_random_shift_f32"
    elif imgs.dtype == torch.uint8:
        fn = "augcl_random_shift_u8_f32" if out.dtype == torch.float32 else "augcl_random_shift_u8"
    else:
        raise TypeError(f"random_shift: unsupported dtype {imgs.dtype}")
    _lib.call(fn, _lib.ptr(imgs), _lib.ptr(out), B, C, H, W, int(pad), _lib.ptr(oy), _lib.ptr(ox), seed, off, s0, st)
    return out


def random_conv(x, *, weights=None, out=None):
    """src/augmentations.py:265-274: per-sample random 3x3 conv (3->3 channels, shared by the
    sample's frames) on x/255 with replicate padding, sigmoid, *255.
    weights: optional (B,3,3,3,3) tensor replacing the per-sample N(0,1) draw."""
    _check_bchw(x, "random_conv")
    if x.dtype != torch.float32:
        raise TypeError("random_conv: float32 input expected")
    B, C, H, W = x.shape
    if C % 3:
        raise ValueError("random_conv: channel count must be a multiple of 3")
    if weights is not None:
        weights = torch.as_tensor(weights, device=x.device, dtype=torch.float32).contiguous()
        if weights.numel() != B * 81:
            raise ValueError("random_conv: weights must hold B*81 values")
    seed, off, s0 = _next_call()
    if out is None:
        out = torch.empty_like(x)
    _lib.call("augcl_random_conv_f32", _lib.ptr(x), _lib.ptr(out), B, C // 3, H, W, _lib.ptr(weights), seed, off, s0,
              _lib.current_stream(x.device))
    return out


def shift_conv(obs, pad=4, *, oy=None, ox=None, weights=None, want_shift=True, want_aug=True, stacked=False):
    """Fused SVEA/AugCL front end (src/algorithms/svea.py:68 + :49): one pass over the (uint8 or
    f32) observation producing random_shift(obs) and random_conv(random_shift(obs)).
    stacked=True returns ONE (2B,C,H,W) tensor: rows [0,B) = shifted, rows [B,2B) = shifted + conv."""
    _check_bchw(obs, "shift_conv")
    B, C, H, W = obs.shape
    oy = _i32(oy, B, obs.device, "oy")
    ox = _i32(ox, B, obs.device, "ox")
    if weights is not None:
        weights = torch.as_tensor(weights, device=obs.device, dtype=torch.float32).contiguous()
    seed, off, s0 = _next_call()
    if stacked:
        both = torch.empty((2 * B, C, H, W), dtype=torch.float32, device=obs.device)
        o_s, o_a = both[:B], both[B:]
    else:
        o_s = torch.empty((B, C, H, W), dtype=torch.float32, device=obs.device) if want_shift else None
        o_a = torch.empty((B, C, H, W), dtype=torch.float32, device=obs.device) if want_aug else None
    fn = {torch.uint8: "augcl_shift_conv_u8_f32", torch.float32: "augcl_shift_conv_f32"}[obs.dtype]
    _lib.call(fn, _lib.ptr(obs), _lib.ptr(o_s), _lib.ptr(o_a), B, C // 3, H, W, int(pad), _lib.ptr(oy), _lib.ptr(ox),
              _lib.ptr(weights), seed, off, s0, _lib.current_stream(obs.device))
    return both if stacked else (o_s, o_a)


# overlay image source: the reference streams COCO / Places365 through a 16-worker DataLoader
# (augmentations.py:19-86).  That host-side loader is outside the accelerated path; anything
# callable as source(batch_size, image_size, device) -> (B,3,H,W) f32 in [0,1] can be plugged in.
_overlay_source = None


def set_overlay_source(source):
    global _overlay_source
    _overlay_source = source


class SyntheticOverlaySource:
    """U[0,1) images generated on the device (benchmarks / tests: the datasets are not shipped)."""

    def __init__(self, seed=0):
        self.gen = None
        self.seed = seed

    def __call__(self, batch_size, image_size, device):
        if self.gen is None or self.gen.device != torch.device(device):
            self.gen = torch.Generator(device=device)
            self.gen.manual_seed(self.seed)
        return torch.rand((batch_size, 3, image_size, image_size), generator=self.gen, device=device)


def random_overlay(x, dataset="coco", *, imgs=None, alpha=0.5, out=None):
    """src/augmentations.py:253-262: ((1-alpha)*(x/255.0) + alpha*imgs)*255.0 with alpha=0.5; the same
    image is laid over every frame of a sample."""
    _check_bchw(x, "random_overlay")
    if x.dtype != torch.float32:
        raise TypeError("random_overlay: float32 input expected")
    B, C, H, W = x.shape
    if imgs is None:
        if dataset not in ("coco", "places365_standard"):
            raise NotImplementedError(f'overlay has not been implemented for dataset "{dataset}"')
        if _overlay_source is None:
            raise FileNotFoundError("failed to find image data: no overlay source registered "
                                    "(augmentations.set_overlay_source)")
        imgs = _overlay_source(B, W, x.device)
    imgs = torch.as_tensor(imgs, device=x.device, dtype=torch.float32).contiguous()
    if tuple(imgs.shape) != (B, 3, H, W):
        raise ValueError(f"random_overlay: overlay batch must be {(B, 3, H, W)}, got {tuple(imgs.shape)}")
    if out is None:
        out = torch.empty_like(x)
    _lib.call("augcl_random_overlay_f32", _lib.ptr(x), _lib.ptr(imgs), _lib.ptr(out), B, C // 3, H, W, float(alpha),
              _lib.current_stream(x.device))
    return out


JITTER_RANGES = dict(contrast=(0.0, 2.0), saturation=(0.0, 3.0), hue=(-0.5, 0.5))  # get_jitter_model, L15-16


def color_jitter(imgs, p=1, *, idx=None, order=None, contrast=None, saturation=None, hue=None, mode="compat"):
    """src/augmentations.py:225-237.  In place; returns imgs.view(b,c,h,w).

    One ColorJitter(brightness=0, contrast=(0,2), saturation=(0,3), hue=(-.5,.5)) parameter set
    and op order per call; applied to frames idx ~ randint(0, b, int(p*b*frames)) of the
    (b*frames,3,h,w) view -- the reference's own (quirky) index range is kept.
    mode="compat": literal float path (clamp to [0,1] on 0..255 data, as the reference runs it);
    mode="sane": the frame is scaled to [0,1] for the jitter and back."""
    _check_bchw(imgs, "color_jitter")
    if imgs.dtype != torch.float32:
        raise TypeError("color_jitter: float32 input expected")
    b, c, h, w = imgs.shape
    frames = c // 3
    if idx is None:
        idx = np.random.randint(0, b, int(p * b * frames))
    idx_t = torch.as_tensor(np.asarray(idx), dtype=torch.int32).to(imgs.device).contiguous()
    if order is None:
        order = torch.randperm(4).tolist()
    order = [int(v) for v in order]
    # torchvision draws a factor only when its op comes up; identity values otherwise
    draw = lambda lo, hi: float(torch.empty(1).uniform_(lo, hi).item())  # noqa: E731
    for op in order:
        if op == 1 and contrast is None:
            contrast = draw(*JITTER_RANGES["contrast"])
        if op == 2 and saturation is None:
            saturation = draw(*JITTER_RANGES["saturation"])
        if op == 3 and hue is None:
            hue = draw(*JITTER_RANGES["hue"])
    scale = {"compat": 1.0, "sane": 255.0}[mode]
    order_arr = (_lib.ctypes.c_int32 * 4)(*order)
    _lib.call("augcl_color_jitter_f32", _lib.ptr(imgs), b, frames, h, w, _lib.ptr(idx_t), int(idx_t.numel()),
              _lib.ctypes.cast(order_arr, _lib.ctypes.c_void_p), float(contrast), float(saturation), float(hue), 1.0,
              scale, _lib.current_stream(imgs.device))
    return imgs.view(b, c, h, w)


aug_to_func = {
    "conv": random_conv,
    "shift": random_shift,
    "color_jitter": color_jitter,
    "identity": identity,
    "overlay": random_overlay,
}
