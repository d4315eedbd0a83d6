"""Drop-in for the hot-path operators of the reference's `src/augmentations.py`.

Same names, argument meaning and return conventions as the reference:

    aug_to_func["shift" | "conv" | "overlay" | "color_jitter" | "identity" |
                "crop" | "flip" | "rotate" | "cutout_color" | "grayscale" | "splice" | "CS_splice"](x, **params)

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


def set_conv_weight_stream(name):
    """Where random_conv's per-sample weights come from when none are passed: "philox" (default: drawn inside the kernel,
    keyed by (seed, call, global sample index) -- no host step, invariant to sharding) or "reference": the reference's own
    stream, one CPU `torch.randn(3, 3, 3, 3)` per sample in order on torch's default generator
    (src/augmentations.py:268-269), uploaded once per batch -- a run seeded like a reference run then sees the SAME
    augmented batches (graph capture needs the device stream: hipGraph replay keeps "philox")."""
    if name not in ("philox", "reference"):
        raise ValueError(f"conv weight stream must be 'philox' or 'reference', got {name!r}")
    _rng["conv_stream"] = name


def _conv_weights(weights, B, device):
    """explicit (B,3,3,3,3) weights, the string "reference" / the module's reference-stream mode, or None (in-kernel draw)"""
    if weights is None and _rng.get("conv_stream", "philox") == "reference":
        weights = "reference"
    if isinstance(weights, str):
        if weights != "reference":
            raise ValueError("random_conv: weights must be a tensor, 'reference' or None")
        weights = torch.stack([torch.randn(3, 3, 3, 3) for _ in range(B)])  # the reference's draw order, CPU generator
    if weights is not None:
        weights = torch.as_tensor(weights, device=device, dtype=torch.float32).contiguous()
        if weights.numel() != B * 81:
            raise ValueError("random_conv: weights must hold B*81 values")
    return weights


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
    weights: optional (B,3,3,3,3) tensor replacing the per-sample N(0,1) draw, or "reference" = draw them like the
    reference does (one CPU torch.randn(3,3,3,3) per sample, in order: see set_conv_weight_stream)."""
    _check_bchw(x, "random_conv")
    if x.dtype != torch.float32:
        raise TypeError("random_conv: float32 input expected")
    B, C, H, W = x.shape
    if C % 3:
        raise ValueError("random_conv: channel count must be a multiple of 3")
    weights = _conv_weights(weights, B, x.device)
    seed, off, s0 = _next_call()
    if out is None:
        out = torch.empty_like(x)
    _lib.call("augcl_random_conv_f32", _lib.ptr(x), _lib.ptr(out), B, C // 3, H, W, _lib.ptr(weights), seed, off, s0,
              _lib.current_stream(x.device))
    return out


def shift_conv(obs, pad=4, *, oy=None, ox=None, weights=None, want_shift=True, want_aug=True, stacked=False, out=None):
    """Fused SVEA/AugCL front end (src/algorithms/svea.py:68 + :49): one pass over the (uint8 or
    f32) observation producing random_shift(obs) and random_conv(random_shift(obs)).
    stacked=True returns ONE (2B,C,H,W) tensor: rows [0,B) = shifted, rows [B,2B) = shifted + conv."""
    _check_bchw(obs, "shift_conv")
    B, C, H, W = obs.shape
    oy = _i32(oy, B, obs.device, "oy")
    ox = _i32(ox, B, obs.device, "ox")
    weights = _conv_weights(weights, B, obs.device)
    seed, off, s0 = _next_call()
    if stacked:
        both = out if out is not None else torch.empty((2 * B, C, H, W), dtype=torch.float32, device=obs.device)
        assert both.shape == (2 * B, C, H, W) and both.dtype == torch.float32 and both.is_contiguous()
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


# ---------------------------------------------------------------- rest of the registry (index / select operators)
_AFFINE = {0: (1, 0, 0, 0, 1, 0), 1: (0, 1, 0, -1, 0, -1), 2: (-1, 0, -2, 0, -1, -1), 3: (0, -1, -2, 1, 0, 0),
           4: (1, 0, 0, 0, -1, -1)}  # -1 -> W-1, -2 -> H-1 (filled in per call)


def _affine_gather(x, coef, Hout, Wout, name):
    _check_bchw(x, name)
    if x.dtype != torch.float32:
        raise TypeError(f"{name}: float32 input expected")
    B, C, H, W = x.shape
    x = x.contiguous()
    coef = np.ascontiguousarray(coef, dtype=np.int32).reshape(B, 6)
    # every source index the map can produce must be inside the frame (corners suffice: the map is affine)
    for ys, xs in ((0, 0), (Hout - 1, 0), (0, Wout - 1), (Hout - 1, Wout - 1)):
        sy = coef[:, 0] * ys + coef[:, 1] * xs + coef[:, 2]
        sx = coef[:, 3] * ys + coef[:, 4] * xs + coef[:, 5]
        if B and (sy.min() < 0 or sy.max() >= H or sx.min() < 0 or sx.max() >= W):
            raise ValueError(f"{name}: source window leaves the {H}x{W} frame")
    coef_t = torch.from_numpy(coef).to(x.device, non_blocking=True)
    out = torch.empty((B, C, Hout, Wout), dtype=torch.float32, device=x.device)
    _lib.call("augcl_affine_gather_f32", _lib.ptr(x), _lib.ptr(out), B, C, H, W, Hout, Wout, _lib.ptr(coef_t),
              _lib.current_stream(x.device))
    return out


def _kind_coef(kind, H, W):
    rows = np.array([_AFFINE[int(k)] for k in kind], dtype=np.int32).reshape(-1, 6)
    fix = np.where(rows[:, [2, 5]] == -1, W - 1, np.where(rows[:, [2, 5]] == -2, H - 1, rows[:, [2, 5]]))
    rows[:, [2, 5]] = fix
    return rows


def random_crop(x, size=84, w1=None, h1=None, return_w1_h1=False):
    """src/augmentations.py:385-414: out[i,c,y,x] = x[i,c,w1[i]+y,h1[i]+x]; w1,h1 ~ LongTensor(n).random_(0, crop_max)
    on torch's default CPU generator (upper bound exclusive, as in the reference), or given."""
    assert (w1 is None and h1 is None) or (w1 is not None and h1 is not None), \
        "must either specify both w1 and h1 or neither of them"
    _check_bchw(x, "random_crop")
    n = x.shape[0]
    crop_max = x.shape[-1] - size
    if crop_max <= 0:
        return (x, None, None) if return_w1_h1 else x
    if w1 is None:
        w1 = torch.LongTensor(n).random_(0, crop_max)
        h1 = torch.LongTensor(n).random_(0, crop_max)
    w1n = torch.as_tensor(w1).cpu().numpy().astype(np.int32).reshape(n)
    h1n = torch.as_tensor(h1).cpu().numpy().astype(np.int32).reshape(n)
    coef = np.tile(np.array([1, 0, 0, 0, 1, 0], np.int32), (n, 1))
    coef[:, 2], coef[:, 5] = w1n, h1n
    cropped = _affine_gather(x, coef, size, size, "random_crop")
    return (cropped, w1, h1) if return_w1_h1 else cropped


def random_flip(images, p=0.2, *, mask=None):
    """src/augmentations.py:139-168: horizontal flip of the samples with np.random.uniform <= p."""
    _check_bchw(images, "random_flip")
    B, _, H, W = images.shape
    if mask is None:
        mask = np.random.uniform(0.0, 1.0, size=(B,)) <= p
    kind = np.where(np.asarray(mask, bool).reshape(B), 4, 0)
    return _affine_gather(images, _kind_coef(kind, H, W), H, W, "random_flip")


def random_rotation(images, p=0.3, *, rot=None):
    """src/augmentations.py:171-213: with probability p a sample is turned by randint(1,4) quarter turns
    (torch.rot90 over the two spatial dims); rot (ints in 0..3) overrides the draw."""
    _check_bchw(images, "random_rotation")
    B, _, H, W = images.shape
    if H != W:
        raise ValueError("random_rotation: square frames expected")
    if rot is None:
        rnd = np.random.uniform(0.0, 1.0, size=(B,))
        rnd_rot = np.random.randint(1, 4, size=(B,))
        rot = rnd_rot * (rnd <= p)
    rot = np.asarray(rot).reshape(B)
    if B and (rot.min() < 0 or rot.max() > 3):
        raise ValueError("random_rotation: rot must be in 0..3")
    return _affine_gather(images, _kind_coef(rot, H, W), H, W, "random_rotation")


def random_cutout_color(imgs, min_cut=10, max_cut=30, *, w1=None, h1=None, color=None):
    """src/augmentations.py:327-350: rows h1..2*h1-1, cols w1..2*w1-1 of every channel take a per-(sample,channel)
    colour randint(0,255)/255.  The reference computes this in numpy and returns a HOST tensor; this one stays on
    the device (its callers move the result there anyway)."""
    _check_bchw(imgs, "random_cutout_color")
    if imgs.dtype != torch.float32:
        raise TypeError("random_cutout_color: float32 input expected")
    n, c, h, w = imgs.shape
    if w1 is None:
        w1 = np.random.randint(min_cut, max_cut, n)
    if h1 is None:
        h1 = np.random.randint(min_cut, max_cut, n)
    if color is None:
        color = np.random.randint(0, 255, size=(n, c)) / 255.0
    box = np.stack([np.asarray(h1).reshape(n), np.asarray(w1).reshape(n)], axis=1).astype(np.int32)
    if n and box.min() < 0:
        raise ValueError("random_cutout_color: negative box origin")
    color = np.ascontiguousarray(color, dtype=np.float32).reshape(n, c)
    imgs = imgs.detach().contiguous()
    box_t = torch.from_numpy(np.ascontiguousarray(box)).to(imgs.device, non_blocking=True)
    col_t = torch.from_numpy(color).to(imgs.device, non_blocking=True)
    out = torch.empty_like(imgs)
    _lib.call("augcl_cutout_color_f32", _lib.ptr(imgs), _lib.ptr(out), n, c, h, w, _lib.ptr(box_t), _lib.ptr(col_t),
              _lib.current_stream(imgs.device))
    return out


def random_grayscale(images, p=0.3, *, mask=None):
    """src/augmentations.py:89-136, literal: t = uint8(images*255) (the cast wraps mod 256 beyond 255, as torch's
    host cast does), the samples with np.random.uniform <= p take per-frame gray = uint8(.2989r+.587g+.114b);
    returns t/255.  Written for [0,1] frames; on 0..255 floats the wrap makes it the reference's own garbage."""
    _check_bchw(images, "random_grayscale")
    if images.dtype != torch.float32:
        raise TypeError("random_grayscale: float32 input expected")
    B, C, H, W = images.shape
    if C % 3:
        raise ValueError("random_grayscale: channel count must be a multiple of 3")
    if mask is None:
        mask = np.random.uniform(0.0, 1.0, size=(B,)) <= p
    mask_t = torch.from_numpy(np.ascontiguousarray(np.asarray(mask).reshape(B), dtype=np.uint8)).to(
        images.device, non_blocking=True)
    images = images.contiguous()
    out = torch.empty_like(images)
    _lib.call("augcl_random_grayscale_f32", _lib.ptr(images), _lib.ptr(out), B, C // 3, H, W, _lib.ptr(mask_t),
              _lib.current_stream(images.device))
    return out


KORNIA_HSV_EPS = 1e-6  # kornia.color.rgb_to_hsv default


def _splice(x, name, cs_mode, hue_thres, sat_thres, val_thres, imgs):
    _check_bchw(x, name)
    if x.dtype != torch.float32:
        raise TypeError(f"{name}: float32 input expected")
    n, c, h, w = x.shape
    if c % 3:
        raise ValueError(f"{name}: channel count must be a multiple of 3")
    if imgs is None:
        if _overlay_source is None:
            raise FileNotFoundError("failed to find image data: no overlay source registered "
                                    "(augmentations.set_overlay_source)")
        imgs = _overlay_source(n, w, x.device)
    imgs = torch.as_tensor(imgs, device=x.device, dtype=torch.float32).contiguous()
    if tuple(imgs.shape) != (n, 3, h, w):
        raise ValueError(f"{name}: overlay batch must be {(n, 3, h, w)}, got {tuple(imgs.shape)}")
    x = x.contiguous()
    out = torch.empty_like(x)
    _lib.call("augcl_splice_f32", _lib.ptr(x), _lib.ptr(imgs), _lib.ptr(out), n, c // 3, h, w, int(cs_mode),
              float(hue_thres), float(sat_thres), float(val_thres), KORNIA_HSV_EPS, _lib.current_stream(x.device))
    return out


def splice(x, hue_thres=0, sat_thres=0, val_thres=0.6, *, imgs=None):
    """src/augmentations.py:298-309: pixels of x/255 whose (hue, saturation, value) all exceed the thresholds are
    kept, the rest is replaced by an overlay image; *255.  (0.6 val for video_hard, 0.4 for video_easy.)
    Frame j of the (n*frames,3,h,w) view pairs with overlay image j % n, as the reference's tiling does."""
    return _splice(x, "splice", 0, hue_thres, sat_thres, val_thres, imgs)


def CS_splice(x, hue_thres=3.5, sat_thres=0, val_thres=0, *, imgs=None):
    """src/augmentations.py:312-324 (cartpole variant): overlay everywhere the first mask holds, except where
    the (hue>0, sat, value>0.6) mask holds too."""
    return _splice(x, "CS_splice", 1, hue_thres, sat_thres, val_thres, imgs)


def batch_from_obs(obs, batch_size=32):
    """src/augmentations.py:353-362: copy a single observation along the batch dimension."""
    if isinstance(obs, torch.Tensor):
        if len(obs.shape) == 3:
            obs = obs.unsqueeze(0)
        return obs.repeat(batch_size, 1, 1, 1)
    if len(obs.shape) == 3:
        obs = np.expand_dims(obs, axis=0)
    return np.repeat(obs, repeats=batch_size, axis=0)


def prepare_pad_batch(obs, next_obs, action, batch_size=32, *, device="cuda"):
    """src/augmentations.py:365-371 (PAD test-time adaptation batch): numpy obs/next_obs/action -> device, tiled
    batch_size times, obs and next_obs independently random-cropped to 84."""
    batch_obs = batch_from_obs(torch.from_numpy(obs).to(device), batch_size)
    batch_next_obs = batch_from_obs(torch.from_numpy(next_obs).to(device), batch_size)
    batch_action = torch.from_numpy(action).to(device).unsqueeze(0).repeat(batch_size, 1)
    f = lambda t: t if t.dtype == torch.float32 else t.float()  # noqa: E731  (the gather kernel is f32)
    return random_crop(f(batch_obs)), random_crop(f(batch_next_obs)), batch_action


aug_to_func = {
    "grayscale": random_grayscale,
    "flip": random_flip,
    "rotate": random_rotation,
    "crop": random_crop,
    "cutout_color": random_cutout_color,
    "splice": splice,
    "CS_splice": CS_splice,
    "conv": random_conv,
    "shift": random_shift,
    "color_jitter": color_jitter,
    "identity": identity,
    "overlay": random_overlay,
}
