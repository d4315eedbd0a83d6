"""oracle/aug_oracle.py -- numpy restatement of AugCL's augmentation operators
plus a ctypes view of the C restatement (oracle/aug_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py.  Nothing under augcl_amd/ imports this module.

Every function takes the random parameters explicitly; none touches an RNG.
Pinning status per operator is stated in aug_oracle.c's header and DESIGN.md.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liboracle_aug.so")
_lib = None

_f32p = ctypes.POINTER(ctypes.c_float)
_u8p = ctypes.POINTER(ctypes.c_uint8)
_i32p = ctypes.POINTER(ctypes.c_int32)


def build(force=False):
    """Compile the C checker with gcc (no GPU, no reference needed)."""
    src = os.path.join(_HERE, "aug_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "liboracle_aug.so"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_LIB_PATH)
    return _lib


def set_num_threads(n):
    lib().oracle_set_num_threads(int(n))
    return lib().oracle_get_max_threads()


def _p(a, ty):
    return a.ctypes.data_as(ty)


# --------------------------------------------------------------------------- numpy
def random_shift_np(x, oy, ox, pad=4):
    """src/augmentations.py:378-382: replicate pad + per-sample integer crop.
    x: (B,C,H,W) any dtype; oy/ox: (B,) ints in [0, 2*pad]."""
    B, C, H, W = x.shape
    ys = np.clip(np.arange(H)[None, :] + np.asarray(oy)[:, None] - pad, 0, H - 1)  # (B,H)
    xs = np.clip(np.arange(W)[None, :] + np.asarray(ox)[:, None] - pad, 0, W - 1)  # (B,W)
    b = np.arange(B)[:, None, None, None]
    c = np.arange(C)[None, :, None, None]
    return x[b, c, ys[:, None, :, None], xs[:, None, None, :]]


def random_conv_np(x, weights):
    """src/augmentations.py:265-274. x (B,C,H,W) f32 in 0..255; weights (B,3,3,3,3)."""
    B, C, H, W = x.shape
    fr = C // 3
    xn = (x.reshape(B, fr, 3, H, W) / np.float32(255.0)).astype(np.float32)
    xp = np.pad(xn, ((0, 0), (0, 0), (0, 0), (1, 1), (1, 1)), mode="edge")
    w = np.asarray(weights, dtype=np.float32).reshape(B, 3, 3, 3, 3)
    acc = np.zeros((B, fr, 3, H, W), dtype=np.float32)
    for i in range(3):
        for ky in range(3):
            for kx in range(3):
                win = xp[:, :, i, ky:ky + H, kx:kx + W]  # (B,fr,H,W)
                acc += w[:, None, :, i, ky, kx, None, None] * win[:, :, None]
    one = np.float32(1.0)
    out = one / (one + np.exp(-acc, dtype=np.float32))
    return (out * np.float32(255.0)).reshape(B, C, H, W)


def random_overlay_np(x, overlay, alpha=0.5):
    """src/augmentations.py:253-262, literal op order (bit-exact vs torch f32)."""
    B, C, H, W = x.shape
    imgs = np.tile(np.asarray(overlay, dtype=np.float32), (1, C // 3, 1, 1))
    a = np.float32(alpha)
    om = np.float32(1.0 - alpha)
    return ((om * (x / np.float32(255.0))) + a * imgs) * np.float32(255.0)


# --------------------------------------------------------------------------- C
def random_shift_c(x, oy, ox, pad=4, out_dtype=None):
    x = np.ascontiguousarray(x)
    B, C, H, W = x.shape
    oy = np.ascontiguousarray(oy, dtype=np.int32)
    ox = np.ascontiguousarray(ox, dtype=np.int32)
    L = lib()
    if x.dtype == np.float32:
        out = np.empty_like(x)
        L.oracle_random_shift_f32(_p(x, _f32p), _p(out, _f32p), B, C, H, W, pad, _p(oy, _i32p), _p(ox, _i32p))
    elif x.dtype == np.uint8 and out_dtype in (None, np.uint8):
        out = np.empty_like(x)
        L.oracle_random_shift_u8(_p(x, _u8p), _p(out, _u8p), B, C, H, W, pad, _p(oy, _i32p), _p(ox, _i32p))
    elif x.dtype == np.uint8:
        out = np.empty(x.shape, dtype=np.float32)
        L.oracle_random_shift_u8_f32(_p(x, _u8p), _p(out, _f32p), B, C, H, W, pad, _p(oy, _i32p), _p(ox, _i32p))
    else:
        raise TypeError(x.dtype)
    return out


def random_conv_c(x, weights):
    x = np.ascontiguousarray(x, dtype=np.float32)
    B, C, H, W = x.shape
    w = np.ascontiguousarray(weights, dtype=np.float32).reshape(B, 81)
    out = np.empty_like(x)
    lib().oracle_random_conv_f32(_p(x, _f32p), _p(out, _f32p), B, C // 3, H, W, _p(w, _f32p))
    return out


def random_overlay_c(x, overlay, alpha=0.5):
    x = np.ascontiguousarray(x, dtype=np.float32)
    B, C, H, W = x.shape
    ov = np.ascontiguousarray(overlay, dtype=np.float32)
    assert ov.shape == (B, 3, H, W)
    out = np.empty_like(x)
    lib().oracle_random_overlay_f32(_p(x, _f32p), _p(ov, _f32p), _p(out, _f32p), B, C // 3, H, W,
                                    ctypes.c_float(alpha))
    return out


def color_jitter_c(x, idx, order, contrast, saturation, hue, bound=1.0, scale=1.0):
    """In-place on a copy; returns the jittered (B,C,H,W) array.
    idx indexes the (B*frames,3,H,W) view (src/augmentations.py:233-236)."""
    out = np.array(x, dtype=np.float32, order="C", copy=True)
    B, C, H, W = out.shape
    idx = np.ascontiguousarray(idx, dtype=np.int32)
    order = np.ascontiguousarray(order, dtype=np.int32)
    assert order.shape == (4,)
    lib().oracle_color_jitter_f32(_p(out, _f32p), B, C // 3, H, W, _p(idx, _i32p), int(idx.size), _p(order, _i32p),
                                  ctypes.c_double(contrast), ctypes.c_double(saturation), ctypes.c_double(hue),
                                  ctypes.c_float(bound), ctypes.c_float(scale))
    return out


# ---------------------------------------------------------------- rest of the registry: numpy
# index / select restatements, explicit random parameters.  Bit-exact operators.
def affine_coef(kind, n, H=84, W=84, a=None, b=None):
    """int32 (n,6) {ay,by,cy,ax,bx,cx} rows for out[y,x] = in[ay*y+by*x+cy, ax*y+bx*x+cx].
    kind: per-sample array of 0 identity / 1 rot90 / 2 rot180 / 3 rot270 / 4 hflip / 5 crop(a=row0, b=col0)."""
    table = {0: (1, 0, 0, 0, 1, 0), 1: (0, 1, 0, -1, 0, W - 1), 2: (-1, 0, H - 1, 0, -1, W - 1),
             3: (0, -1, H - 1, 1, 0, 0), 4: (1, 0, 0, 0, -1, W - 1)}
    out = np.empty((n, 6), np.int32)
    for i in range(n):
        k = int(kind[i])
        out[i] = (1, 0, int(a[i]), 0, 1, int(b[i])) if k == 5 else table[k]
    return out


def affine_gather_np(x, coef, Hout, Wout):
    """One integer-affine gather covers random_crop / random_flip / random_rotation."""
    x = np.asarray(x)
    B = x.shape[0]
    yy, xx = np.mgrid[0:Hout, 0:Wout]
    out = np.empty(x.shape[:2] + (Hout, Wout), x.dtype)
    for i in range(B):
        ay, by, cy, ax, bx, cx = (int(v) for v in coef[i])
        out[i] = x[i][:, ay * yy + by * xx + cy, ax * yy + bx * xx + cx]
    return out


def random_crop_np(x, w1, h1, size=84):
    """src/augmentations.py:385-433: windows[arange(n), w1, h1] on the NHWC view -> out[i,c,y,x] =
    x[i,c,w1[i]+y,h1[i]+x] (the reference's `w1` walks ROWS, `h1` columns)."""
    x = np.asarray(x)
    return np.stack([x[i, :, int(w1[i]):int(w1[i]) + size, int(h1[i]):int(h1[i]) + size] for i in range(x.shape[0])])


def random_flip_np(x, mask):
    """src/augmentations.py:139-168: out = mask*flip([3]) + (1-mask)*x, one mask bit per sample."""
    x = np.asarray(x)
    out = x.copy()
    m = np.asarray(mask, bool)
    out[m] = x[m][..., ::-1]
    return out


def random_rotation_np(x, rot):
    """src/augmentations.py:171-213: rot[i] in {0,1,2,3} quarter turns (torch.rot90 over dims [2,3])."""
    x = np.asarray(x)
    return np.stack([np.rot90(x[i], int(rot[i]), axes=(1, 2)) for i in range(x.shape[0])])


def random_cutout_color_np(x, w1, h1, color):
    """src/augmentations.py:327-350: img[:, h1:2*h1, w1:2*w1] = color[i][:,None,None] (numpy slicing clips)."""
    out = np.array(x, copy=True)
    for i in range(out.shape[0]):
        h, w = int(h1[i]), int(w1[i])
        out[i, :, h:h + h, w:w + w] = np.asarray(color[i], out.dtype).reshape(-1, 1, 1)
    return out


def random_grayscale_np(x, mask):
    """src/augmentations.py:89-136, literal: uint8(x*255) with C-style wrap (torch CPU cast), per-frame
    gray = uint8(.2989 r + .587 g + .114 b) in f32 left-to-right, out = (mask ? gray : t) / 255."""
    x = np.asarray(x, np.float32)
    B, C, H, W = x.shape
    t = ((x * np.float32(255.0)).astype(np.int64) & 255).astype(np.float32).reshape(B, C // 3, 3, H, W)
    gray = (t[:, :, 0] * np.float32(0.2989) + t[:, :, 1] * np.float32(0.587)) + t[:, :, 2] * np.float32(0.114)
    gray = np.floor(gray).astype(np.float32)  # values in [0,255]: uint8 cast == truncation
    m = np.asarray(mask, bool)[:, None, None, None, None]
    out = np.where(m, gray[:, :, None], t)
    return (out.reshape(B, C, H, W) / np.float32(255.0)).astype(np.float32)


def rgb_to_hsv_np(rgb, eps=1e-6):
    """kornia.color.rgb_to_hsv (kornia is unpinned in the reference's setup/conda.yaml:22; algorithm of 0.5.x
    kornia/color/hsv.py restated: parity unpinned for the arithmetic).  rgb (N,3,H,W) f32 -> h (radians), s, v."""
    f = np.float32
    r, g, b = rgb[:, 0], rgb[:, 1], rgb[:, 2]
    mx = np.maximum(np.maximum(r, g), b)
    mn = np.minimum(np.minimum(r, g), b)
    dc = mx - mn
    s = dc / (mx + f(eps))
    d = np.where(dc == 0, f(1.0), dc)
    rc, gc, bc = mx - r, mx - g, mx - b
    h6 = np.where(r == mx, bc - gc, np.where(g == mx, (f(2.0) * d + rc) - bc, (f(4.0) * d + gc) - rc))
    h = (h6 / d) / f(6.0)
    h = np.fmod(h, f(1.0))
    h = np.where((h != 0) & (h < 0), h + f(1.0), h)
    return f(2 * np.pi) * h, s, mx


def splice_np(x, imgs, hue_thres=0.0, sat_thres=0.0, val_thres=0.6, cs=False, eps=1e-6):
    """src/augmentations.py:298-324 (+ create_hsv_mask L277-285).  cs=True is CS_splice (pass hue_thres=3.5,
    val_thres=0).  Frame j of the (n*frames,3,h,w) view takes overlay image j % n (batch-dim tiling, L302/L316)."""
    x = np.asarray(x, np.float32)
    n, c, hh, ww = x.shape
    rgb = x.reshape(-1, 3, hh, ww) / np.float32(255.0)
    ov = np.tile(np.asarray(imgs, np.float32), (c // 3, 1, 1, 1))
    h, s, v = rgb_to_hsv_np(rgb, eps)
    f = np.float32
    keep = (h > f(hue_thres)) & (s > f(sat_thres)) & (v > f(val_thres))
    if cs:
        keep2 = (h > f(0)) & (s > f(sat_thres)) & (v > f(0.6))
        use = keep2 | ~keep
    else:
        use = keep
    out = np.where(use[:, None], rgb, ov)
    return (out.reshape(n, c, hh, ww) * np.float32(255.0)).astype(np.float32)


def green_screen_np(x, bg):
    """src/env/wrappers.py:209-260 do_green_screen (+ rgb_to_hsv L209-228), vectorised in float64 exactly as the
    reference's per-pixel Python arithmetic: keyed pixels (h*360 in [100,185], s*255 in [80,255], v*255 in
    [70,255]) take the background.  x, bg: (...,3,H,W) uint8."""
    x = np.asarray(x)
    bg = np.asarray(bg)
    r, g, b = (x[..., i, :, :].astype(np.float64) / 255.0 for i in range(3))
    maxc = np.maximum(np.maximum(r, g), b)
    minc = np.minimum(np.minimum(r, g), b)
    eq = minc == maxc
    d = np.where(eq, 1.0, maxc - minc)
    s = np.where(eq, 0.0, (maxc - minc) / np.where(maxc == 0, 1.0, maxc))
    rc, gc, bc = (maxc - r) / d, (maxc - g) / d, (maxc - b) / d
    h = np.where(r == maxc, bc - gc, np.where(g == maxc, 2.0 + rc - bc, 4.0 + gc - rc))
    h = np.where(eq, 0.0, (h / 6.0) % 1.0)
    hh, ss, vv = h * 360, s * 255, maxc * 255
    key = (100 <= hh) & (hh <= 185) & (80 <= ss) & (ss <= 255) & (70 <= vv) & (vv <= 255)
    return np.where(key[..., None, :, :], bg, x).astype(np.uint8)


def interpolate_bg_np(bg, size):
    """VideoWrapper._interpolate_bg (src/env/wrappers.py:336-340): uint8 (...,3,Hb,Wb) -> uint8 (...,3,H,W) through
    float32 / 255, F.interpolate(bilinear, align_corners=False), * 255 and a TRUNCATING uint8 cast.  The cast makes the
    result sensitive to the last bit, so this restates ATen's float32 arithmetic operation by operation (pinned against
    the reference's function on this torch build, tests/golden/eval_golden.npz): source coordinate
    fma(scale, dst + 0.5, -0.5) clamped at 0 with scale = float32(in) / float32(out); weights (1 - l, l);
    w0 * a + w1 * b as fma(w0, a, w1 * b), horizontal pass first.  (A float32 fma = the float64 product-sum rounded
    once: exact, because a float32 product fits a float64 and the sum of two such terms rounds correctly here.)"""
    f = np.float32
    bg = np.asarray(bg)
    Hb, Wb = bg.shape[-2:]
    H, W = int(size[0]), int(size[1])
    fma = lambda a, b, c: (np.asarray(a, np.float64) * np.asarray(b, np.float64) + np.asarray(c, np.float64)).astype(f)  # noqa: E731

    def axis(n_in, n_out):
        d = np.arange(n_out)
        if n_in == n_out:
            return d, d, np.ones(n_out, f), np.zeros(n_out, f)
        scale = f(n_in) / f(n_out)
        src = np.maximum(fma(np.full(n_out, scale, f), (d.astype(f) + f(0.5)).astype(f), np.full(n_out, -0.5, f)), f(0)).astype(f)
        i0 = np.minimum(np.floor(src).astype(np.int64), n_in - 1)
        i1 = np.minimum(i0 + 1, n_in - 1)
        l1 = np.clip((src - i0.astype(f)).astype(f), 0, 1).astype(f)
        return i0, i1, (f(1) - l1).astype(f), l1

    y0, y1, hy0, hy1 = axis(Hb, H)
    x0, x1, wx0, wx1 = axis(Wb, W)
    x = (bg.astype(f) / f(255.0)).astype(f)
    rows0, rows1 = x[..., y0, :], x[..., y1, :]
    top = fma(wx0, rows0[..., x0], (wx1 * rows0[..., x1]).astype(f))
    bot = fma(wx0, rows1[..., x0], (wx1 * rows1[..., x1]).astype(f))
    out = fma(hy0[:, None], top, (hy1[:, None] * bot).astype(f))
    return (out * f(255.0)).astype(np.uint8)
