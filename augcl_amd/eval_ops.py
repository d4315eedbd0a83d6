"""Eval-side pixel operators: the DMControl-GB `video_easy` / `video_hard` wrappers key the
simulator's green background out of every rendered frame with a per-pixel Python loop
(src/env/wrappers.py:209-260, "not optimized for speed").  Same function, one HIP launch, uint8-exact."""
import numpy as np
import torch

from . import _lib


def green_screen(x, bg, out=None):
    """Batched, device-resident: x, bg (N,3,H,W) or (3,H,W) uint8 CUDA tensors -> same shape uint8."""
    _lib.require_cuda(x, "green_screen")
    if x.dtype != torch.uint8 or bg.dtype != torch.uint8 or x.shape != bg.shape:
        raise TypeError("green_screen: x and bg must be uint8 tensors of the same shape")
    shape = x.shape
    x4 = x.contiguous().view(-1, 3, shape[-2], shape[-1])
    bg4 = bg.to(x.device).contiguous().view(-1, 3, shape[-2], shape[-1])
    if out is None:
        out = torch.empty_like(x4)
    _lib.call("augcl_green_screen_u8", x4.data_ptr(), bg4.data_ptr(), out.data_ptr(), x4.shape[0], shape[-2], shape[-1],
              _lib.current_stream(x.device))
    return out.view(shape)


def do_green_screen(x, bg, device=None):
    """Drop-in for wrappers.do_green_screen(x, bg): numpy uint8 (3,H,W) in, numpy uint8 (3,H,W) out."""
    assert isinstance(x, np.ndarray) and isinstance(bg, np.ndarray), "inputs must be numpy arrays"
    assert x.dtype == np.uint8 and bg.dtype == np.uint8, "inputs must be uint8 arrays"
    if device is None:
        if not torch.cuda.is_available():
            raise _lib.AugclError("do_green_screen runs on the GPU: no CPU path")
        device = torch.device("cuda", torch.cuda.current_device())
    xt = torch.from_numpy(np.ascontiguousarray(x)).to(device)
    bt = torch.from_numpy(np.ascontiguousarray(bg)).to(device)
    return green_screen(xt, bt).cpu().numpy()


def interpolate_bg(bg, size):
    """VideoWrapper._interpolate_bg (src/env/wrappers.py:336-340) on the device: (3,Hb,Wb) or (N,3,Hb,Wb) uint8 CUDA tensor
    -> the same with spatial size `size`, uint8-exact (float/255, bilinear align_corners=False, *255, truncating cast)."""
    _lib.require_cuda(bg, "interpolate_bg")
    if bg.dtype != torch.uint8:
        raise TypeError("interpolate_bg: uint8 frames expected")
    H, W = int(size[0]), int(size[1])
    b4 = bg.contiguous().view(-1, 3, bg.shape[-2], bg.shape[-1])
    out = torch.empty((b4.shape[0], 3, H, W), dtype=torch.uint8, device=bg.device)
    _lib.call("augcl_interpolate_bg_u8", b4.data_ptr(), out.data_ptr(), b4.shape[0], bg.shape[-2], bg.shape[-1], H, W,
              _lib.current_stream(bg.device))
    return out.view(*bg.shape[:-2], H, W)


def greenscreen(obs, bg_raw, out=None):
    """VideoWrapper._greenscreen (src/env/wrappers.py:342-348) in ONE launch: the raw video frame(s) `bg_raw`
    ((N,)3,Hb,Wb uint8) are resized to the observation's size only where the green key lets them through."""
    _lib.require_cuda(obs, "greenscreen")
    if obs.dtype != torch.uint8 or bg_raw.dtype != torch.uint8:
        raise TypeError("greenscreen: uint8 tensors expected")
    shape = obs.shape
    x4 = obs.contiguous().view(-1, 3, shape[-2], shape[-1])
    b4 = bg_raw.to(obs.device).contiguous().view(-1, 3, bg_raw.shape[-2], bg_raw.shape[-1])
    if b4.shape[0] != x4.shape[0]:
        raise ValueError("greenscreen: one background frame per observation frame")
    if out is None:
        out = torch.empty_like(x4)
    _lib.call("augcl_green_screen_resize_u8", x4.data_ptr(), b4.data_ptr(), out.data_ptr(), x4.shape[0], shape[-2], shape[-1],
              b4.shape[-2], b4.shape[-1], _lib.current_stream(obs.device))
    return out.view(shape)
