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
