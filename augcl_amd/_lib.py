"""ctypes binding of libaugcl_hip.so (C ABI: include/augcl.h).

The HIP library is THE product path.  There is no CPU fallback: if the shared
object is missing, lacks a symbol, or no GPU is visible when an op is called,
we raise -- loudly -- instead of computing the result some other way.
"""
import ctypes
import os
import re
import subprocess

_PKG = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_PKG)
LIB_PATH = os.path.join(_PKG, "libaugcl_hip.so")
HEADER_PATH = os.path.join(_ROOT, "include", "augcl.h")

AUGCL_OK = 0
AUGCL_E_UNSUPPORTED = -3  # include/augcl.h


class AugclError(RuntimeError):
    pass


def build(force=False, verbose=False):
    """Compile libaugcl_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU)."""
    csrc = os.path.join(_PKG, "csrc")
    cmd = ["make", "-j4", "-C", csrc] + (["-B"] if force else [])  # hipcc per file: 4 jobs fit the build container
    r = subprocess.run(cmd, capture_output=True, text=True)
    if verbose or r.returncode != 0:
        print(r.stdout + r.stderr)
    if r.returncode != 0:
        raise AugclError("building libaugcl_hip.so failed")
    return LIB_PATH


def source_hash():
    """16 hex digits over every source the shared object is built from (csrc/*, include/augcl.h): profiles/*.json carry
    the hash of the tree whose binary produced their counters, and bench.py only quotes counters whose hash is this one."""
    import glob
    import hashlib
    h = hashlib.sha256()
    csrc = os.path.join(_PKG, "csrc")
    files = sorted(f for f in glob.glob(os.path.join(csrc, "*")) if f.endswith((".hip", ".cpp", ".h", "Makefile")))
    for f in files + [HEADER_PATH]:
        h.update(os.path.basename(f).encode() + b"\0")
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def declared_symbols():
    """Every function name include/augcl.h declares (used by the symbol-export test)."""
    txt = open(HEADER_PATH).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(augcl_[a-z0-9_]+)\s*\(", txt)))


_c = ctypes
_P = _c.c_void_p
_I = _c.c_int
_U64 = _c.c_uint64
_I64 = _c.c_int64
_SZ = _c.c_size_t
_F = _c.c_float
_D = _c.c_double

# name -> argtypes (restype is int unless listed in _RESTYPE)
_SIGS = {
    "augcl_abi_version": [],
    "augcl_last_error": [],
    "augcl_device_count": [],
    "augcl_philox_shift_offsets": [_U64, _U64, _I, _I, _I, _P, _P, _P],
    "augcl_philox_conv_weights": [_U64, _U64, _I, _I, _P, _P],
    "augcl_philox_normal": [_U64, _U64, _SZ, _P, _P],
    "augcl_philox_replay_indices": [_U64, _U64, _I, _I, _P, _I, _P],
    "augcl_random_shift_f32": [_P, _P, _I, _I, _I, _I, _I, _P, _P, _U64, _U64, _I, _P],
    "augcl_random_shift_u8": [_P, _P, _I, _I, _I, _I, _I, _P, _P, _U64, _U64, _I, _P],
    "augcl_random_shift_u8_f32": [_P, _P, _I, _I, _I, _I, _I, _P, _P, _U64, _U64, _I, _P],
    "augcl_random_conv_f32": [_P, _P, _I, _I, _I, _I, _P, _U64, _U64, _I, _P],
    "augcl_random_overlay_f32": [_P, _P, _P, _I, _I, _I, _I, _F, _P],
    "augcl_color_jitter_f32": [_P, _I, _I, _I, _I, _P, _I, _P, _D, _D, _D, _F, _F, _P],
    "augcl_shift_conv_u8_f32": [_P, _P, _P, _I, _I, _I, _I, _I, _P, _P, _P, _U64, _U64, _I, _P],
    "augcl_shift_conv_f32": [_P, _P, _P, _I, _I, _I, _I, _I, _P, _P, _P, _U64, _U64, _I, _P],
    "augcl_replay_shift_conv_u8_f32": [_P, _P, _P, _I, _P, _P, _I, _I, _I, _I, _I, _P, _P, _P, _U64, _U64, _I, _P],
    "augcl_affine_gather_f32": [_P, _P, _I, _I, _I, _I, _I, _I, _P, _P],
    "augcl_cutout_color_f32": [_P, _P, _I, _I, _I, _I, _P, _P, _P],
    "augcl_random_grayscale_f32": [_P, _P, _I, _I, _I, _I, _P, _P],
    "augcl_splice_f32": [_P, _P, _P, _I, _I, _I, _I, _I, _F, _F, _F, _F, _P],
    "augcl_replay_sample": [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P],
    "augcl_green_screen_u8": [_P, _P, _P, _I, _I, _I, _P],
    "augcl_interpolate_bg_u8": [_P, _P, _I, _I, _I, _I, _I, _P],
    "augcl_green_screen_resize_u8": [_P, _P, _P, _I, _I, _I, _I, _I, _P],
    "augcl_gauss_head_forward": [_P, _P, _P, _P, _P, _P, _I, _I, _D, _D, _P],
    "augcl_gauss_head_backward": [_P, _P, _P, _P, _I, _P, _P, _P, _I, _I, _D, _D, _P],
    "augcl_ln_tanh_forward": [_P, _P, _P, _P, _I, _I, _D, _P],
    "augcl_ln_tanh_backward": [_P, _P, _P, _P, _I, _P, _P, _P, _P, _P, _I, _I, _D, _P],
    "augcl_proj_ln_tanh_workspace_slabs": [_I, _I, _I],
    "augcl_proj_ln_tanh_forward": [_P, _P, _P, _P, _P, _P, _P, _P, _SZ, _I, _I, _I, _D, _P],
    "augcl_proj_ln_tanh_cat_forward": [_P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _P, _SZ, _I, _I, _I, _D, _P],
    "augcl_td_target_f32": [_P, _P, _P, _P, _P, _P, _I, _D, _P, _I, _P],
    "augcl_actor_loss_forward": [_P, _P, _P, _P, _P, _I, _P],
    "augcl_actor_loss_backward": [_P, _P, _P, _P, _P, _P, _P, _I, _P],
    "augcl_alpha_step_f64": [_P, _P, _I, _D, _D, _D, _D, _D, _P, _P, _I, _P],
    "augcl_q_loss_forward": [_P, _P, _P, _I, _I, _P, _P, _P],
    "augcl_q_loss_backward": [_P, _P, _P, _P, _I, _I, _P, _P, _P, _P],
    "augcl_adam_f32": [_P, _P, _P, _P, _SZ, _D, _D, _D, _D, _c.c_int64, _D, _P],
    "augcl_adam_f32_devstep": [_P, _P, _P, _P, _SZ, _D, _D, _D, _D, _P, _I, _D, _P],
    "augcl_gather_f32": [_P, _P, _P, _I, _P, _P, _D, _D, _D, _P],
    "augcl_ema_f32": [_P, _P, _SZ, _D, _P],
    "augcl_ema2_f32": [_P, _P, _SZ, _D, _P, _P, _SZ, _D, _P],
    "augcl_mlp_out_forward": [_P, _P, _I64, _P, _I64, _P, _I64, _P, _I, _I, _I, _P],
    "augcl_mlp_out_backward": [_P, _P, _P, _P, _I64, _P, _P, _P, _P, _I, _I, _I, _P],
    "augcl_mask_colsum": [_P, _P, _P, _P, _P, _I, _I, _I, _P],
    "augcl_proj_dx_f32": [_P, _P, _P, _I, _I, _I, _P],
    "augcl_enc_bf16_conv_first_forward": [_P, _P, _P, _P, _I, _I, _I, _I, _I, _F, _I, _P],
    "augcl_enc_bf16_conv32_forward": [_P, _P, _P, _P, _I, _I, _I, _I, _I, _P],
    "augcl_enc_bf16_conv32_backward_data": [_P, _P, _P, _P, _I, _I, _I, _P],
    "augcl_enc_bf16_conv32_backward_weights": [_P, _P, _P, _P, _P, _SZ, _I, _I, _I, _P],
    "augcl_enc_bf16_conv32_backward_weights_mfma": [_P, _P, _P, _P, _P, _SZ, _I, _I, _I, _P],
    "augcl_enc_bf16_conv_first_backward_weights": [_P, _P, _P, _P, _P, _SZ, _I, _I, _I, _I, _I, _F, _P],
    "augcl_enc_nchw_f32_to_nhwc_bf16": [_P, _P, _I, _I, _I, _P],
    "augcl_enc_amax_f32": [_P, _SZ, _P, _P],
    "augcl_enc_x2_conv32_forward": [_P, _P, _P, _P, _I, _I, _I, _I, _P, _SZ, _P, _P, _P],
    "augcl_enc_x2_conv32_backward_data": [_P, _P, _P, _P, _I, _I, _I, _P, _SZ, _P, _P, _P],
    "augcl_enc_x2_conv32_backward_weights": [_P, _P, _P, _P, _P, _SZ, _I, _I, _I, _P, _P, _P],
    "augcl_enc_stack_forward": [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _F, _I, _P, _SZ, _P, _P, _P],
    "augcl_enc_stack_forward_chained": [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _F, _P, _SZ, _P, _P, _P],
    "augcl_enc_frag_workspace_bytes": [_I, _I],
    "augcl_enc_stack_backward": [_P, _P, _P, _P, _P, _P, _P, _P, _SZ, _P, _SZ, _I, _I, _I, _I, _I, _I, _F, _I, _I, _P, _SZ, _P, _P, _P, _P,
                                 _SZ, _P],
    "augcl_enc_stack_backward_chain_bytes": [_I, _I, _I, _I, _I, _I],
    "augcl_enc_stack_forward_dual": [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _F, _I, _P, _SZ, _P, _P],
    "augcl_enc_stack_infer_scratch_bytes": [_I, _I, _I, _I, _I, _I],
    "augcl_enc_stack_infer": [_P, _P, _P, _P, _P, _P, _P, _SZ, _I, _I, _I, _I, _I, _I, _F, _I, _P, _SZ, _P, _P],
    "augcl_enc_stack_infer_u8": [_P, _P, _P, _P, _P, _P, _P, _SZ, _I, _I, _I, _I, _I, _I, _F, _I, _P, _SZ, _P, _P],
    "augcl_enc_conv_first_forward_u8": [_P, _P, _P, _P, _I, _I, _I, _I, _I, _F, _I, _P],
    "augcl_replay_gather": [_P, _P, _P, _I, _I, _I, _I, _P],
    "augcl_enc_conv_first_forward": [_P, _P, _P, _P, _I, _I, _I, _I, _I, _F, _I, _P],
    "augcl_enc_conv32_forward": [_P, _P, _P, _P, _I, _I, _I, _I, _P],
    "augcl_enc_conv32_backward_data": [_P, _P, _P, _P, _I, _I, _I, _P],
    "augcl_enc_wgrad_workspace_bytes": [_I],
    "augcl_enc_x3_conv32_forward": [_P, _P, _P, _P, _I, _I, _I, _I, _P],
    "augcl_enc_x3_conv32_backward_data": [_P, _P, _P, _P, _I, _I, _I, _P],
    "augcl_enc_x3_conv32_backward_weights": [_P, _P, _P, _P, _P, _SZ, _I, _I, _I, _P],
    "augcl_enc_conv32_backward_weights": [_P, _P, _P, _P, _P, _SZ, _I, _I, _I, _P],
    "augcl_enc_conv_first_backward_weights": [_P, _P, _P, _P, _P, _SZ, _I, _I, _I, _I, _I, _F, _P],
}
_RESTYPE = {"augcl_last_error": _c.c_char_p, "augcl_enc_wgrad_workspace_bytes": _SZ, "augcl_enc_frag_workspace_bytes": _SZ,
            "augcl_enc_stack_infer_scratch_bytes": _SZ, "augcl_enc_stack_backward_chain_bytes": _SZ}

_lib = None


def load():
    """dlopen the library and bind every declared entry point.  Raises if anything is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise AugclError(
            f"{LIB_PATH} not found: the HIP extension is required (no CPU fallback). "
            "Build it with `python -c 'import __graft_entry__ as g; g.build()'` or `make -C augcl_amd/csrc`.")
    lib = ctypes.CDLL(LIB_PATH)
    for name, argtypes in _SIGS.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise AugclError(f"libaugcl_hip.so does not export {name}") from e
        fn.argtypes = argtypes
        fn.restype = _RESTYPE.get(name, _I)
    ver = lib.augcl_abi_version()
    if ver != 6:
        raise AugclError(f"libaugcl_hip.so ABI version {ver} != 6 (stale build: run `make -C augcl_amd/csrc`)")
    _lib = lib
    return lib


def check(status, what=""):
    if status != AUGCL_OK:
        msg = load().augcl_last_error()
        raise AugclError(f"{what} failed ({status}): {msg.decode() if msg else '?'}")


def call(name, *args):
    lib = load()
    check(getattr(lib, name)(*args), name)


def ptr(t):
    """Device pointer of a torch tensor (or None)."""
    return None if t is None else t.data_ptr()


_raw_stream = None


def current_stream(device=None):
    """Raw handle (hipStream_t as an integer) of torch's current stream on `device`: what every entry point takes.
    torch._C._cuda_getCurrentRawStream when it exists (no Stream object per call: this sits on the launch path of every
    kernel, and the bf16 mode is host-bound when it runs without hipGraphs)."""
    global _raw_stream
    import torch
    if _raw_stream is None:
        _raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", False)
    if _raw_stream:
        idx = device.index if (device is not None and hasattr(device, "index") and device.index is not None) else torch.cuda.current_device()
        return _raw_stream(idx)
    return torch.cuda.current_stream(device).cuda_stream


def require_cuda(t, name="tensor"):
    if not t.is_cuda:
        raise AugclError(f"{name} must live on the GPU: augcl_amd has no CPU path (got device {t.device})")
    if not t.is_contiguous():
        raise AugclError(f"{name} must be contiguous")
