"""CPU: the C-ABI library builds, loads and exports every symbol include/augcl.h declares.
No compute call is made (there is no GPU here)."""
import ctypes

from augcl_amd import _lib


def test_header_and_binding_agree():
    declared = set(_lib.declared_symbols())
    assert declared, "include/augcl.h declares nothing?"
    assert declared == set(_lib._SIGS), (declared - set(_lib._SIGS), set(_lib._SIGS) - declared)


def test_library_exports_every_declared_symbol():
    _lib.build()
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in _lib.declared_symbols():
        assert hasattr(lib, name), f"libaugcl_hip.so lacks {name}"
    assert _lib.load().augcl_abi_version() == 1


def test_bad_arguments_are_reported_not_crashed():
    lib = _lib.load()
    rc = lib.augcl_random_shift_f32(None, None, 1, 9, 84, 84, 4, None, None, 0, 0, 0, None)
    assert rc == -1 and b"null" in lib.augcl_last_error()
    rc = lib.augcl_random_conv_f32(None, None, 1, 3, 84, 84, None, 0, 0, 0, None)
    assert rc == -1
    order = (ctypes.c_int32 * 4)(1, 1, 2, 3)
    rc = lib.augcl_color_jitter_f32(1 << 20, 1, 3, 84, 84, None, 0, ctypes.cast(order, ctypes.c_void_p), 1.0, 1.0,
                                    0.0, 1.0, 1.0, None)
    assert rc == -1 and b"permutation" in lib.augcl_last_error()


def test_ops_fail_loudly_without_gpu_tensor():
    import pytest
    import torch

    from augcl_amd import augmentations as aug
    x = torch.zeros(1, 9, 84, 84)
    with pytest.raises(_lib.AugclError):
        aug.random_shift(x)
    with pytest.raises(_lib.AugclError):
        aug.random_conv(x)
