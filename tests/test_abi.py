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
    assert _lib.load().augcl_abi_version() == 4


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
    # round-2 entry points: argument checks come before any launch, so they answer without a GPU too
    assert lib.augcl_replay_shift_conv_u8_f32(None, None, None, 6, None, None, 0, 3, 84, 84, 4, None, None, None, 0, 0, 0, None) == 0  # B = 0
    assert lib.augcl_replay_shift_conv_u8_f32(None, None, None, 6, None, None, 2, 3, 84, 84, 4, None, None, None, 0, 0, 0, None) == -1
    assert lib.augcl_replay_shift_conv_u8_f32(16, None, None, 6, 16, 16, 2, 3, 84, 84, 4, None, None, None, 0, 0, 0, None) == -1
    assert b"slot table" in lib.augcl_last_error()
    assert lib.augcl_shift_conv_u8_f32(16, 16, 16, 2, 3, 50, 50, 4, None, None, None, 0, 0, 0, None) == -1 and b"84x84" in lib.augcl_last_error()
    assert lib.augcl_proj_ln_tanh_workspace_slabs(512, 14112, 100) == 63 and lib.augcl_proj_ln_tanh_workspace_slabs(1024, 14112, 100) == 21
    assert lib.augcl_proj_ln_tanh_workspace_slabs(8, 14112, 300) == 0 and lib.augcl_proj_ln_tanh_workspace_slabs(8, 1000, 100) == 0
    assert lib.augcl_proj_ln_tanh_forward(None, None, None, None, None, None, None, None, 0, 8, 1000, 100, 1e-5, None) == -1
    assert b"not covered" in lib.augcl_last_error()
    assert lib.augcl_proj_ln_tanh_cat_forward(None, None, None, None, None, None, None, None, None, 6, 4, None, 0, 8, 14112, 100, 1e-5, None) == -1
    assert lib.augcl_interpolate_bg_u8(None, None, 0, 240, 320, 84, 84, None) == 0 and lib.augcl_interpolate_bg_u8(None, None, 1, 240, 320, 84, 84, None) == -1
    assert lib.augcl_green_screen_resize_u8(None, None, None, 1, 84, 84, 0, 320, None) == -1
    assert lib.augcl_alpha_step_f64(None, None, 4, -6.0, 1e-4, 0.5, 0.999, 1e-8, None, None, 3, None) == -1
    assert lib.augcl_actor_loss_forward(None, None, None, None, None, 4, None) == -1
    assert lib.augcl_mlp_out_forward(None, None, 0, None, 0, None, 0, None, 2, 0, 1024, None) == 0  # empty batch
    assert lib.augcl_mlp_out_forward(None, None, 0, None, 0, None, 0, None, 2, 8, 1024, None) == -1
    assert lib.augcl_mlp_out_backward(None, None, None, None, 0, None, None, None, None, 3, 8, 1024, None) == -1
    assert b"T = 1 or 2" in lib.augcl_last_error()
    assert lib.augcl_mask_colsum(None, None, None, None, None, 8, 16, 0, None) == -1
    assert lib.augcl_proj_dx_f32(None, None, None, 8, 102, 14112, None) == -1 and b"not covered" in lib.augcl_last_error()
    assert lib.augcl_ema2_f32(None, None, 0, 0.5, None, None, 0, 0.5, None) == 0
    assert lib.augcl_ema2_f32(None, None, 4, 0.5, None, None, 0, 0.5, None) == -1
    assert lib.augcl_adam_f32_devstep(None, None, None, None, 0, 1e-3, 0.9, 0.999, 1e-8, None, 1, 1.0, None) == -1  # no step state
    assert lib.augcl_gather_f32(None, None, None, 3, None, None, 0.0, 0.0, 0.0, None) == -1 and b"null table" in lib.augcl_last_error()


def test_ops_fail_loudly_without_gpu_tensor():
    import pytest
    import torch

    from augcl_amd import augmentations as aug
    x = torch.zeros(1, 9, 84, 84)
    with pytest.raises(_lib.AugclError):
        aug.random_shift(x)
    with pytest.raises(_lib.AugclError):
        aug.random_conv(x)
