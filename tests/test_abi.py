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
    assert _lib.load().augcl_abi_version() == 6


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


def test_chain_planner_and_infer_argument_checks():
    """Host side of augcl_enc_stack_infer (csrc/encoder_chain.hip, encoder_stack.cpp): what a chain launch fits (160 KiB of LDS,
    1280 staged pixels), tile rounds of the longest item, scratch size, argument checks -- no launch, no GPU."""
    lib = _lib.load()
    rounds = lib.augcl_internal_chain_rounds
    # the training stack's two chains: layers 1-4 on half images (25 input rows of 41 pixels), layers 5-10 on whole 33 x 33 images
    assert rounds(41, 41, 4, 2) == 8 + 7 + 6 + 5 and rounds(33, 33, 6, 1) == 8 + 7 + 6 + 5 + 5 + 4
    assert rounds(41, 41, 4, 1) == -1 and rounds(41, 41, 1, 1) == -1   # a whole 41 x 41 image does not fit LDS
    assert rounds(33, 33, 7, 1) == -1 and rounds(33, 33, 0, 1) == -1   # at most six layers, at least one
    assert rounds(13, 13, 6, 1) == 1 + 1 + 1 + 1 + 1 + 1               # 1 x 1 output: one tile per layer
    assert rounds(13, 13, 6, 2) == -1                                   # more parts than output rows
    assert rounds(21, 21, 2, 9) == 2 and rounds(21, 21, 2, 12) == -1    # 17 rows in parts of 2: nine parts hold them, twelve leave empty ones
    assert rounds(9, 400, 1, 7) == -1                                   # 3 input rows of 400 pixels: more than 1280 staged pixels
    # more parts = shorter items (what the planner trades against halo recomputation for small batches)
    assert rounds(41, 41, 4, 33) < rounds(41, 41, 4, 11) < rounds(41, 41, 4, 3) < rounds(41, 41, 4, 2)
    sz = lib.augcl_enc_stack_infer_scratch_bytes
    assert sz(11, 512, 9, 84, 84, 2) == 2 * 512 * 32 * 41 * 41 * 4 and sz(11, 0, 9, 84, 84, 2) == 0
    assert sz(11, 3, 9, 84, 84, 2) % 512 == 0                           # two 256-byte aligned buffers
    p = ctypes.c_void_p
    one = (p * 11)(*([1 << 20] * 11))
    assert lib.augcl_enc_stack_infer(None, one, one, None, None, 16, 16, 0, 11, 4, 9, 84, 84, 2, 255.0, 3, 16, 0, 16, None) == -1
    assert lib.augcl_enc_stack_infer(16, one, one, one, None, 16, 16, 0, 11, 4, 9, 84, 84, 2, 255.0, 3, 16, 0, 16, None) == -1
    assert b"weight set B" in lib.augcl_last_error()
    assert lib.augcl_enc_stack_infer(16, one, one, None, None, 16, 16, 0, 11, 4, 9, 20, 20, 2, 255.0, 3, 16, 0, 16, None) == -1
    assert b"too small" in lib.augcl_last_error()
    # frag_ws too small for ten 32 -> 32 layers: refused before anything is launched
    assert lib.augcl_enc_stack_infer(16, one, one, None, None, 16, 16, 1 << 30, 11, 4, 9, 84, 84, 2, 255.0, 3, 16, 64, 16, None) == -1
    assert b"frag_ws" in lib.augcl_last_error()
    assert lib.augcl_enc_stack_infer(16, one, one, None, None, 16, 16, 1 << 30, 11, 4, 9, 84, 84, 2, 255.0, 2, 16, 64, 16, None) == -1
    assert b"precision" in lib.augcl_last_error()
    # bf16 mode: 80-byte pixel slots -- a whole 41 x 41 image and all ten 32 -> 32 layers are one work item / one launch
    rb = lib.augcl_internal_chain_bf16_rounds
    assert rb(41, 41, 10, 1) == 12 + 11 + 10 + 9 + 8 + 7 + 6 + 5 + 5 + 4 and rb(41, 41, 13, 1) == -1 and rb(47, 47, 3, 1) == -1
    assert lib.augcl_enc_stack_infer(16, one, one, None, None, 16, 16, 1 << 30, 11, 4, 9, 84, 84, 2, 255.0, 1, 16, 64, None, None) == -1
    assert b"frag_ws" in lib.augcl_last_error()


def test_chained_gradient_forward_argument_checks():
    """augcl_enc_stack_forward_chained (ABI v6: the gradient-carrying forward of precision 3 through the layer chains): everything it
    refuses is refused before a launch -- no GPU needed."""
    lib = _lib.load()
    p = ctypes.c_void_p
    one = (p * 11)(*([1 << 20] * 11))
    fn = lib.augcl_enc_stack_forward_chained
    assert fn(None, one, one, one, 11, 4, 9, 84, 84, 2, 255.0, 16, 1 << 30, 16, None, None) == -1            # null input
    assert fn(16, one, one, one, 1, 4, 9, 84, 84, 2, 255.0, 16, 1 << 30, 16, None, None) == _lib.AUGCL_E_UNSUPPORTED  # a lone first layer has no chain
    assert fn(16, one, one, one, 11, 0, 9, 84, 84, 2, 255.0, 16, 1 << 30, 16, None, None) == 0                # an empty batch is done
    assert fn(16, one, one, one, 11, 4, 9, 84, 84, 2, 255.0, 16, 64, 16, None, None) == -1 and b"frag_ws" in lib.augcl_last_error()
    assert fn(16, one, one, one, 11, 4, 9, 84, 84, 2, 255.0, 16, 1 << 30, None, None, None) == -1 and b"amax" in lib.augcl_last_error()
    assert fn(16, one, one, one, 11, 4, 9, 20, 20, 2, 255.0, 16, 1 << 30, 16, None, None) == -1 and b"too small" in lib.augcl_last_error()
