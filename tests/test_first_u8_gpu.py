"""GPU: the first encoder layer fed with uint8 observations (csrc/encoder_first_u8.hip, ABI v5) -- NormalizeImg + Conv2d(9,32,3,
stride 2) of src/algorithms/modules.py:85-90,145-155 on the replay ring's own bytes (src/utils.py:206-214) -- against the same
layer in float64, against the float32 kernel on the .float() of the same bytes, and through the whole no-grad stack.
Tolerance (written here): 2e-6 of the output tensor's maximum against float64 (float32 accumulation of 81 exact products)."""
import ctypes

import pytest
import torch
import torch.nn.functional as F

from augcl_amd import _lib, encoder_hip, encoder_ops

pytestmark = pytest.mark.gpu

P = ctypes.c_void_p


def _layer_u8(x8, w, b, relu):
    B = x8.shape[0]
    out = torch.empty((B, 32, 41, 41), dtype=torch.float32, device=x8.device)
    _lib.call("augcl_enc_conv_first_forward_u8", P(x8.data_ptr()), P(w.data_ptr()), P(b.data_ptr()) if b is not None else None,
              P(out.data_ptr()), B, 9, 84, 84, 2, 255.0, relu, _lib.current_stream(x8.device))
    return out


@pytest.mark.parametrize("B,relu,bias", [(1, 1, True), (3, 0, True), (64, 1, False), (256, 1, True), (513, 1, True)])
def test_first_layer_from_uint8_against_float64_and_the_f32_kernel(cuda, B, relu, bias):
    g = torch.Generator(device=cuda).manual_seed(B)
    x8 = torch.randint(0, 256, (B, 9, 84, 84), device=cuda, generator=g, dtype=torch.uint8)
    x8[0, :, :2] = 255
    x8[0, :, -2:] = 0
    w = torch.randn(32, 9, 3, 3, device=cuda, generator=g) * 0.2
    b = torch.randn(32, device=cuda, generator=g) if bias else None
    got = _layer_u8(x8, w, b, relu)
    want = F.conv2d(x8.double() / 255.0, w.double(), None if b is None else b.double(), stride=2)
    want = torch.relu(want) if relu else want
    scale = float(want.abs().max())
    assert float((got.double() - want).abs().max()) <= 2e-6 * scale
    # the float32 kernel on the .float() of the same bytes: both are float32-accurate, so they agree to the same bound
    ref = torch.empty_like(got)
    _lib.call("augcl_enc_conv_first_forward", P(x8.float().data_ptr()), P(w.data_ptr()), P(b.data_ptr()) if b is not None else None,
              P(ref.data_ptr()), B, 9, 84, 84, 2, 255.0, relu, _lib.current_stream(cuda))
    assert float((got - ref).abs().max()) <= 2e-6 * scale
    assert torch.equal(got, _layer_u8(x8, w, b, relu))  # no atomics in the data path


def test_first_layer_from_uint8_rejects_what_it_does_not_cover(cuda):
    x8 = torch.zeros((2, 9, 100, 100), dtype=torch.uint8, device=cuda)
    w, out = torch.zeros(32, 9, 3, 3, device=cuda), torch.empty((2, 32, 49, 49), device=cuda)
    rc = _lib.load().augcl_enc_conv_first_forward_u8(P(x8.data_ptr()), P(w.data_ptr()), None, P(out.data_ptr()), 2, 9, 100, 100, 2,
                                                    255.0, 1, _lib.current_stream(cuda))
    assert rc == _lib.AUGCL_E_UNSUPPORTED and b"84" in _lib.load().augcl_last_error()


@pytest.mark.parametrize("B,sets", [(8, 1), (256, 1), (64, 2), (256, 2)])
def test_no_grad_stack_from_uint8_equals_the_stack_from_float(cuda, B, sets):
    """augcl_enc_stack_infer_u8 vs augcl_enc_stack_infer on the .float() of the same observations (one and two weight sets: the
    target pass of SAC.calculate_target_Q, src/algorithms/sac.py:94-101): the eleven-layer outputs agree to float32 rounding."""
    g = torch.Generator(device=cuda).manual_seed(B + sets)
    x8 = torch.randint(0, 256, (B, 9, 84, 84), device=cuda, generator=g, dtype=torch.uint8)
    def mk():
        ws = [torch.randn(32, 9, 3, 3, device=cuda, generator=g) * (2.0 / 81) ** 0.5]
        ws += [torch.randn(32, 32, 3, 3, device=cuda, generator=g) * (2.0 / 288) ** 0.5 for _ in range(10)]
        return ws, [torch.randn(32, device=cuda, generator=g) * 0.05 for _ in range(11)]
    wa, ba = mk()
    wb, bb = mk() if sets == 2 else (None, None)
    from_u8 = encoder_hip.conv_stack_infer(x8, wa, ba, wb, bb, 2, 255.0, "f32")
    from_f32 = encoder_hip.conv_stack_infer(x8.float(), wa, ba, wb, bb, 2, 255.0, "f32")
    assert from_u8 is not None and from_u8.shape == (sets * B, 32, 21, 21)
    scale = float(from_f32.abs().max())
    assert float((from_u8 - from_f32).abs().max()) <= 1e-5 * scale
    if sets == 2:
        a, b_ = encoder_ops.conv_stack_dual(x8, wa, ba, wb, bb, precision="f32")
        assert torch.equal(a, from_u8[:B]) and torch.equal(b_, from_u8[B:])
    else:
        with torch.no_grad():
            assert torch.equal(encoder_ops.conv_stack(x8, wa, ba, precision="f32"), from_u8)
    # and with gradients enabled a uint8 batch is converted and takes the gradient-carrying kernels
    if B == 8:
        wr = [t.clone().requires_grad_(True) for t in wa]
        y = encoder_ops.conv_stack(x8, wr, ba, precision="f32")
        assert y.requires_grad and float((y.detach() - from_f32[:B]).abs().max()) <= 1e-5 * scale
