"""conv_stack: the shared conv encoder (x/scale -> conv3x3(stride s) -> [ReLU -> conv3x3]*) as ONE op on the
hand-written gfx950 kernels behind the C ABI (augcl_enc_* in include/augcl.h): MFMA implicit-GEMM forward /
backward-data / backward-weights with fused /255, bias, ReLU and ReLU-mask epilogues; one autograd node for the whole
stack, activations kept post-ReLU in a per-call arena.  There is no other backend: without the library or a GPU
tensor the op raises.  (tools/aten_encoder.py swaps an ATen stack in for A/B timing -- outside the package.)

Precision: "f32" = parity path (f32 NCHW activations; the 32 -> 32 layers split every f32 operand exactly into three bf16
parts and run the six significant cross products on the bf16 MFMA pipe with f32 accumulation -- float32 accuracy at
2.67x the f32 MFMA rate, csrc/encoder_x3.hip); "f32_mfma" = the same tensors on the native f32 MFMA pipe (kept for
A/B); "bf16" = performance mode of BASELINE config 2 (bf16 NHWC activations, bf16 MFMA, f32 accumulation / master
weights / weight gradients).  One precision per process, selected with `set_precision` (`args.precision` calls it).
"""
import torch

from . import _lib

PRECISION = "f32"


def set_precision(name):
    global PRECISION
    if name not in ("f32", "f32_mfma", "bf16"):
        raise ValueError(f"precision must be 'f32', 'f32_mfma' or 'bf16', got {name!r}")
    PRECISION = name


def conv_stack_dual(x, weights_a, biases_a, weights_b, biases_b, first_stride=2, input_scale=255.0):
    """No-grad forward of TWO stacks with identical shapes over the same input in one launch per layer; returns the two
    outputs, or None when the fused path does not cover the case (it needs a 9 x H x 84 input and stride 2)."""
    if not x.is_cuda or len(weights_a) != len(weights_b) or len(weights_a) < 2:
        return None
    if any(a.shape != b.shape for a, b in zip(weights_a, weights_b)) or x.dtype != torch.float32:
        return None
    from . import encoder_hip
    out = encoder_hip.conv_stack_dual(x, weights_a, biases_a, weights_b, biases_b, first_stride, float(input_scale), PRECISION)
    if out is None:
        return None
    B = x.shape[0]
    return out[:B], out[B:]


def conv_stack(x, weights, biases, first_stride=2, input_scale=255.0, relu_first=False):
    """x: (B,C,H,W) f32 (0..255 when input_scale=255).  Returns the last conv's output (no final ReLU)."""
    _lib.require_cuda(x, "conv_stack input")
    from . import encoder_hip
    if relu_first:  # HeadCNN stacks (num_head_layers > 0): per-layer path, f32
        return encoder_hip.ConvStackLayerwiseFn.apply(x, first_stride, float(input_scale), True, len(weights),
                                                      "f32_mfma" if PRECISION == "f32_mfma" else "f32", *weights, *biases)
    for w in weights:
        assert w.shape[0] == 32 and w.is_contiguous(), "encoder kernels are built for 32 contiguous filters"
    precision = PRECISION if len(weights) > 1 else "f32"
    return encoder_hip.ConvStackFn.apply(x, first_stride, float(input_scale), precision, len(weights), *weights, *biases)
