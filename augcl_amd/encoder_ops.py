"""conv_stack: the shared conv encoder (x/scale -> conv3x3(stride s) -> [ReLU -> conv3x3]*) as ONE op.

Backends
  "hip"   (default, the product): hand-written gfx950 kernels behind the C ABI
          (augcl_enc_* in include/augcl.h): f32 MFMA implicit-GEMM forward / backward-data /
          backward-weights with fused /255, bias, ReLU and ReLU-mask epilogues; one autograd node
          for the whole stack, activations kept post-ReLU in a per-call workspace.
  "torch" developer baseline for A/B timing only (ATen/MIOpen convs on the GPU); selected with
          AUGCL_ENCODER=torch.  Never a fallback: if the HIP path is selected and the library or a
          GPU is missing, the op raises.
"""
import os

import torch
import torch.nn.functional as F

from . import _lib

BACKEND = os.environ.get("AUGCL_ENCODER", "hip")
PRECISION = os.environ.get("AUGCL_PRECISION", "f32")  # "f32" = parity path; "bf16" = performance mode (config 2)


def set_precision(name):
    global PRECISION
    assert name in ("f32", "bf16")
    PRECISION = name


def set_backend(name):
    global BACKEND
    assert name in ("hip", "torch")
    BACKEND = name


def _conv_stack_torch(x, weights, biases, first_stride, input_scale, relu_first):
    h = x / input_scale if input_scale != 1.0 else x
    for i, (w, b) in enumerate(zip(weights, biases)):
        if i > 0 or relu_first:
            h = torch.relu(h)
        h = F.conv2d(h, w, b, stride=first_stride if i == 0 else 1)
    return h


def conv_stack_dual(x, weights_a, biases_a, weights_b, biases_b, first_stride=2, input_scale=255.0):
    """No-grad forward of TWO stacks with identical shapes over the same input in one launch per layer; returns the two
    outputs, or None when the fused path does not cover the case (HIP backend, 9 x H x 84 input, stride 2)."""
    if BACKEND != "hip" or not x.is_cuda or len(weights_a) != len(weights_b) or len(weights_a) < 2:
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
    if BACKEND == "torch":
        return _conv_stack_torch(x, weights, biases, first_stride, input_scale, relu_first)
    _lib.require_cuda(x, "conv_stack input")
    from . import encoder_hip
    if relu_first:  # HeadCNN stacks (num_head_layers > 0): per-layer path, f32
        return encoder_hip.ConvStackLayerwiseFn.apply(x, first_stride, float(input_scale), True, len(weights), *weights,
                                                      *biases)
    for w in weights:
        assert w.shape[0] == 32 and w.is_contiguous(), "encoder kernels are built for 32 contiguous filters"
    precision = PRECISION if len(weights) > 1 else "f32"
    return encoder_hip.ConvStackFn.apply(x, first_stride, float(input_scale), precision, len(weights), *weights, *biases)
