"""conv_stack: the shared conv encoder (x/scale -> conv3x3(stride s) -> [ReLU -> conv3x3]*) as ONE op on the
hand-written gfx950 kernels behind the C ABI (augcl_enc_* in include/augcl.h): MFMA implicit-GEMM forward /
backward-data / backward-weights with fused /255, bias, ReLU and ReLU-mask epilogues; one autograd node for the whole
stack, activations kept post-ReLU in a per-call arena.  There is no other backend: without the library or a GPU
tensor the op raises.  (tools/aten_encoder.py swaps an ATen stack in for A/B timing -- outside the package.)

Precision: "f32" = parity path (f32 NCHW activations; the 32 -> 32 layers write every f32 operand as an f16 pair hi + lo' 2^-11
under a per-tensor power-of-two scale and run the three significant cross products on the f16 MFMA pipe with f32
accumulation -- float32 accuracy at 5.3x the f32 MFMA rate, csrc/encoder_x2.hip); "f32_x3" = the same tensors with every
operand split exactly into three bf16 parts, six products (csrc/encoder_x3.hip: no dynamic-range limit, half the rate);
"f32_mfma" = the native f32 MFMA pipe (kept for A/B); "bf16" = performance mode of BASELINE config 2 (bf16 NHWC activations, bf16 MFMA, f32 accumulation / master
weights / weight gradients).  `set_precision` sets the process default; an agent built with `args.precision` (or
`agent.set_precision(name)`) carries its own choice on its CNN modules, so a bf16 agent and an f32 agent (the pretrained
agent AugCL loads, an evaluation copy) can live in one process.
"""
import torch

from . import _lib

PRECISION = "f32"


def set_precision(name):
    global PRECISION
    PRECISION = _check(name)


def _check(name):
    if name not in ("f32", "f32_x3", "f32_mfma", "bf16"):
        raise ValueError(f"precision must be 'f32', 'f32_x3', 'f32_mfma' or 'bf16', got {name!r}")
    return name


def conv_stack_dual(x, weights_a, biases_a, weights_b, biases_b, first_stride=2, input_scale=255.0, precision=None):
    """No-grad forward of TWO stacks with identical shapes over the same input in one launch per layer; returns the two
    outputs, or None when the fused path does not cover the case (it needs a 9 x H x 84 input and stride 2)."""
    if not x.is_cuda or len(weights_a) != len(weights_b) or len(weights_a) < 2:
        return None
    if any(a.shape != b.shape for a, b in zip(weights_a, weights_b)) or x.dtype not in (torch.float32, torch.uint8):
        return None  # (uint8: observations as the replay ring stores them -- the first layer reads them as they are)
    from . import encoder_hip
    out = encoder_hip.conv_stack_dual(x, weights_a, biases_a, weights_b, biases_b, first_stride, float(input_scale),
                                      _check(precision or PRECISION))
    if out is None:
        return None
    B = x.shape[0]
    return out[:B], out[B:]


def conv_stack(x, weights, biases, first_stride=2, input_scale=255.0, relu_first=False, precision=None):
    """x: (B,C,H,W) f32 (0..255 when input_scale=255), or uint8 observations as the replay ring stores them (no-grad passes of
    precision "f32" feed them to the first layer as they are; every other case converts first).  Returns the last conv's output
    (no final ReLU).  precision: None = the process default (`set_precision`)."""
    _lib.require_cuda(x, "conv_stack input")
    from . import encoder_hip
    prec = _check(precision or PRECISION)
    if x.dtype == torch.uint8:
        if not relu_first and not torch.is_grad_enabled() and prec == "f32":
            out = encoder_hip.conv_stack_infer(x, [w.detach() for w in weights], [b.detach() for b in biases], None, None,
                                               first_stride, float(input_scale), prec)
            if out is not None:
                return out
        x = x.to(torch.float32)
    if relu_first:  # HeadCNN stacks (num_head_layers > 0): per-layer path, f32
        return encoder_hip.ConvStackLayerwiseFn.apply(x, first_stride, float(input_scale), True, len(weights),
                                                      "f32_mfma" if prec == "f32_mfma" else "f32_x3", *weights, *biases)
    for w in weights:
        assert w.shape[0] == 32 and w.is_contiguous(), "encoder kernels are built for 32 contiguous filters"
    if len(weights) == 1 and prec == "bf16":
        prec = "f32"  # a lone first layer has no bf16 activation to hand on
    if not torch.is_grad_enabled():
        # under torch.no_grad() (the actor step's detached pass, src/algorithms/sac.py:119-122; acting) nothing will ask for a
        # gradient, but the parameters still say requires_grad: hand them over detached, so that the node takes its no-grad
        # form (LDS-resident layer chains, no activation arena, no gate bits) instead of the gradient-carrying kernels
        weights, biases = [w.detach() for w in weights], [b.detach() for b in biases]
    return encoder_hip.ConvStackFn.apply(x, first_stride, float(input_scale), prec, len(weights), *weights, *biases)
