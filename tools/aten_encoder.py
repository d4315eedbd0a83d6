"""A/B aid (NOT part of the product): swap an ATen/MIOpen conv stack in for the HIP encoder, to time the same update()
with the vendor library's convolutions.

    python - <<EOF
    import torch, tools.aten_encoder as t; t.install()
    from augcl_amd import bench_update; print(bench_update.run(torch.device("cuda:0"), None, 1, 0, steps=50))
    EOF
"""
import torch
import torch.nn.functional as F

from augcl_amd import encoder_ops


def _conv_stack_aten(x, weights, biases, first_stride=2, input_scale=255.0, relu_first=False):
    h = x / input_scale if input_scale != 1.0 else x
    for i, (w, b) in enumerate(zip(weights, biases)):
        if i > 0 or relu_first:
            h = torch.relu(h)
        h = F.conv2d(h, w, b, stride=first_stride if i == 0 else 1)
    return h


def install():
    encoder_ops.conv_stack = _conv_stack_aten
    encoder_ops.conv_stack_dual = lambda *a, **k: None
