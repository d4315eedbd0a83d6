"""GPU parity: hand-written HIP encoder kernels (f32 MFMA implicit GEMM: forward, backward-data,
backward-weights) vs a plain fp32 reference of the same op (torch conv2d on the CPU = what the
oracle's cnn_forward runs) and vs the golden vectors minted from the reference's SharedCNN."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

import inputs as gi
from augcl_amd import _lib, encoder_ops
from oracle import sac_oracle as so

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def ref_stack(x, ws, bs, stride, scale):
    h = x / scale
    for i, (w, b) in enumerate(zip(ws, bs)):
        if i:
            h = torch.relu(h)
        h = F.conv2d(h, w, b, stride=stride if i == 0 else 1)
    return h


def rel_err(a, b):
    a, b = a.double(), b.double()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


@pytest.mark.parametrize("B,C,H,n_layers,stride", [(3, 9, 84, 11, 2), (2, 9, 84, 4, 2), (5, 3, 40, 3, 2),
                                                   (2, 12, 50, 2, 1), (1, 9, 84, 1, 2), (7, 9, 36, 5, 2)])
def test_conv_stack_forward_backward(cuda, B, C, H, n_layers, stride):
    rs = np.random.RandomState(B * 100 + H)
    x = torch.from_numpy(rs.randint(0, 256, (B, C, H, H)).astype(np.float32))
    ws = [torch.from_numpy((rs.standard_normal((32, C if i == 0 else 32, 3, 3)) * np.sqrt(2.0 / (9 * (C if i == 0 else 32)))).astype(np.float32)) for i in range(n_layers)]
    bs = [torch.from_numpy((0.05 * rs.standard_normal(32)).astype(np.float32)) for _ in range(n_layers)]
    wr = [w.clone().requires_grad_(True) for w in ws]
    br = [b.clone().requires_grad_(True) for b in bs]
    ref = ref_stack(x, wr, br, stride, 255.0)
    gout = torch.from_numpy(rs.standard_normal(tuple(ref.shape)).astype(np.float32))
    (ref * gout).sum().backward()

    wg = [w.clone().to(cuda).requires_grad_(True) for w in ws]
    bg = [b.clone().to(cuda).requires_grad_(True) for b in bs]
    out = encoder_ops.conv_stack(x.to(cuda), wg, bg, first_stride=stride, input_scale=255.0)
    (out * gout.to(cuda)).sum().backward()
    torch.cuda.synchronize()
    assert out.shape == ref.shape
    assert rel_err(out.cpu(), ref.detach()) <= 1e-5
    for i in range(n_layers):
        assert rel_err(wg[i].grad.cpu(), wr[i].grad) <= 2e-5, f"dW layer {i}"
        assert rel_err(bg[i].grad.cpu(), br[i].grad) <= 2e-5, f"db layer {i}"


def test_no_grad_path_and_determinism(cuda):
    rs = np.random.RandomState(0)
    x = torch.from_numpy(rs.randint(0, 256, (4, 9, 84, 84)).astype(np.float32)).to(cuda)
    hp = dict(so.DEFAULTS)
    st = so.synth_state(5, so.critic_shapes((9, 84, 84), 6, hp))
    ids = so.conv_layer_ids(11)
    ws = [st[f"encoder.shared_cnn.layers.{i}.weight"].to(cuda).requires_grad_(True) for i in ids]
    bs = [st[f"encoder.shared_cnn.layers.{i}.bias"].to(cuda).requires_grad_(True) for i in ids]
    with torch.no_grad():
        a = encoder_ops.conv_stack(x, ws, bs)
        a2 = encoder_ops.conv_stack(x, ws, bs)
    b = encoder_ops.conv_stack(x, ws, bs)
    # under no_grad the 32 -> 32 layers run as LDS-resident chains (per-item scales, csrc/encoder_chain.hip), with gradients as
    # the per-layer f16-pair kernels (per-tensor scales): the same float32-accurate products, equal to float32 rounding
    assert torch.equal(a, a2) and torch.equal(b, encoder_ops.conv_stack(x, ws, bs))
    assert float((a - b).abs().max()) <= 2e-6 * float(b.abs().max())
    b.square().sum().backward()
    g1 = [w.grad.clone() for w in ws]
    for w in ws:
        w.grad = None
    encoder_ops.conv_stack(x, ws, bs).square().sum().backward()
    for g, w in zip(g1, ws):
        assert torch.equal(g, w.grad)  # fixed-order slab reduction: bitwise reproducible gradients


def test_shared_cnn_matches_reference_golden(cuda):
    """Same synthetic weights / batch as tests/golden/make_golden_sac.py's nets block."""
    G = np.load(os.path.join(ROOT, "tests", "golden", "sac_golden.npz"))
    hp = dict(so.DEFAULTS)
    cs = so.synth_state(31, so.critic_shapes((9, 84, 84), 6, hp))
    ob, _, _, _, _ = gi.transition_batch(33, 3, 6)
    ids = so.conv_layer_ids(11)
    ws = [cs[f"encoder.shared_cnn.layers.{i}.weight"].to(cuda) for i in ids]
    bs = [cs[f"encoder.shared_cnn.layers.{i}.bias"].to(cuda) for i in ids]
    out = encoder_ops.conv_stack(torch.from_numpy(ob.astype(np.float32)).to(cuda), ws, bs)
    want = torch.from_numpy(G["nets/cnn_out"])
    assert rel_err(out.cpu(), want) <= 1e-5


@pytest.mark.parametrize("precision", ["f32", "f32_x3"])
def test_full_batch_linearity_property(cuda, precision):
    """config-2 size (B=256): conv stack of a batch == stack of its halves (size-independent property).  Bit for bit on the
    split-bf16 / f32 MFMA kernels, whose arithmetic per image does not know the batch.  The f16-pair kernels (precision "f32")
    derive a power-of-two scale from the BATCH's maximum (exact, but it moves what falls into f16's subnormal range) and
    alternate the sign of their accumulation strip by strip to cancel the matrix pipe's rounding bias, so an image's result
    depends on its position in the launch in the last bits: held to 1e-6 of the output's range instead."""
    rs = np.random.RandomState(1)
    x = torch.from_numpy(rs.randint(0, 256, (256, 9, 84, 84)).astype(np.float32)).to(cuda)
    ws = [torch.from_numpy((rs.standard_normal((32, 9 if i == 0 else 32, 3, 3)) * 0.08).astype(np.float32)).to(cuda) for i in range(11)]
    bs = [torch.zeros(32, device=cuda) for _ in range(11)]
    with torch.no_grad():
        full = encoder_ops.conv_stack(x, ws, bs, precision=precision)
        again = encoder_ops.conv_stack(x, ws, bs, precision=precision)
        halves = torch.cat([encoder_ops.conv_stack(x[:100].contiguous(), ws, bs, precision=precision),
                            encoder_ops.conv_stack(x[100:].contiguous(), ws, bs, precision=precision)])
    assert full.shape == (256, 32, 21, 21)
    assert torch.equal(full, again)  # run-to-run: bit for bit in every precision
    if precision == "f32":
        assert float((full - halves).abs().max()) <= 1e-6 * float(full.abs().max())
    else:
        assert torch.equal(full, halves)


def _l2(a, b):
    a, b = a.double(), b.double()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


@pytest.mark.parametrize("B,gate_safe", [(3, True), (8, True), (4, False)])
def test_bf16_perf_mode_tracks_f32(cuda, B, gate_safe):
    """bf16 performance mode (NOT the parity path): same stack on bf16 MFMA / bf16 NHWC activations, checked against
    the f32 CPU reference.  With gate-safe weights (every ReLU input far from 0) only bf16 rounding remains: forward
    <= 2e-3, weight gradients <= 2e-2 (relative L2) through all 11 layers.  With He-scaled weights the bf16 forward
    noise (~1e-2) flips the ReLU gates of near-zero pre-activations, an inherent property of reduced-precision
    training: there the gradients are only required to stay aligned (cosine > 0.95)."""
    hp = dict(so.DEFAULTS)
    st = so.synth_state(5, so.critic_shapes((9, 84, 84), 6, hp), gate_safe=gate_safe)
    ids = so.conv_layer_ids(11)
    rs = np.random.RandomState(B)
    x = torch.from_numpy(rs.randint(0, 256, (B, 9, 84, 84)).astype(np.float32))
    ws = [st[f"encoder.shared_cnn.layers.{i}.weight"] for i in ids]
    bs = [st[f"encoder.shared_cnn.layers.{i}.bias"] for i in ids]
    wr = [w.clone().requires_grad_(True) for w in ws]
    br = [b.clone().requires_grad_(True) for b in bs]
    ref = ref_stack(x, wr, br, 2, 255.0)
    gout = torch.from_numpy(rs.standard_normal(tuple(ref.shape)).astype(np.float32))
    (ref * gout).sum().backward()
    wg = [w.clone().to(cuda).requires_grad_(True) for w in ws]
    bg = [b.clone().to(cuda).requires_grad_(True) for b in bs]
    encoder_ops.set_precision("bf16")
    try:
        out = encoder_ops.conv_stack(x.to(cuda), wg, bg, first_stride=2, input_scale=255.0)
        (out * gout.to(cuda)).sum().backward()
        torch.cuda.synchronize()
    finally:
        encoder_ops.set_precision("f32")
    assert out.dtype == torch.float32 and out.shape == ref.shape
    assert _l2(out.detach().cpu(), ref.detach()) <= (2e-3 if gate_safe else 3e-2)
    for i in range(11):
        g, r = wg[i].grad.cpu().double().flatten(), wr[i].grad.double().flatten()
        if gate_safe:
            assert _l2(wg[i].grad.cpu(), wr[i].grad) <= 2e-2, f"dW layer {i}"
            assert _l2(bg[i].grad.cpu(), br[i].grad) <= 2e-2, f"db layer {i}"
        else:
            assert float(g @ r / (g.norm() * r.norm())) > 0.95, f"dW layer {i} direction"


@pytest.mark.parametrize("precision", ["f32", "f32_x3", "f32_mfma", "bf16"])
@pytest.mark.parametrize("B", [1, 6, 256])
def test_dual_stack_forward_equals_two_passes(cuda, B, precision):
    """augcl_enc_stack_forward_dual: two weight sets over the same input in one launch per layer give exactly what two
    separate forward passes give (same kernels, same arithmetic per image)."""
    rs = np.random.RandomState(B)
    x = torch.from_numpy(rs.randint(0, 256, (B, 9, 84, 84)).astype(np.float32)).to(cuda)
    mk = lambda seed: ([torch.from_numpy((np.random.RandomState(seed + i).standard_normal((32, 9 if i == 0 else 32, 3, 3)) * 0.1).astype(np.float32)).to(cuda) for i in range(11)],  # noqa: E731
                       [torch.from_numpy((np.random.RandomState(seed + 50 + i).standard_normal(32) * 0.05).astype(np.float32)).to(cuda) for i in range(11)])
    (wa, ba), (wb, bb) = mk(100), mk(200)
    encoder_ops.set_precision(precision)
    try:
        with torch.no_grad():
            both = encoder_ops.conv_stack_dual(x, wa, ba, wb, bb)
            one_a = encoder_ops.conv_stack(x, wa, ba)
            one_b = encoder_ops.conv_stack(x, wb, bb)
        torch.cuda.synchronize()
    finally:
        encoder_ops.set_precision("f32")
    assert both is not None
    if precision == "f32":  # f16 pairs: the same arithmetic up to the strip-wise sign pattern (see the linearity test)
        for d, o in ((both[0], one_a), (both[1], one_b)):
            assert float((d - o).abs().max()) <= 1e-6 * float(o.abs().max())
    else:
        assert torch.equal(both[0], one_a) and torch.equal(both[1], one_b)
    # shapes without the pipelined first layer fall back (None), they do not fail
    with torch.no_grad():
        assert encoder_ops.conv_stack_dual(torch.rand(2, 3, 40, 40, device=cuda), [w[:, :3].contiguous() if i == 0 else w for i, w in enumerate(wa)], ba,
                                           [w[:, :3].contiguous() if i == 0 else w for i, w in enumerate(wb)], bb) is None


@pytest.mark.parametrize("pre", ["augcl_enc_x3_conv32", "augcl_enc_conv32"])
@pytest.mark.parametrize("B,H,W", [(2, 12, 70), (3, 70, 12), (1, 10, 100), (2, 33, 35), (5, 11, 11), (1, 3, 9), (1, 5, 6), (2, 6, 9),
                                   (1, 9, 130), (260, 41, 41), (3, 23, 23)])
def test_conv32_layer_entry_points_on_odd_shapes(cuda, B, H, W, pre):
    """The three 32 -> 32 layer kernels through the C ABI -- the split-bf16 form (augcl_enc_x3_*: precision "f32_x3") and the
    native f32 MFMA form -- on shapes the square training stack never produces: wide rows (more than one 64-lane chunk per
    row in the backward-data loader; rows too wide for the split tiles, which hand over to the f32 MFMA kernel), tall thin
    images, single-strip images, the smallest legal sizes (one quad per row), a batch that is not a multiple of the grid --
    against float64 ATen convolutions.  H x W is the layer's INPUT size."""
    import ctypes
    import torch.nn.functional as F
    from augcl_amd import _lib
    L = _lib.load()
    g = torch.Generator(device=cuda).manual_seed(B * 1000 + H * 10 + W)
    x = torch.relu(torch.randn(B, 32, H, W, device=cuda, generator=g))
    w = torch.randn(32, 32, 3, 3, device=cuda, generator=g) * 0.06
    bias = torch.randn(32, device=cuda, generator=g) * 0.1
    dz = torch.randn(B, 32, H - 2, W - 2, device=cuda, generator=g)
    st = _lib.current_stream(cuda)
    p = lambda t: t.data_ptr()  # noqa: E731
    # forward (+ ReLU)
    y = torch.full((B, 32, H - 2, W - 2), float("nan"), device=cuda)
    _lib.call(pre + "_forward", p(x), p(w), p(bias), p(y), B, H, W, 1, st)
    want = torch.relu(F.conv2d(x.double(), w.double(), bias.double()))
    assert float((y.double() - want).abs().max()) <= 2e-5 * float(want.abs().max())
    # backward-data, gated by the layer input's ReLU
    dx = torch.full((B, 32, H, W), float("nan"), device=cuda)
    _lib.call(pre + "_backward_data", p(dz), p(w), p(x), p(dx), B, H - 2, W - 2, st)
    want = F.conv_transpose2d(dz.double(), w.double()) * (x > 0)
    assert float((dx.double() - want).abs().max()) <= 2e-5 * float(want.abs().max())
    # ... and without a gate
    _lib.call(pre + "_backward_data", p(dz), p(w), None, p(dx), B, H - 2, W - 2, st)
    assert float((dx.double() - F.conv_transpose2d(dz.double(), w.double())).abs().max()) <= 2e-5 * float(want.abs().max() + 1)
    # backward-weights
    if W - 2 >= 8:  # the kernel's documented width range (10 <= W <= 170)
        ws = torch.empty(L.augcl_enc_wgrad_workspace_bytes(32) // 4, device=cuda)
        dw, db = torch.full((32, 32, 3, 3), float("nan"), device=cuda), torch.full((32,), float("nan"), device=cuda)
        _lib.call(pre + "_backward_weights", p(x), p(dz), p(dw), p(db), p(ws), ws.numel() * 4, B, H, W, st)
        xd = x.double().requires_grad_(False)
        wd = w.double().clone().requires_grad_(True)
        F.conv2d(xd, wd).backward(dz.double())
        assert float((dw.double() - wd.grad).abs().max()) <= 3e-5 * float(wd.grad.abs().max())
        assert float((db.double() - dz.double().sum((0, 2, 3))).abs().max()) <= 3e-5 * float(dz.double().sum((0, 2, 3)).abs().max() + 1)



@pytest.mark.parametrize("scale_x,scale_dz", [(1.0, 1.0), (3e4, 1e-9), (1e-6, 5e5)])
@pytest.mark.parametrize("B,H,W", [(2, 12, 70), (3, 70, 12), (1, 10, 100), (2, 33, 35), (5, 11, 11), (1, 3, 9), (1, 5, 6), (2, 6, 9),
                                   (1, 9, 130), (260, 41, 41), (3, 23, 23)])
def test_x2_layer_entry_points_on_odd_shapes(cuda, B, H, W, scale_x, scale_dz):
    """The f16-pair form of the three 32 -> 32 layer kernels (augcl_enc_x2_*: what precision "f32" runs inside the stack) through
    the C ABI on the same odd shapes, with the tensors' maxima from augcl_enc_amax_f32 -- at magnitudes far outside f16's own
    range (activations of 1e5, gradients of 1e-9: the per-tensor power-of-two scales carry them) -- against float64 ATen
    convolutions; the maxima the kernels hand on must be the exact maxima of what they wrote."""
    import torch.nn.functional as F
    from augcl_amd import _lib
    L = _lib.load()
    g = torch.Generator(device=cuda).manual_seed(B * 1000 + H * 10 + W)
    x = torch.relu(torch.randn(B, 32, H, W, device=cuda, generator=g)) * scale_x
    w = torch.randn(32, 32, 3, 3, device=cuda, generator=g) * 0.06
    bias = torch.randn(32, device=cuda, generator=g) * 0.1 * scale_x
    dz = torch.randn(B, 32, H - 2, W - 2, device=cuda, generator=g) * scale_dz
    st = _lib.current_stream(cuda)
    p = lambda t: t.data_ptr()  # noqa: E731
    fws = torch.empty(L.augcl_enc_frag_workspace_bytes(1, 1), dtype=torch.uint8, device=cuda)
    am = torch.full((4,), float("nan"), device=cuda)  # [max|x|, max|dz|, max|y|, max|dx|]
    _lib.call("augcl_enc_amax_f32", p(x), x.numel(), p(am), st)
    _lib.call("augcl_enc_amax_f32", p(dz), dz.numel(), p(am) + 4, st)
    assert float(am[0]) == float(x.abs().max()) and float(am[1]) == float(dz.abs().max())
    # forward (+ ReLU)
    y = torch.full((B, 32, H - 2, W - 2), float("nan"), device=cuda)
    _lib.call("augcl_enc_x2_conv32_forward", p(x), p(w), p(bias), p(y), B, H, W, 1, p(fws), fws.numel(), p(am), p(am) + 8, st)
    want = torch.relu(F.conv2d(x.double(), w.double(), bias.double()))
    assert float((y.double() - want).abs().max()) <= 2e-5 * float(want.abs().max())
    assert float(am[2]) == float(y.abs().max())
    # backward-data, gated by the layer input's ReLU
    dx = torch.full((B, 32, H, W), float("nan"), device=cuda)
    _lib.call("augcl_enc_x2_conv32_backward_data", p(dz), p(w), p(x), p(dx), B, H - 2, W - 2, p(fws), fws.numel(), p(am) + 4, p(am) + 12, st)
    want = F.conv_transpose2d(dz.double(), w.double()) * (x > 0)
    assert float((dx.double() - want).abs().max()) <= 2e-5 * float(want.abs().max())
    assert float(am[3]) == float(dx.abs().max())
    # ... and without a gate (and without handing the maximum on)
    _lib.call("augcl_enc_x2_conv32_backward_data", p(dz), p(w), None, p(dx), B, H - 2, W - 2, p(fws), fws.numel(), p(am) + 4, None, st)
    full = F.conv_transpose2d(dz.double(), w.double())
    assert float((dx.double() - full).abs().max()) <= 2e-5 * float(full.abs().max())
    # backward-weights
    if W - 2 >= 8:
        ws = torch.empty(L.augcl_enc_wgrad_workspace_bytes(32) // 4, device=cuda)
        dw, db = torch.full((32, 32, 3, 3), float("nan"), device=cuda), torch.full((32,), float("nan"), device=cuda)
        _lib.call("augcl_enc_x2_conv32_backward_weights", p(x), p(dz), p(dw), p(db), p(ws), ws.numel() * 4, B, H, W, p(am), p(am) + 4, st)
        wd = w.double().clone().requires_grad_(True)
        F.conv2d(x.double(), wd).backward(dz.double())
        assert float((dw.double() - wd.grad).abs().max()) <= 3e-5 * float(wd.grad.abs().max())
        dbw = dz.double().sum((0, 2, 3))
        assert float((db.double() - dbw).abs().max()) <= 3e-5 * float(dbw.abs().max() + dz.double().abs().sum((0, 2, 3)).max() * 1e-3)


def test_x2_layer_degenerate_inputs(cuda):
    """f16-pair layer kernels at the edges of their scaling: an all-zero input (maximum 0: the scale clamps, the output is exactly
    relu(bias)), an all-zero gradient, and a tensor with ONE element 2^20 above the rest (the small elements keep an absolute
    accuracy of 2^-36 of the maximum, i.e. the result stays within float32 rounding OF THE LARGE TERM'S SCALE)."""
    import torch.nn.functional as F
    from augcl_amd import _lib
    L = _lib.load()
    B, H, W = 3, 14, 17
    g = torch.Generator(device=cuda).manual_seed(5)
    w = torch.randn(32, 32, 3, 3, device=cuda, generator=g) * 0.06
    bias = torch.randn(32, device=cuda, generator=g)
    st = _lib.current_stream(cuda)
    p = lambda t: t.data_ptr()  # noqa: E731
    fws = torch.empty(L.augcl_enc_frag_workspace_bytes(1, 1), dtype=torch.uint8, device=cuda)
    am = torch.zeros(4, device=cuda)
    x = torch.zeros(B, 32, H, W, device=cuda)
    _lib.call("augcl_enc_amax_f32", p(x), x.numel(), p(am), st)
    y = torch.full((B, 32, H - 2, W - 2), float("nan"), device=cuda)
    _lib.call("augcl_enc_x2_conv32_forward", p(x), p(w), p(bias), p(y), B, H, W, 1, p(fws), fws.numel(), p(am), p(am) + 4, st)
    assert torch.equal(y, torch.relu(bias).view(1, 32, 1, 1).expand_as(y))
    assert float(am[1]) == float(torch.relu(bias).max())
    dz = torch.zeros(B, 32, H - 2, W - 2, device=cuda)
    dx = torch.full((B, 32, H, W), float("nan"), device=cuda)
    _lib.call("augcl_enc_amax_f32", p(dz), dz.numel(), p(am) + 8, st)
    _lib.call("augcl_enc_x2_conv32_backward_data", p(dz), p(w), None, p(dx), B, H - 2, W - 2, p(fws), fws.numel(), p(am) + 8, None, st)
    assert float(dx.abs().max()) == 0.0
    # one outlier
    x = torch.relu(torch.randn(B, 32, H, W, device=cuda, generator=g))
    x[1, 7, 5, 6] = 2.0 ** 20
    _lib.call("augcl_enc_amax_f32", p(x), x.numel(), p(am), st)
    _lib.call("augcl_enc_x2_conv32_forward", p(x), p(w), p(bias), p(y), B, H, W, 0, p(fws), fws.numel(), p(am), None, st)
    want = F.conv2d(x.double(), w.double(), bias.double())
    err = (y.double() - want).abs()
    assert float(err.max()) <= 2.0 ** 20 * 0.06 * 4 * 2.0 ** -24 * 8  # float32 rounding at the outlier's scale
    far = torch.ones_like(err, dtype=torch.bool)
    far[1, :, 3:6, 4:7] = False  # outputs that do not see the outlier: absolute error <= sum |w| x 2^-36 x 2^20 (~ 2e-5 of a value of order 1)
    assert float(err[far].max()) <= 288 * 0.3 * 2.0 ** -36 * 2.0 ** 20 + 1e-6
