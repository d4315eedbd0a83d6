"""The no-grad conv stack as LDS-resident layer chains (augcl_enc_stack_infer, csrc/encoder_chain.hip) -- the path of
SAC.calculate_target_Q's two passes, the actor step's detached pass and acting (src/algorithms/sac.py:73-101, 119-122;
src/algorithms/modules.py:145-165) -- against float64 ATen convolutions of the same stack and against the per-layer f16-pair
kernels.  Tolerances are relative to the output's largest magnitude (float32 arithmetic: 1e-5 is north_star's bound, the
chains sit at ~1e-6 like the per-layer kernels)."""
import ctypes

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _stack(seed, n_layers, C=9, wscale=None, bscale=0.05, dev=None):
    rs = np.random.RandomState(seed)
    ws, bs = [], []
    for i in range(n_layers):
        cin = C if i == 0 else 32
        s = wscale if wscale is not None else float(np.sqrt(2.0 / (cin * 9)))  # He: activations keep their scale
        ws.append(torch.from_numpy((rs.standard_normal((32, cin, 3, 3)) * s).astype(np.float32)).to(dev))
        bs.append(torch.from_numpy((rs.standard_normal(32) * bscale).astype(np.float32)).to(dev))
    return ws, bs


def _ref64(x, ws, bs, stride, scale):
    y = x.double() / scale
    for i, (w, b) in enumerate(zip(ws, bs)):
        y = F.conv2d(y, w.double(), b.double(), stride=stride if i == 0 else 1)
        if i < len(ws) - 1:
            y = torch.relu(y)
    return y


def _rel(a, b):
    return float((a.double() - b).abs().max() / b.abs().max().clamp_min(1e-30))


@pytest.mark.parametrize("B,n_layers", [(1, 11), (2, 11), (3, 2), (5, 5), (8, 11), (33, 8), (64, 11), (256, 11), (300, 11)])
def test_infer_single_stack_against_float64(cuda, B, n_layers):
    from augcl_amd import encoder_hip
    x = torch.from_numpy(np.random.RandomState(B).randint(0, 256, (B, 9, 84, 84)).astype(np.float32)).to(cuda)
    ws, bs = _stack(10 + n_layers, n_layers, dev=cuda)
    out = encoder_hip.conv_stack_infer(x, ws, bs, None, None, 2, 255.0)
    assert out is not None
    want = _ref64(x, ws, bs, 2, 255.0)
    assert out.shape == want.shape
    assert _rel(out, want) <= 3e-6
    # the same pass through the per-layer f16-pair kernels (what a gradient-carrying forward runs): both float32-accurate
    encoder_hip._CHAINS = False
    try:
        with torch.no_grad():
            layerwise = encoder_hip.ConvStackFn.apply(x, 2, 255.0, "f32", n_layers, *ws, *bs)
    finally:
        encoder_hip._CHAINS = True
    assert _rel(layerwise, want) <= 3e-6
    assert _rel(out, layerwise.double()) <= 3e-6


@pytest.mark.parametrize("B", [1, 4, 37, 256])
def test_infer_two_weight_sets_equal_two_single_passes(cuda, B):
    """out[:B] / out[B:] of the two-set call = the single-set calls with each set.  The plan (parts per image) may differ
    between a B-image and a 2B-image launch, and with it the items' scales: equal to float32 rounding, not bit for bit."""
    from augcl_amd import encoder_hip
    x = torch.from_numpy(np.random.RandomState(7 + B).randint(0, 256, (B, 9, 84, 84)).astype(np.float32)).to(cuda)
    (wa, ba), (wb, bb) = _stack(1, 11, dev=cuda), _stack(2, 11, dev=cuda)
    both = encoder_hip.conv_stack_infer(x, wa, ba, wb, bb, 2, 255.0)
    assert both is not None and both.shape[0] == 2 * B
    for half, (w, b) in ((both[:B], (wa, ba)), (both[B:], (wb, bb))):
        one = encoder_hip.conv_stack_infer(x, w, b, None, None, 2, 255.0)
        want = _ref64(x, w, b, 2, 255.0)
        assert _rel(half, want) <= 3e-6 and _rel(one, want) <= 3e-6
        assert _rel(half, one.double()) <= 2e-6


def test_infer_is_reproducible_and_batch_position_free(cuda):
    """Two calls give the same bits; an image's result does not depend on which other images share the launch as long as
    the plan is the same (scales inside a chain are per work item; the chain's INPUT scale is the tensor's maximum, so the
    companions are given the same maximum here)."""
    from augcl_amd import encoder_hip
    rs = np.random.RandomState(3)
    x = torch.from_numpy(rs.randint(0, 256, (512, 9, 84, 84)).astype(np.float32)).to(cuda)
    x[:, :, 0, 0] = 255.0  # every image holds the batch maximum
    ws, bs = _stack(5, 11, dev=cuda)
    a = encoder_hip.conv_stack_infer(x, ws, bs, None, None, 2, 255.0)
    b = encoder_hip.conv_stack_infer(x, ws, bs, None, None, 2, 255.0)
    assert torch.equal(a, b)
    perm = torch.from_numpy(rs.permutation(512)).to(cuda)
    c = encoder_hip.conv_stack_infer(x[perm].contiguous(), ws, bs, None, None, 2, 255.0)
    # the first layer's output maximum (the chains' input scale) is a batch quantity: equal here up to its power of two
    assert _rel(c, a[perm].double()) <= 1e-6


@pytest.mark.parametrize("H,W,n_layers", [(84, 84, 11), (100, 100, 11), (64, 64, 6), (50, 70, 4), (36, 84, 3), (84, 40, 9)])
def test_infer_other_frame_sizes(cuda, H, W, n_layers):
    """Frame sizes the training stack never sees (the chain planner cuts them differently; the first layer takes its generic
    kernel when the pipelined one does not cover the shape)."""
    from augcl_amd import encoder_hip
    B = 6
    x = torch.from_numpy(np.random.RandomState(H + W).randint(0, 256, (B, 9, H, W)).astype(np.float32)).to(cuda)
    ws, bs = _stack(H, n_layers, dev=cuda)
    out = encoder_hip.conv_stack_infer(x, ws, bs, None, None, 2, 255.0)
    want = _ref64(x, ws, bs, 2, 255.0)
    if out is None:
        pytest.skip("shape not covered by the chains (the caller falls back to the per-layer kernels)")
    assert out.shape == want.shape and _rel(out, want) <= 3e-6


@pytest.mark.parametrize("scale_x,scale_w", [(1.0, 1.0), (3e4, 1e-3), (1e-6, 30.0)])
def test_infer_magnitudes_and_outlier_image(cuda, scale_x, scale_w):
    """Activations / weights far from O(1), and ONE image 2^12 brighter than its companions: the chain's input scale is the
    tensor's maximum, so the dim images sit 2^12 below it -- inside the f16 pairs' full-accuracy range; every image is
    compared on its own scale."""
    from augcl_amd import encoder_hip
    B = 9
    rs = np.random.RandomState(11)
    x = torch.from_numpy(rs.randint(0, 256, (B, 9, 84, 84)).astype(np.float32)).to(cuda) * scale_x
    x[4] *= 4096.0
    ws, bs = _stack(21, 11, dev=cuda, bscale=0.0)
    ws = [w * scale_w if i == 0 else w for i, w in enumerate(ws)]
    out = encoder_hip.conv_stack_infer(x, ws, bs, None, None, 2, 255.0)
    want = _ref64(x, ws, bs, 2, 255.0)
    for i in range(B):
        assert _rel(out[i], want[i]) <= 5e-6, i


@pytest.mark.parametrize("B,Hi,Wi,nl,parts", [(3, 41, 41, 4, 2), (2, 33, 33, 6, 1), (4, 41, 41, 1, 1), (5, 23, 23, 1, 3), (2, 20, 37, 3, 2),
                                              (1, 30, 12, 5, 4), (7, 13, 13, 5, 3), (2, 9, 80, 2, 5), (260, 33, 33, 6, 1),
                                              (2, 41, 41, 2, 37), (1, 15, 15, 6, 3)])
def test_chain_launch_on_odd_shapes(cuda, B, Hi, Wi, nl, parts):
    """One chain launch through its (internal) entry point: layer counts 1..6, parts that do not divide the rows, wide and
    narrow rows, a batch that is not a multiple of the grid, one-row parts -- against float64 convolutions."""
    from augcl_amd import _lib
    L = _lib.load()
    if L.augcl_internal_chain_rounds(Hi, Wi, nl, parts) < 0:
        pytest.skip("plan does not fit LDS")
    g = torch.Generator(device=cuda).manual_seed(B * 1000 + Hi * 10 + Wi + nl)
    x = torch.relu(torch.randn(B, 32, Hi, Wi, device=cuda, generator=g))
    ws = [torch.randn(32, 32, 3, 3, device=cuda, generator=g) * 0.083 for _ in range(nl)]
    bs = [torch.randn(32, device=cuda, generator=g) * 0.1 for _ in range(nl)]
    st = ctypes.c_void_p(_lib.current_stream(cuda))
    fb = L.augcl_enc_frag_workspace_bytes(1, 1)
    frag = torch.empty(nl * fb, dtype=torch.uint8, device=cuda)
    warr = (ctypes.c_void_p * nl)(*[w.data_ptr() for w in ws])
    assert L.augcl_internal_x2_prep_weights(warr, nl, 2, ctypes.c_void_p(frag.data_ptr()), None, 0, st) == 0  # mode 2: forward fragments, unscaled residuals
    am = torch.zeros(4, device=cuda)
    am[0] = x.abs().max()
    out = torch.full((B, 32, Hi - 2 * nl, Wi - 2 * nl), float("nan"), device=cuda)
    fa = (ctypes.c_void_p * nl)(*[frag.data_ptr() + j * fb for j in range(nl)])
    ba = (ctypes.c_void_p * nl)(*[b.data_ptr() for b in bs])
    L.augcl_internal_chain_forward.argtypes = [ctypes.c_void_p] * 6 + [ctypes.c_int] * 6 + [ctypes.c_void_p] * 3
    for relu_last in (0, 1):
        am[2:] = 0
        rc = L.augcl_internal_chain_forward(x.data_ptr(), out.data_ptr(), fa, ba, None, None, nl, B, Hi, Wi, parts, relu_last,
                                            am.data_ptr(), am.data_ptr() + 8, st)
        assert rc == 0, L.augcl_last_error()
        y = x.double()
        for j in range(nl):
            y = F.conv2d(y, ws[j].double(), bs[j].double())
            if j < nl - 1 or relu_last:
                y = torch.relu(y)
        assert not torch.isnan(out).any()
        assert _rel(out, y) <= 3e-6
        # the launch leaves max |out| for a following chain
        assert abs(float(am[2]) - float(out.abs().max())) <= 1e-6 * float(out.abs().max())


@pytest.mark.parametrize("B,n_layers,dual", [(1, 11, False), (3, 11, True), (8, 4, False), (64, 11, False), (256, 11, True), (300, 11, False)])
def test_infer_bf16_mode_equals_the_per_layer_bf16_kernels(cuda, B, n_layers, dual):
    """precision "bf16" (BASELINE config 2's bf16 leg; NOT the parity path): the chains round every layer's output to bf16
    exactly where the per-layer kernels store bf16, with the same products in the same order -- the two paths agree to the bit,
    and both sit at bf16 resolution from the float64 stack."""
    from augcl_amd import encoder_hip, encoder_ops
    x = torch.from_numpy(np.random.RandomState(B).randint(0, 256, (B, 9, 84, 84)).astype(np.float32)).to(cuda)
    (wa, ba), (wb, bb) = _stack(31, n_layers, dev=cuda), _stack(32, n_layers, dev=cuda)
    with torch.no_grad():
        if dual:
            out = encoder_hip.conv_stack_infer(x, wa, ba, wb, bb, 2, 255.0, "bf16")
        else:
            out = encoder_hip.conv_stack_infer(x, wa, ba, None, None, 2, 255.0, "bf16")
        assert out is not None
        encoder_hip._CHAINS = False
        try:
            if dual:
                a_, b_ = encoder_ops.conv_stack_dual(x, wa, ba, wb, bb, precision="bf16")
                layerwise = torch.cat([a_, b_])
            else:
                layerwise = encoder_ops.conv_stack(x, wa, ba, precision="bf16")
        finally:
            encoder_hip._CHAINS = True
    assert out.shape == layerwise.shape
    assert torch.equal(out, layerwise), float((out - layerwise).abs().max())
    want = _ref64(x, wa, ba, 2, 255.0)
    assert _rel(out[:B], want) <= 3e-2


@pytest.mark.parametrize("H,W,n_layers", [(100, 100, 11), (64, 64, 6), (50, 70, 4), (120, 84, 12)])
def test_infer_bf16_mode_other_frame_sizes(cuda, H, W, n_layers):
    """Frames whose first-layer output does not fit LDS whole (the planner cuts parts / several chains, the chain boundary is
    bf16 NHWC) against the per-layer bf16 kernels."""
    from augcl_amd import encoder_hip, encoder_ops
    B = 5
    x = torch.from_numpy(np.random.RandomState(H + W).randint(0, 256, (B, 9, H, W)).astype(np.float32)).to(cuda)
    ws, bs = _stack(H + 1, n_layers, dev=cuda)
    with torch.no_grad():
        out = encoder_hip.conv_stack_infer(x, ws, bs, None, None, 2, 255.0, "bf16")
        if out is None:
            pytest.skip("shape not covered by the chains")
        encoder_hip._CHAINS = False
        try:
            layerwise = encoder_ops.conv_stack(x, ws, bs, precision="bf16")
        finally:
            encoder_hip._CHAINS = True
    assert torch.equal(out, layerwise), float((out - layerwise).abs().max())


@pytest.mark.parametrize("B,H,W,n_layers", [(4, 84, 84, 11), (37, 84, 84, 11), (3, 100, 100, 11), (2, 64, 72, 5)])
def test_bf16_gradient_carrying_forward_through_chains_equals_per_layer(cuda, B, H, W, n_layers):
    """bf16 mode with gradients: the chains also store every layer's bf16 output for the backward pass (each row once, by the
    part that owns it) -- the forward value and every dW / db of the backward pass that reads those stores are bit-identical to
    the per-layer kernels' (same products, same order, same roundings)."""
    from augcl_amd import encoder_hip, encoder_ops
    x = torch.from_numpy(np.random.RandomState(B + H).randint(0, 256, (B, 9, H, W)).astype(np.float32)).to(cuda)
    ws, bs = _stack(41, n_layers, dev=cuda)
    res = []
    for chains in (True, False):
        encoder_hip._CHAINS = chains
        try:
            wg = [w.clone().requires_grad_(True) for w in ws]
            bg = [b.clone().requires_grad_(True) for b in bs]
            out = encoder_ops.conv_stack(x, wg, bg, precision="bf16")
            g = torch.from_numpy(np.random.RandomState(5).standard_normal(tuple(out.shape)).astype(np.float32)).to(cuda)
            (out * g).sum().backward()
            torch.cuda.synchronize()
            res.append((out.detach().clone(), [w.grad.clone() for w in wg], [b.grad.clone() for b in bg]))
        finally:
            encoder_hip._CHAINS = True
    (o1, dw1, db1), (o0, dw0, db0) = res
    assert torch.equal(o1, o0)
    for a, b in zip(dw1 + db1, dw0 + db0):
        assert torch.equal(a, b), float((a - b).abs().max())


def _kept(out, B, hs, wsz):
    """what the gradient-carrying f32 forward left for the backward pass: (activations, maxima, gate words) per intermediate layer,
    read out of the autograd node's arena"""
    fn = out.grad_fn
    _, _, _, n_layers, act_ptrs, _, amax_ptr, bits = fn.meta
    arena = fn.saved_tensors[1]
    base = arena.data_ptr()
    acts, words = [], []
    for i, p in enumerate(act_ptrs):
        o = (p - base) // 4
        acts.append(arena[o:o + B * 32 * hs[i] * wsz[i]].view(B, 32, hs[i], wsz[i]))
        bo = (bits[i] - base) // 4
        words.append(arena[bo:bo + B * hs[i] * wsz[i]].view(torch.int32).view(B, hs[i], wsz[i]))
    ao = (amax_ptr - base) // 4
    return acts, arena[ao:ao + n_layers].clone(), words


@pytest.mark.parametrize("B,H,W,n_layers", [(3, 84, 84, 11), (37, 84, 84, 11), (300, 84, 84, 11), (2, 100, 100, 11), (5, 64, 72, 5),
                                            (4, 84, 84, 2), (1, 84, 84, 11), (16, 50, 70, 4), (2, 100, 100, 14), (3, 84, 60, 7)])
def test_f32_gradient_carrying_forward_through_chains(cuda, B, H, W, n_layers):
    """Precision "f32" with gradients (the critic's pass, src/algorithms/sac.py:103-116): the 32 -> 32 layers as chains that also
    store every layer's float32 output, gate words and maximum (augcl_enc_stack_forward_chained, ABI v6).  What the backward pass
    is handed must be self-consistent -- maxima that bound the stored tensors tightly, gate words exactly (stored value > 0), every
    row of every layer written (parts own disjoint rows) -- every layer float32-accurate against float64, and the gradients of
    the backward pass that reads those stores as close to float64's as the per-layer kernels'."""
    from augcl_amd import encoder_hip, encoder_ops
    x = torch.from_numpy(np.random.RandomState(B + H).randint(0, 256, (B, 9, H, W)).astype(np.float32)).to(cuda)
    ws, bs = _stack(43, n_layers, dev=cuda)
    hs, wsz = [(H - 3) // 2 + 1], [(W - 3) // 2 + 1]
    for _ in range(1, n_layers):
        hs.append(hs[-1] - 2)
        wsz.append(wsz[-1] - 2)
    g = torch.from_numpy(np.random.RandomState(5).standard_normal((B, 32, hs[-1], wsz[-1])).astype(np.float32)).to(cuda)

    def ref64(gates):
        """float64 forward / backward; gates = None: ReLU; else the gate of every intermediate layer as the HIP forward took it (a
        pre-activation within rounding of zero may fall on either side, and ONE flipped gate moves a random-sign sum over B x h x w
        terms by 1 / sqrt(terms) ~ 1e-3: the gradients are compared on the gates the kernels actually used, as the referee does)"""
        w64 = [w.double().requires_grad_(True) for w in ws]
        b64 = [b.double().requires_grad_(True) for b in bs]
        y, acts64 = x.double() / 255.0, []
        for i in range(n_layers):
            y = F.conv2d(y, w64[i], b64[i], stride=2 if i == 0 else 1)
            if i < n_layers - 1:
                y = torch.relu(y) if gates is None else y * gates[i]
                acts64.append(y.detach())
        (y * g.double()).sum().backward()
        return y.detach(), acts64, [w.grad for w in w64] + [b.grad for b in b64]

    want_out, want_acts, _ = ref64(None)
    max_b = encoder_hip._CHAINS_GRAD_MAX_B
    for chains in (True, False):
        encoder_hip._CHAINS_GRAD = chains
        encoder_hip._CHAINS_GRAD_MAX_B = 1 << 30  # (the product takes the chains up to one image per CU; the kernel covers any batch)
        try:
            wg = [w.clone().requires_grad_(True) for w in ws]
            bg = [b.clone().requires_grad_(True) for b in bs]
            out = encoder_ops.conv_stack(x, wg, bg, precision="f32")
            acts, amax, words = _kept(out, B, hs, wsz)
            bit = torch.arange(32, device=cuda, dtype=torch.int32).view(1, 32, 1, 1)
            for i, a in enumerate(acts):
                assert torch.isfinite(a).all()
                # the maximum bounds the stored values to rounding (the backward-weights kernel scales its f16 pairs by it: 2^15 of
                # f16's 65504 are used) and is tight; not to the bit: the stored value is the (hi + lo) pair the next layer read, the
                # maximum that of the float32 it was cut from, and a part's halo rows are another part's rows under another scale
                assert float(a.abs().max()) * (1 - 1e-6) <= float(amax[i]) <= float(a.abs().max()) * (1 + 1e-6), (i, float(amax[i]), float(a.abs().max()))
                assert torch.equal(((words[i].unsqueeze(1) >> bit) & 1).bool(), a > 0), i
                assert _rel(a, want_acts[i]) <= 3e-6, (i, _rel(a, want_acts[i]))
            assert _rel(out, want_out) <= 3e-6
            gates = [(a > 0).double() for a in acts]
            (out * g).sum().backward()
            torch.cuda.synchronize()
            _, _, want = ref64(gates)
            for k, (got, r) in enumerate(zip([w.grad for w in wg] + [b.grad for b in bg], want)):
                assert _rel(got, r) <= 3e-6, (chains, k, _rel(got, r))
        finally:
            encoder_hip._CHAINS_GRAD = True
            encoder_hip._CHAINS_GRAD_MAX_B = max_b


def test_f32_kept_forward_arena_is_fully_written(cuda):
    """every row of every kept layer is stored once: a poisoned arena comes back without a single poison value (two parts per
    image in the first chain: the rows of the second part start where the first part's end)"""
    from augcl_amd import _lib, encoder_hip
    B, n_layers = 7, 11
    x = torch.from_numpy(np.random.RandomState(3).randint(0, 256, (B, 9, 84, 84)).astype(np.float32)).to(cuda)
    ws, bs = _stack(44, n_layers, dev=cuda)
    hs = [41 - 2 * i for i in range(n_layers)]
    acts = [torch.full((B, 32, h, h), float("nan"), device=cuda) for h in hs]
    bits = [torch.full((B, h, h), -1, dtype=torch.int32, device=cuda) for h in hs[:-1]]
    amax = torch.zeros(n_layers, device=cuda)
    fws = encoder_hip._frag_ws(cuda, n_layers - 1)
    P = encoder_hip._ptr_array
    rc = _lib.load().augcl_enc_stack_forward_chained(
        x.data_ptr(), P([w.data_ptr() for w in ws]), P([b.data_ptr() for b in bs]), P([a.data_ptr() for a in acts]), n_layers, B, 9, 84,
        84, 2, 255.0, fws.data_ptr(), fws.numel(), amax.data_ptr(), P([b.data_ptr() for b in bits] + [None]), _lib.current_stream(cuda))
    assert rc == 0
    torch.cuda.synchronize()
    for a in acts:
        assert torch.isfinite(a).all()
    want = _ref64(x, ws, bs, 2, 255.0)
    assert _rel(acts[-1], want) <= 3e-6
    for i, w_ in enumerate(bits):
        bit = torch.arange(32, device=cuda, dtype=torch.int32).view(1, 32, 1, 1)
        assert torch.equal(((w_.unsqueeze(1) >> bit) & 1).bool(), acts[i] > 0), i
