"""autograd node for the whole conv stack on the hand-written HIP encoder kernels (include/augcl.h,
augcl_enc_*).  Activations are stored post-ReLU (the forward kernels fuse bias + the following ReLU);
backward walks the stack once: backward-weights from (a_{l-1}, dz_l), backward-data with the ReLU mask
of the layer below fused in.  No gradient is produced for the image input (never needed on this path)."""
import torch

from . import _lib

_ws = {}


def _workspace(dev, c_in, count=1):
    """count > 1: one slice per layer, so the stack backward can reduce every layer's slabs in one batched launch."""
    key = (dev, c_in, count)
    if key not in _ws:
        if dev.type == "cuda" and torch.cuda.is_current_stream_capturing():
            raise _lib.AugclError("encoder workspace requested for the first time inside a hipGraph capture: it would live "
                                  "in the graph's private pool and be reused by eager calls; run one eager update first "
                                  "(SAC.enable_hip_graphs keeps warmup >= 1)")
        n = _lib.load().augcl_enc_wgrad_workspace_bytes(int(c_in))
        _ws[key] = torch.empty(n // 4 * count, dtype=torch.float32, device=dev)
    return _ws[key]


_fws = {}


_FWS_MAX_LAYERS32 = 32  # the C ABI's limit (augcl_enc_stack_*: at most 33 layers), two weight sets


def _frag_ws(dev, n_layers32, sets=1):
    """scratch for the prepared weight fragments of the split layers (augcl_enc_stack_*'s frag_ws): ONE buffer per device,
    allocated once at the ABI's maximum (32 layers x 2 weight sets, ~3.5 MB) and never replaced -- captured hipGraphs bake
    its address in, so growing it on demand would leave them writing into freed memory (the acting graph is captured with
    one weight set, the first update() then asks for two).  Its contents only live inside one stack call."""
    lib = _lib.load()
    need = lib.augcl_enc_frag_workspace_bytes(int(n_layers32), int(sets))
    t = _fws.get(dev)
    if t is None:
        if dev.type == "cuda" and torch.cuda.is_current_stream_capturing():
            raise _lib.AugclError("weight-fragment workspace requested for the first time inside a hipGraph capture; run one "
                                  "eager update first (SAC.enable_hip_graphs keeps warmup >= 1)")
        t = _fws[dev] = torch.empty(max(lib.augcl_enc_frag_workspace_bytes(_FWS_MAX_LAYERS32, 2), 16), dtype=torch.uint8,
                                    device=dev)
    if t.numel() < need:
        raise _lib.AugclError(f"weight-fragment workspace: {n_layers32} layers x {sets} sets exceed the ABI maximum")
    return t


def _out_hw(h, stride):
    return (h - 3) // stride + 1


def _conv32_entry(precision):
    """prefix of the three 32 -> 32 layer entry points for a precision name (the per-layer path has no bf16 form, and no
    f16-pair form: single-layer calls have no producer that leaves the tensor maxima, so they take the split-bf16 kernels)"""
    return "augcl_enc_conv32" if precision == "f32_mfma" else "augcl_enc_x3_conv32"


class ConvStackLayerwiseFn(torch.autograd.Function):  # per-layer calls: HeadCNN stacks (relu_first) only
    @staticmethod
    def forward(ctx, x, first_stride, input_scale, relu_first, n_layers, precision, *wb):
        weights, biases = wb[:n_layers], wb[n_layers:]
        assert x.dtype == torch.float32 and x.is_contiguous()
        B, C, H, W = x.shape
        st = _lib.current_stream(x.device)
        call, p = _lib.call, _lib.ptr
        if relu_first:  # HeadCNN: the stack starts with ReLU on an un-activated input
            x = torch.relu(x)
        acts = [x]
        h, w_ = H, W
        need_grad = any(t.requires_grad for t in wb)
        for i in range(n_layers):
            wi, bi = weights[i].detach(), biases[i].detach()
            assert wi.is_contiguous() and wi.shape[0] == 32, "encoder kernels are built for 32 filters"
            relu = 1 if i < n_layers - 1 else 0
            if i == 0 and (wi.shape[1] != 32 or first_stride != 1 or input_scale != 1.0):
                ho, wo = _out_hw(h, first_stride), _out_hw(w_, first_stride)
                out = torch.empty((B, 32, ho, wo), dtype=torch.float32, device=x.device)
                call("augcl_enc_conv_first_forward", p(acts[-1]), p(wi), p(bi), p(out), B, wi.shape[1], h, w_,
                     first_stride, input_scale, relu, st)
            else:
                ho, wo = h - 2, w_ - 2
                out = torch.empty((B, 32, ho, wo), dtype=torch.float32, device=x.device)
                call(_conv32_entry(precision) + "_forward", p(acts[-1]), p(wi), p(bi), p(out), B, h, w_, relu, st)
            if need_grad or i == n_layers - 1:
                acts.append(out)
            else:
                acts = [acts[0], out] if i == 0 else [acts[0], out]  # keep only what the next layer reads
            h, w_ = ho, wo
        out = acts[-1]
        if need_grad:
            ctx.save_for_backward(*acts[:-1], *[t.detach() for t in weights])
            ctx.meta = (first_stride, input_scale, n_layers, tuple(x.shape), bool(relu_first), precision)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        first_stride, input_scale, n_layers, xshape, relu_first, precision = ctx.meta
        pre = _conv32_entry(precision)
        want_dx = ctx.needs_input_grad[0]
        dx = None
        saved = ctx.saved_tensors
        acts, weights = saved[:n_layers], saved[n_layers:]
        B = xshape[0]
        dev = grad_out.device
        st = _lib.current_stream(dev)
        call, p = _lib.call, _lib.ptr
        dz = grad_out.contiguous()
        dws, dbs = [None] * n_layers, [None] * n_layers
        for i in range(n_layers - 1, -1, -1):
            a_prev, wi = acts[i], weights[i]
            dw = torch.empty_like(wi)
            db = torch.empty(32, dtype=torch.float32, device=dev)
            first = i == 0 and (wi.shape[1] != 32 or first_stride != 1 or input_scale != 1.0)
            ws = _workspace(dev, wi.shape[1])
            if first:
                call("augcl_enc_conv_first_backward_weights", p(a_prev), p(dz), p(dw), p(db), p(ws), ws.numel() * 4, B,
                     wi.shape[1], a_prev.shape[2], a_prev.shape[3], first_stride, input_scale, st)
            else:
                call(pre + "_backward_weights", p(a_prev), p(dz), p(dw), p(db), p(ws), ws.numel() * 4, B,
                     a_prev.shape[2], a_prev.shape[3], st)
            dws[i], dbs[i] = dw, db
            if i > 0:
                dact = torch.empty_like(a_prev)
                call(pre + "_backward_data", p(dz), p(wi), p(a_prev), p(dact), B, dz.shape[2], dz.shape[3], st)
                dz = dact
            elif want_dx:
                if first:
                    raise NotImplementedError("gradient w.r.t. the image input of the strided first layer is not "
                                              "needed on the SAC path and not implemented")
                dx = torch.empty_like(a_prev)  # HeadCNN: input is the shared CNN's (pre-ReLU) output
                call(pre + "_backward_data", p(dz), p(wi), p(a_prev) if relu_first else None, p(dx), B,
                     dz.shape[2], dz.shape[3], st)
        return (dx, None, None, None, None, None, *dws, *dbs)


# backward-weights of the bf16 mode: "…_mfma" = bf16 MFMA with transposed LDS reads (37 us / layer); the other entry
# point keeps the f32 MFMA pipe with bf16 inputs widened while staging (59 us).  Switchable for A/B timing.
_BF16_WGRAD = "augcl_enc_bf16_conv32_backward_weights_mfma"

# precision name -> the C ABI's precision code (include/augcl.h, augcl_enc_stack_*): "f32" = f32 NCHW activations with
# the 32 -> 32 layers' products as f16 pairs on the f16 matrix pipe (float32-accurate, the parity path); "f32_x3" = the same
# with three-way bf16 splits (six products, no range limit); "f32_mfma" = the same tensors on the native f32 MFMA pipe
# (round 1-2's parity path, kept for A/B); "bf16" = bf16 NHWC performance mode.
PRECISION_CODE = {"f32_mfma": 0, "bf16": 1, "f32_x3": 2, "f32": 3}
_SPLIT = ("f32", "f32_x3")  # the precisions whose stack calls take a weight-fragment workspace
_GATE_BITS = True           # (A/B switch: False = backward-data gates on the float activations, as the split-bf16 path does)
_AMAX = 64                  # floats appended to an arena for the per-tensor maxima of precision "f32" (>= 2 x layers)


def _ptr_array(ptrs):
    return (_lib.ctypes.c_void_p * len(ptrs))(*ptrs)


_CHAINS_BWD = True  # (A/B switch) bf16 mode: backward-data of the whole stack as one chain launch
_U8_FIRST = True  # (A/B switch) uint8 observations go into the first layer as they are (no-grad passes of precision "f32")
_CHAINS = True  # (A/B switch) no-grad passes of precision "f32" through augcl_enc_stack_infer: LDS-resident layer chains
_CHAINS_GRAD = True  # (A/B switch) ... and the gradient-carrying forward of precision "f32" through augcl_enc_stack_forward_chained
# ... for batches of at most this many images.  Measured (tools/keep_bench.py, one MI355X, forward / forward + backward of the
# eleven-layer stack, us): B = 64: 125 / 520 against the per-layer kernels' 159-165 / 550; 128: 160 / 672 against 192 / 688;
# 256: 250 / 911 against 283 / 933; 320: 356 / 1131 against 332 / 1084; 512: 438 / 1417 against 440 / 1408.  The stores, un-scales
# and gate bits in the wave that issues the matrix instructions cost what the chain saves once every CU has more than one
# image (and 320 images x 2 parts on 256 CUs quantise badly): above one image per CU the per-layer kernels run.
_CHAINS_GRAD_MAX_B = 256
_infer_ws = {}


def _infer_scratch(dev, n_layers, b_total, C, H, W, first_stride):
    """scratch of augcl_enc_stack_infer (two buffers of the first layer's output + the maxima behind them): one tensor per
    (device, size), never replaced, so captured hipGraphs may keep its address"""
    lib = _lib.load()
    need = lib.augcl_enc_stack_infer_scratch_bytes(int(n_layers), int(b_total), int(C), int(H), int(W), int(first_stride))
    key = (dev, need)
    t = _infer_ws.get(key)
    if t is None:
        if dev.type == "cuda" and torch.cuda.is_current_stream_capturing():
            raise _lib.AugclError("no-grad conv-stack scratch requested for the first time inside a hipGraph capture; run one "
                                  "eager pass of this shape first")
        t = _infer_ws[key] = torch.empty(need + 4 * _AMAX * 2, dtype=torch.uint8, device=dev)
    return t, need


def conv_stack_infer(x, weights_a, biases_a, weights_b, biases_b, first_stride, input_scale, precision="f32"):
    """No-grad forward of one conv stack (weights_b is None) or of two stacks with identical shapes over the SAME input,
    f16-pair arithmetic, 32 -> 32 layers as LDS-resident chains (augcl_enc_stack_infer).  Returns (B or 2B, 32, h, w) f32 --
    rows [0,B) from set A, [B,2B) from set B -- or None when the shape is not covered."""
    n_layers = len(weights_a)
    B, C, H, W = x.shape
    u8 = x.dtype == torch.uint8  # observations as the replay ring stores them: the first layer reads them as they are (ABI v5)
    if not _CHAINS or not x.is_contiguous() or not (u8 or x.dtype == torch.float32) or 2 * (n_layers + 1) > 2 * _AMAX:
        return None
    if precision not in ("f32", "bf16") or (precision == "bf16" and n_layers < 2):
        return None
    if u8 and (not _U8_FIRST or precision != "f32" or n_layers < 2 or (C, H, W) != (9, 84, 84) or first_stride != 2 or x.data_ptr() % 8):
        return None
    dev = x.device
    sets = 1 if weights_b is None else 2
    h, w_ = _out_hw(H, first_stride), _out_hw(W, first_stride)
    h, w_ = h - 2 * (n_layers - 1), w_ - 2 * (n_layers - 1)
    if h < 1 or w_ < 1:
        return None
    out = torch.empty((sets * B, 32, h, w_), dtype=torch.float32, device=dev)
    ws, need = _infer_scratch(dev, n_layers, sets * B, C, H, W, first_stride)
    fws = _frag_ws(dev, max(n_layers - 1, 1), sets)
    d = lambda ts: _ptr_array([t.detach().data_ptr() for t in ts])  # noqa: E731
    entry = _lib.load().augcl_enc_stack_infer_u8 if u8 else _lib.load().augcl_enc_stack_infer
    rc = entry(
        x.data_ptr(), d(weights_a), d(biases_a), d(weights_b) if sets == 2 else None, d(biases_b) if sets == 2 else None,
        out.data_ptr(), ws.data_ptr(), need, n_layers, B, C, H, W, first_stride, input_scale, PRECISION_CODE[precision],
        fws.data_ptr(), fws.numel(), ws.data_ptr() + need, _lib.current_stream(dev))
    if rc == _lib.AUGCL_E_UNSUPPORTED:
        return None
    _lib.check(rc, "augcl_enc_stack_infer")
    return out


def conv_stack_dual(x, weights_a, biases_a, weights_b, biases_b, first_stride, input_scale, precision):
    """Two conv stacks with the same shapes (different weights) over the SAME input, no gradients: one launch per layer
    for both (augcl_enc_stack_forward_dual).  Returns a (2B,32,h,w) f32 tensor -- rows [0,B) from set A, [B,2B) from
    set B -- or None when the shape is not covered (the caller runs two ordinary passes)."""
    n_layers = len(weights_a)
    B, C, H, W = x.shape
    if n_layers < 2 or C != 9 or W != 84 or first_stride != 2 or not x.is_contiguous():
        return None
    if precision in ("f32", "bf16"):
        out = conv_stack_infer(x, weights_a, biases_a, weights_b, biases_b, first_stride, input_scale, precision)
        if out is not None:
            return out
    if x.dtype == torch.uint8:  # the per-layer kernels read float32 observations
        x = x.to(torch.float32)
    dev = x.device
    bf16 = precision == "bf16"
    adt = torch.bfloat16 if bf16 else torch.float32
    hs, wsz = [_out_hw(H, first_stride)], [_out_hw(W, first_stride)]
    for _ in range(1, n_layers):
        hs.append(hs[-1] - 2)
        wsz.append(wsz[-1] - 2)
    numel = [2 * B * 32 * h * w_ for h, w_ in zip(hs, wsz)]
    out = torch.empty((2 * B, 32, hs[-1], wsz[-1]), dtype=torch.float32, device=dev)
    big = (max(numel[:-1]) + 7) // 8 * 8
    assert 2 * n_layers <= _AMAX
    arena = torch.empty(2 * big + (_AMAX if precision == "f32" else 0), dtype=adt, device=dev)  # two ping-pong buffers (+ maxima)
    ptrs = [arena.data_ptr() + (i & 1) * big * arena.element_size() for i in range(n_layers - 1)] + [out.data_ptr()]
    d = lambda ts: _ptr_array([t.detach().data_ptr() for t in ts])  # noqa: E731
    fws = _frag_ws(dev, n_layers - 1, 2) if precision in _SPLIT else None
    _lib.call("augcl_enc_stack_forward_dual", x.data_ptr(), d(weights_a), d(biases_a), d(weights_b), d(biases_b),
              _ptr_array(ptrs), n_layers, B, C, H, W, first_stride, input_scale, PRECISION_CODE[precision],
              _lib.ptr(fws), fws.numel() if fws is not None else 0,
              arena.data_ptr() + 2 * big * 4 if precision == "f32" else None, _lib.current_stream(dev))
    return out


class ConvStackFn(torch.autograd.Function):
    """The SharedCNN stack through the whole-stack C entry points: ONE FFI call per forward, one per backward, one
    activation arena per pass.  precision "f32" = parity path (f32 NCHW activations; 32 -> 32 layers with float32-accurate
    products on the bf16 MFMA pipe), "f32_mfma" = the same on the native f32 MFMA pipe, "bf16" = performance mode (bf16
    NHWC activations, bf16 MFMA, f32 accumulation / master weights / weight gradients)."""

    @staticmethod
    def forward(ctx, x, first_stride, input_scale, precision, n_layers, *wb):
        weights = [t.detach() for t in wb[:n_layers]]
        biases = [t.detach() for t in wb[n_layers:]]
        assert x.dtype == torch.float32 and x.is_contiguous()
        B, C, H, W = x.shape
        dev = x.device
        bf16 = precision == "bf16"
        adt = torch.bfloat16 if bf16 else torch.float32
        hs = [_out_hw(H, first_stride)]
        wsz = [_out_hw(W, first_stride)]
        for _ in range(1, n_layers):
            hs.append(hs[-1] - 2)
            wsz.append(wsz[-1] - 2)
        numel = [B * 32 * h * w_ for h, w_ in zip(hs, wsz)]
        need_grad = any(t.requires_grad for t in wb)
        if not need_grad and precision in ("f32", "bf16"):  # nothing is kept for a backward pass: the layers run as LDS-resident chains
            out = conv_stack_infer(x, weights, biases, None, None, first_stride, input_scale, precision)
            if out is not None:
                return out
        out = torch.empty((B, 32, hs[-1], wsz[-1]), dtype=torch.float32, device=dev)
        al = 8  # elements: keeps every slice 16-byte aligned in both element types
        tail = _AMAX if precision == "f32" else 0  # the per-tensor maxima live behind the activations (the backward reads them)
        assert n_layers <= _AMAX
        bits = None
        if need_grad:  # one arena holds every intermediate activation
            offs, tot = [], 0
            for n in numel[:-1]:
                offs.append(tot)
                tot += (n + al - 1) // al * al
            tot = max(tot, al)
            if tail and _GATE_BITS:  # ... and, behind the maxima, one gate word per pixel of every intermediate activation (backward-data's ReLU gate)
                boffs, btot = [], 0
                for h, w_ in zip(hs[:-1], wsz[:-1]):
                    boffs.append(btot)
                    btot += (B * h * w_ + 3) // 4 * 4
                tail += btot
            arena = torch.empty(tot + tail, dtype=adt, device=dev)
            ptrs = [arena.data_ptr() + o * arena.element_size() for o in offs]
            if tail and _GATE_BITS:
                bits = _ptr_array([arena.data_ptr() + (tot + _AMAX + o) * 4 for o in boffs] + [None])
        else:  # two ping-pong buffers
            big = (max(numel[:-1], default=al) + al - 1) // al * al
            tot = 2 * big
            arena = torch.empty(tot + tail, dtype=adt, device=dev)
            ptrs = [arena.data_ptr() + (i & 1) * big * arena.element_size() for i in range(n_layers - 1)]
        ptrs.append(out.data_ptr())
        amax = arena.data_ptr() + tot * 4 if tail else None
        # (bf16 mode: the chains' prepared bf16 fragments live in the same workspace; _CHAINS off = per-layer kernels, no workspace)
        fws = _frag_ws(dev, n_layers - 1) if (precision in _SPLIT or (bf16 and _CHAINS)) and n_layers > 1 else None
        done = False
        if need_grad and precision == "f32" and _CHAINS and _CHAINS_GRAD and n_layers > 1 and bits is not None and B <= _CHAINS_GRAD_MAX_B:
            # every layer's output is written once and read from LDS by the next layer (ABI v6); same tensors for the backward pass
            rc = _lib.load().augcl_enc_stack_forward_chained(
                x.data_ptr(), _ptr_array([t.data_ptr() for t in weights]), _ptr_array([t.data_ptr() for t in biases]),
                _ptr_array(ptrs), n_layers, B, C, H, W, first_stride, input_scale, fws.data_ptr(), fws.numel(), amax, bits,
                _lib.current_stream(dev))
            if rc != _lib.AUGCL_E_UNSUPPORTED:
                _lib.check(rc, "augcl_enc_stack_forward_chained")
                done = True
        if not done:
            _lib.call("augcl_enc_stack_forward", x.data_ptr(), _ptr_array([t.data_ptr() for t in weights]),
                      _ptr_array([t.data_ptr() for t in biases]), _ptr_array(ptrs), n_layers, B, C, H, W, first_stride,
                      input_scale, PRECISION_CODE[precision], _lib.ptr(fws), fws.numel() if fws is not None else 0, amax, bits,
                      _lib.current_stream(dev))
        if need_grad:
            ctx.save_for_backward(x, arena, *weights)
            ctx.meta = (first_stride, input_scale, precision, n_layers, ptrs[:-1], max(numel), amax, bits)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        first_stride, input_scale, precision, n_layers, act_ptrs, max_numel, amax, bits = ctx.meta
        x, arena = ctx.saved_tensors[:2]
        weights = ctx.saved_tensors[2:]
        B, C, H, W = x.shape
        dev = x.device
        bf16 = precision == "bf16"
        g = grad_out.contiguous()
        dws = [torch.empty_like(w) for w in weights]
        dbs = torch.empty((n_layers, 32), dtype=torch.float32, device=dev)
        half_n = (max_numel + 7) // 8 * 8
        dz = torch.empty(2 * half_n + (_AMAX if amax else 0), dtype=torch.bfloat16 if bf16 else torch.float32, device=dev)
        half = half_n * dz.element_size()
        ws32, wsf = _workspace(dev, 32, max(1, n_layers - 1)), _workspace(dev, C)
        fws = _frag_ws(dev, n_layers - 1) if (precision in _SPLIT or (bf16 and _CHAINS)) and n_layers > 1 else None
        dzc, dzc_bytes = None, 0
        if bf16 and _CHAINS and _CHAINS_BWD and n_layers > 1:  # every layer's gradient map, kept for the backward-weights launches (one-launch backward-data)
            dzc_bytes = _lib.load().augcl_enc_stack_backward_chain_bytes(n_layers, B, C, H, W, first_stride)
            dzc = torch.empty(max(dzc_bytes, 16), dtype=torch.uint8, device=dev) if dzc_bytes else None
        _lib.call("augcl_enc_stack_backward", _ptr_array([x.data_ptr()] + list(act_ptrs)),
                  _ptr_array([w.data_ptr() for w in weights]), g.data_ptr(), _ptr_array([t.data_ptr() for t in dws]),
                  _ptr_array([dbs[i].data_ptr() for i in range(n_layers)]), dz.data_ptr(), dz.data_ptr() + half,
                  ws32.data_ptr(), ws32.numel() * 4, wsf.data_ptr(), wsf.numel() * 4, n_layers, B, C, H, W, first_stride,
                  input_scale, PRECISION_CODE[precision], 1 if _BF16_WGRAD.endswith("_mfma") else 0, _lib.ptr(fws),
                  fws.numel() if fws is not None else 0, amax, dz.data_ptr() + 2 * half if amax else None, bits,
                  _lib.ptr(dzc), dzc_bytes if dzc is not None else 0, _lib.current_stream(dev))
        return (None, None, None, None, None, *dws, *[dbs[i] for i in range(n_layers)])
