"""autograd node for the whole conv stack on the hand-written HIP encoder kernels (include/augcl.h,
augcl_enc_*).  Activations are stored post-ReLU (the forward kernels fuse bias + the following ReLU);
backward walks the stack once: backward-weights from (a_{l-1}, dz_l), backward-data with the ReLU mask
of the layer below fused in.  No gradient is produced for the image input (never needed on this path)."""
import torch

from . import _lib

_ws = {}


def _workspace(dev, c_in):
    key = (dev, c_in)
    if key not in _ws:
        n = _lib.load().augcl_enc_wgrad_workspace_bytes(int(c_in))
        _ws[key] = torch.empty(n // 4, dtype=torch.float32, device=dev)
    return _ws[key]


def _out_hw(h, stride):
    return (h - 3) // stride + 1


class ConvStackFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, first_stride, input_scale, relu_first, n_layers, *wb):
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
                call("augcl_enc_conv32_forward", p(acts[-1]), p(wi), p(bi), p(out), B, h, w_, relu, st)
            if need_grad or i == n_layers - 1:
                acts.append(out)
            else:
                acts = [acts[0], out] if i == 0 else [acts[0], out]  # keep only what the next layer reads
            h, w_ = ho, wo
        out = acts[-1]
        if need_grad:
            ctx.save_for_backward(*acts[:-1], *[t.detach() for t in weights])
            ctx.meta = (first_stride, input_scale, n_layers, tuple(x.shape), bool(relu_first))
        return out

    @staticmethod
    def backward(ctx, grad_out):
        first_stride, input_scale, n_layers, xshape, relu_first = ctx.meta
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
                call("augcl_enc_conv32_backward_weights", p(a_prev), p(dz), p(dw), p(db), p(ws), ws.numel() * 4, B,
                     a_prev.shape[2], a_prev.shape[3], st)
            dws[i], dbs[i] = dw, db
            if i > 0:
                dact = torch.empty_like(a_prev)
                call("augcl_enc_conv32_backward_data", p(dz), p(wi), p(a_prev), p(dact), B, dz.shape[2], dz.shape[3], st)
                dz = dact
            elif want_dx:
                if first:
                    raise NotImplementedError("gradient w.r.t. the image input of the strided first layer is not "
                                              "needed on the SAC path and not implemented")
                dx = torch.empty_like(a_prev)  # HeadCNN: input is the shared CNN's (pre-ReLU) output
                call("augcl_enc_conv32_backward_data", p(dz), p(wi), p(a_prev) if relu_first else None, p(dx), B,
                     dz.shape[2], dz.shape[3], st)
        return (dx, None, None, None, None, *dws, *dbs)


class ConvStackBf16Fn(torch.autograd.Function):
    """bf16 PERFORMANCE MODE of the same stack (augcl_enc_bf16_*): activations bf16 NHWC in HBM, bf16 MFMA with
    f32 accumulation, f32 master weights and f32 weight gradients.  Not the parity path (bf16 resolution)."""

    @staticmethod
    def forward(ctx, x, first_stride, input_scale, relu_first, n_layers, *wb):
        weights, biases = wb[:n_layers], wb[n_layers:]
        assert not relu_first, "bf16 mode covers the SharedCNN stack (image input)"
        assert x.dtype == torch.float32 and x.is_contiguous()
        B, C, H, W = x.shape
        dev = x.device
        st = _lib.current_stream(dev)
        call, p = _lib.call, _lib.ptr
        need_grad = any(t.requires_grad for t in wb)
        acts = [x]
        h, w_ = _out_hw(H, first_stride), _out_hw(W, first_stride)
        w0, b0 = weights[0].detach(), biases[0].detach()
        cur = torch.empty((B, h, w_, 32), dtype=torch.bfloat16, device=dev)
        call("augcl_enc_bf16_conv_first_forward", p(x), p(w0), p(b0), p(cur), B, C, H, W, first_stride, input_scale,
             1 if n_layers > 1 else 0, st)
        for i in range(1, n_layers):
            wi, bi = weights[i].detach(), biases[i].detach()
            last = i == n_layers - 1
            ho, wo = h - 2, w_ - 2
            out = (torch.empty((B, 32, ho, wo), dtype=torch.float32, device=dev) if last
                   else torch.empty((B, ho, wo, 32), dtype=torch.bfloat16, device=dev))
            call("augcl_enc_bf16_conv32_forward", p(cur), p(wi), p(bi), p(out), B, h, w_, 0 if last else 1,
                 1 if last else 0, st)
            if need_grad:
                acts.append(cur)
            cur, h, w_ = out, ho, wo
        if n_layers == 1:
            raise NotImplementedError("bf16 mode needs at least two conv layers")
        if need_grad:
            ctx.save_for_backward(*acts, *[t.detach() for t in weights])
            ctx.meta = (first_stride, input_scale, n_layers)
        return cur

    @staticmethod
    def backward(ctx, grad_out):
        first_stride, input_scale, n_layers = ctx.meta
        saved = ctx.saved_tensors
        acts, weights = saved[:n_layers], saved[n_layers:]  # acts[0] = x (f32 NCHW), acts[i>0] = a_i (bf16 NHWC)
        x = acts[0]
        B = x.shape[0]
        dev = grad_out.device
        st = _lib.current_stream(dev)
        call, p = _lib.call, _lib.ptr
        g = grad_out.contiguous()
        dz = torch.empty((B, g.shape[2], g.shape[3], 32), dtype=torch.bfloat16, device=dev)
        call("augcl_enc_nchw_f32_to_nhwc_bf16", p(g), p(dz), B, g.shape[2], g.shape[3], st)
        dws, dbs = [None] * n_layers, [None] * n_layers
        ws32 = _workspace(dev, 32)
        for i in range(n_layers - 1, 0, -1):
            a_prev, wi = acts[i], weights[i]  # (B,Hi,Wi,32) bf16
            dw = torch.empty_like(wi)
            db = torch.empty(32, dtype=torch.float32, device=dev)
            call("augcl_enc_bf16_conv32_backward_weights", p(a_prev), p(dz), p(dw), p(db), p(ws32), ws32.numel() * 4, B,
                 a_prev.shape[1], a_prev.shape[2], st)
            dws[i], dbs[i] = dw, db
            dact = torch.empty_like(a_prev)
            call("augcl_enc_bf16_conv32_backward_data", p(dz), p(wi), p(a_prev), p(dact), B, dz.shape[1], dz.shape[2], st)
            dz = dact
        w0 = weights[0]
        dw = torch.empty_like(w0)
        db = torch.empty(32, dtype=torch.float32, device=dev)
        ws = _workspace(dev, w0.shape[1])
        call("augcl_enc_bf16_conv_first_backward_weights", p(x), p(dz), p(dw), p(db), p(ws), ws.numel() * 4, B, w0.shape[1],
             x.shape[2], x.shape[3], first_stride, input_scale, st)
        dws[0], dbs[0] = dw, db
        return (None, None, None, None, None, *dws, *dbs)
