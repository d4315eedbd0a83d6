"""GPU parity where training actually runs: BASELINE's batch sizes (256 / 512 / 1024), the reference's real initial
state (delta-orthogonal convs, zero bias: src/algorithms/modules.py:52-66) and He-scaled weights -- NOT the gate-safe
nets of the golden fixtures -- with a float64 referee.

Three evaluations of the same computation: `f64` (oracle/sac_oracle.py run in float64 = the referee), `cpu32` (the same
oracle in float32 = what the reference computes on a CPU) and `hip` (the product).  Criterion (the one
tests/test_heads_gpu.py uses for the fused heads): the HIP path must be as close to the referee as the float32 oracle is.

ReLU's gradient is discontinuous at 0, so two correct float32 evaluations can gate a pre-activation that sits within
rounding of zero differently; one such flip moves the weight gradients below it by ~1/sqrt(#pixels).  The encoder tests
therefore ALSO (a) locate every gate that differs between HIP and the referee and require its referee pre-activation
to be ~0, (b) run the HIP backward on the referee's activations and (c) run the referee with HIP's gates forced: with
identical gates the HIP gradients must match float64 to <= 1e-5 -- the residual is gates and nothing else.
Results are also written to gpurun_out/referee_*.json when that directory exists."""
import json
import os
import time

import numpy as np
import pytest
import torch
import torch.nn.functional as F

import test_oracle_golden_sac as T
from augcl_amd import _lib, encoder_hip, encoder_ops
from augcl_amd import augmentations as aug
from augcl_amd import utils as au
from augcl_amd.algorithms import make_agent
from oracle import sac_oracle as so

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OBS = T.OBS
STRICT = 1e-5  # north_star tolerance


def rel_l2(a, b):
    a, b = a.detach().double().cpu().flatten(), b.detach().double().cpu().flatten()
    return float((a - b).norm() / b.norm().clamp_min(1e-300))


def pooled(pairs):
    """relative L2 over a group of tensors taken as one vector"""
    num = sum(float((a.detach().double().cpu() - b.detach().double().cpu()).square().sum()) for a, b in pairs)
    den = sum(float(b.detach().double().cpu().square().sum()) for _, b in pairs)
    return (num / max(den, 1e-300)) ** 0.5


def dump(name, obj):
    d = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(d):
        with open(os.path.join(d, f"referee_{name}.json"), "w") as f:
            json.dump(obj, f, indent=1)


_REF_DEV = {}


def ref_device(cuda):
    """Where the float64 referee runs: the GPU's f64 units through plain ATen ops when this torch build has a double
    convolution there (seconds instead of minutes at B=1024), else the host.  Independent of augcl_amd either way."""
    if "dev" not in _REF_DEV:
        try:
            x = torch.randn(2, 4, 9, 9, dtype=torch.float64, device=cuda, requires_grad=True)
            w = torch.randn(5, 4, 3, 3, dtype=torch.float64, device=cuda, requires_grad=True)
            y = F.conv2d(x, w, stride=2)
            y.sum().backward()
            ok = torch.allclose(y.detach().cpu(), F.conv2d(x.detach().cpu(), w.detach().cpu(), stride=2), rtol=1e-12, atol=1e-12)
            _REF_DEV["dev"] = cuda if ok else torch.device("cpu")
        except Exception:  # noqa: BLE001
            _REF_DEV["dev"] = torch.device("cpu")
    return _REF_DEV["dev"]


def make_states(init, A, seed):
    hp = dict(so.DEFAULTS)
    mk = (lambda s, shp: so.synth_state(s, shp, gate_safe=False)) if init == "he" else so.real_init_state
    cs = mk(seed, so.critic_shapes(OBS, A, hp))
    as_ = mk(seed + 1, so.actor_shapes(OBS, A, hp))
    for k in cs:
        if so.is_cnn(k):
            as_[k] = cs[k]
    return as_, cs


# ------------------------------------------------------------------------------------------------ encoder
class _Gated(torch.autograd.Function):
    """ReLU with the gate given from outside: y = z * gate, dz = dy * gate."""

    @staticmethod
    def forward(ctx, z, gate):
        ctx.save_for_backward(gate)
        return z * gate

    @staticmethod
    def backward(ctx, g):
        (gate,) = ctx.saved_tensors
        return g * gate, None


def ref_stack(x, ws, bs, gates=None):
    """x/255 -> conv s2 -> [ReLU -> conv]*; returns (output, pre-activations of every ReLU)."""
    h = x / 255.0
    pre = []
    for i, (w, b) in enumerate(zip(ws, bs)):
        if i:
            pre.append(h)
            h = torch.relu(h) if gates is None else _Gated.apply(h, gates[i - 1].to(h.dtype))
        h = F.conv2d(h, w, b, stride=2 if i == 0 else 1)
    return h, pre


def ref_run(x, ws, bs, gout, dtype, gates=None, device="cpu"):
    to = lambda t: t.to(device=device, dtype=dtype)  # noqa: E731
    w = [to(t).clone().requires_grad_(True) for t in ws]
    b = [to(t).clone().requires_grad_(True) for t in bs]
    out, pre = ref_stack(to(x), w, b, None if gates is None else [g.to(device) for g in gates])
    (out * to(gout)).sum().backward()
    c = lambda t: t.detach().cpu()  # noqa: E731
    return c(out), [c(t) for t in pre], [c(t.grad) for t in w], [c(t.grad) for t in b]


def _ptrs(ts):
    return (_lib.ctypes.c_void_p * len(ts))(*[t.data_ptr() for t in ts])


def hip_forward(x, ws, bs):
    """augcl_enc_stack_forward through the C ABI; returns the output of every layer (post-ReLU but the last)."""
    B, C, H, W = x.shape
    n = len(ws)
    hs = [(H - 3) // 2 + 1]
    for _ in range(1, n):
        hs.append(hs[-1] - 2)
    acts = [torch.empty((B, 32, h, h), dtype=torch.float32, device=x.device) for h in hs]
    fws = encoder_hip._frag_ws(x.device, n - 1)
    amax = torch.empty(n, dtype=torch.float32, device=x.device)  # precision "f32": max |output| of every layer
    if (encoder_ops.PRECISION == "f32" and n > 1 and encoder_hip._CHAINS and encoder_hip._CHAINS_GRAD
            and B <= encoder_hip._CHAINS_GRAD_MAX_B):
        # what the product's gradient-carrying forward runs at this batch size (round 4): the 32 -> 32 layers as chains that store
        # every layer; per-item scales, so an image's values do not depend on its position in the batch
        rc = _lib.load().augcl_enc_stack_forward_chained(x.data_ptr(), _ptrs(ws), _ptrs(bs), _ptrs(acts), n, B, C, H, W, 2, 255.0,
                                                         fws.data_ptr(), fws.numel(), amax.data_ptr(), None, _lib.current_stream(x.device))
        if rc == 0:
            want = torch.stack([a.abs().max() for a in acts[:-1]])
            assert bool(((amax[:-1] >= want * (1 - 1e-6)) & (amax[:-1] <= want * (1 + 1e-6))).all())
            return acts
        assert rc == _lib.AUGCL_E_UNSUPPORTED
    _lib.call("augcl_enc_stack_forward", x.data_ptr(), _ptrs(ws), _ptrs(bs), _ptrs(acts), n, B, C, H, W, 2, 255.0,
              encoder_hip.PRECISION_CODE[encoder_ops.PRECISION], fws.data_ptr(), fws.numel(), amax.data_ptr(), None,
              _lib.current_stream(x.device))
    if encoder_ops.PRECISION == "f32":  # the kernels' maxima are the tensors' maxima, exactly
        assert torch.equal(amax[:-1], torch.stack([a.abs().max() for a in acts[:-1]]))
    return acts


def node_acts(out, like):
    """the post-ReLU maps the product's autograd node kept for its backward pass (float32 precisions), read from its arena.
    Round 4: with at most one image per CU the gradient-carrying forward of precision "f32" runs the 32 -> 32 layers as chains that
    store every layer (augcl_enc_stack_forward_chained) -- its values, and a handful of its near-zero gates, are not the
    per-layer entry point's: whatever is forced on the referee must be what THIS forward took."""
    act_ptrs, arena = out.grad_fn.meta[4], out.grad_fn.saved_tensors[1]
    res = []
    for ptr, a in zip(act_ptrs, like):
        o = (ptr - arena.data_ptr()) // 4
        res.append(arena[o:o + a.numel()].view(a.shape))
    return res


def hip_backward(x, acts_in, ws, gout):
    """augcl_enc_stack_backward through the C ABI on GIVEN layer inputs (acts_in[0] = x raw, then post-ReLU maps)."""
    B, C, H, W = x.shape
    n = len(ws)
    dev = x.device
    dws = [torch.empty_like(w) for w in ws]
    dbs = [torch.empty(32, dtype=torch.float32, device=dev) for _ in ws]
    big = (max(t.numel() for t in acts_in[1:]) + 7) // 8 * 8
    dz = torch.empty(2 * big, dtype=torch.float32, device=dev)
    ws32, wsf = encoder_hip._workspace(dev, 32, max(1, n - 1)), encoder_hip._workspace(dev, C)
    x2 = encoder_ops.PRECISION == "f32"  # the f16-pair kernels need the fragment workspace and the layer inputs' maxima
    fws = encoder_hip._frag_ws(dev, n - 1) if x2 else None  # ("f32_x3" without it: the in-kernel split path)
    amax_acts = torch.stack([a.abs().max() for a in acts_in[1:]] + [torch.zeros((), device=dev)]).float() if x2 else None
    amax_dz = torch.empty(n, dtype=torch.float32, device=dev) if x2 else None
    _lib.call("augcl_enc_stack_backward", _ptrs(acts_in), _ptrs(ws), gout.data_ptr(), _ptrs(dws), _ptrs(dbs), dz.data_ptr(),
              dz.data_ptr() + 4 * big, ws32.data_ptr(), ws32.numel() * 4, wsf.data_ptr(), wsf.numel() * 4, n, B, C, H, W, 2,
              255.0, encoder_hip.PRECISION_CODE[encoder_ops.PRECISION], 0, _lib.ptr(fws), fws.numel() if x2 else 0,
              _lib.ptr(amax_acts), _lib.ptr(amax_dz), None, None, 0, _lib.current_stream(dev))
    return dws, dbs


@pytest.mark.parametrize("B,init,precision", [(8, "he", "f32"), (8, "delta", "f32"), (256, "he", "f32"), (256, "delta", "f32"),
                                              (512, "he", "f32"), (8, "he", "f32_x3"), (256, "he", "f32_x3"),
                                              (8, "he", "f32_mfma"), (256, "he", "f32_mfma")])
def test_encoder_against_f64_referee(cuda, monkeypatch, B, init, precision):
    """precision "f32" = the parity path (32 -> 32 layers: float32-accurate products as f16 pairs on the f16 MFMA pipe,
    csrc/encoder_x2.hip); "f32_x3" = three-way bf16 splits (csrc/encoder_x3.hip); "f32_mfma" = the native f32 MFMA kernels;
    all held to the same bounds."""
    monkeypatch.setattr(encoder_ops, "PRECISION", precision)
    t0 = time.time()
    _, cs = make_states(init, 6, 11)
    ids = so.conv_layer_ids(11)
    ws = [cs[f"encoder.shared_cnn.layers.{i}.weight"] for i in ids]
    bs = [cs[f"encoder.shared_cnn.layers.{i}.bias"] for i in ids]
    rs = np.random.RandomState(1000 + B)
    x = torch.from_numpy(rs.randint(0, 256, (B, 9, 84, 84)).astype(np.float32))
    gout = torch.from_numpy(rs.standard_normal((B, 32, 21, 21)).astype(np.float32))

    rdev = ref_device(cuda)
    out64, pre64, dw64, db64 = ref_run(x, ws, bs, gout, torch.float64, device=rdev)
    out32, pre32, dw32, db32 = ref_run(x, ws, bs, gout, torch.float32)

    xd, gd = x.to(cuda), gout.to(cuda)
    wd, bd = [w.to(cuda) for w in ws], [b.to(cuda) for b in bs]
    acts = hip_forward(xd, wd, bd)
    rep = {"B": B, "init": init, "precision": precision, "referee_device": str(rdev)}

    # (1) forward: continuous, so the plain criterion applies
    e_hip, e_cpu = rel_l2(acts[-1], out64), rel_l2(out32, out64)
    rep["fwd"] = {"hip": e_hip, "cpu32": e_cpu}
    assert e_hip <= max(2.0 * e_cpu, 1e-6), rep["fwd"]

    # (2) gates: HIP stores post-ReLU maps; a gate differs from the referee's only where the referee's
    #     pre-activation is ~0 (relative to the layer's rms), and only a handful do
    gates_hip, flips, worst = [], 0, 0.0
    for a, z in zip(acts[:-1], pre64):
        g = (a > 0).cpu()
        gates_hip.append(g)
        diff = g != (z > 0)
        n = int(diff.sum())
        flips += n
        if n:
            worst = max(worst, float(z[diff].abs().max() / z.square().mean().sqrt()))
    n_pre = sum(z.numel() for z in pre64)
    flips_cpu = sum(int(((a > 0) != (z > 0)).sum()) for a, z in zip(pre32, pre64))
    rep["gates"] = {"differ": flips, "of": n_pre, "max_|z|/rms_at_differing": worst, "cpu32_differ": flips_cpu}
    assert worst <= 1e-5 and flips <= 5e-6 * n_pre + 10, rep["gates"]

    # (3) HIP backward on the REFEREE's activations (identical gates): pure arithmetic, must be <= 1e-5
    acts_ref = [xd] + [torch.relu(z).float().to(cuda) for z in pre64]
    dws, dbs = hip_backward(xd, acts_ref, wd, gd)
    torch.cuda.synchronize()
    rep["bwd_referee_gates"] = {"dW": [rel_l2(a, b) for a, b in zip(dws, dw64)], "db": [rel_l2(a, b) for a, b in zip(dbs, db64)]}
    assert max(rep["bwd_referee_gates"]["dW"] + rep["bwd_referee_gates"]["db"]) <= STRICT, rep["bwd_referee_gates"]

    # (4) the product path (one autograd node, its own gates) against the referee run with HIP's gates forced
    wg = [w.clone().requires_grad_(True) for w in wd]
    bg = [b.clone().requires_grad_(True) for b in bd]
    out = encoder_ops.conv_stack(xd, wg, bg, first_stride=2, input_scale=255.0)
    kept = [a.clone() for a in node_acts(out, acts[:-1])]  # (the node frees its arena in backward)
    (out * gd).sum().backward()
    torch.cuda.synchronize()
    e_prod = rel_l2(out.detach(), out64)
    rep["fwd"]["product"] = e_prod
    assert e_prod <= max(2.0 * e_cpu, 1e-6), rep["fwd"]
    gates_hip, flips_p, worst_p = [], 0, 0.0  # the product forward's own gates, under the same audit as (2)
    for a, z in zip(kept, pre64):
        g = (a > 0).cpu()
        gates_hip.append(g)
        diff = g != (z > 0)
        if int(diff.sum()):
            flips_p += int(diff.sum())
            worst_p = max(worst_p, float(z[diff].abs().max() / z.square().mean().sqrt()))
    rep["gates"]["product_differ"], rep["gates"]["product_max_|z|/rms_at_differing"] = flips_p, worst_p
    assert worst_p <= 1e-5 and flips_p <= 5e-6 * n_pre + 10, rep["gates"]
    _, _, dw64g, db64g = ref_run(x, ws, bs, gout, torch.float64, gates=gates_hip, device=rdev)
    rep["bwd_hip_gates_forced"] = {"dW": [rel_l2(a.grad, b) for a, b in zip(wg, dw64g)],
                                   "db": [rel_l2(a.grad, b) for a, b in zip(bg, db64g)]}
    assert max(rep["bwd_hip_gates_forced"]["dW"] + rep["bwd_hip_gates_forced"]["db"]) <= STRICT, rep["bwd_hip_gates_forced"]

    # (5) un-forced: both float32 evaluations against the referee, per layer and pooled over the stack
    rep["bwd_free"] = {"hip_dW": [rel_l2(a.grad, b) for a, b in zip(wg, dw64)], "cpu32_dW": [rel_l2(a, b) for a, b in zip(dw32, dw64)],
                       "hip_pooled": pooled([(a.grad, b) for a, b in zip(wg + bg, dw64 + db64)]),
                       "cpu32_pooled": pooled(list(zip(dw32 + db32, dw64 + db64)))}
    rep["seconds"] = time.time() - t0
    dump(f"encoder_B{B}_{init}_{precision}", rep)
    # The plain criterion only means something when BOTH float32 evaluations crossed a fair number of near-zero gates
    # (each differing gate is a discrete ~1/sqrt(#pixels) event; with 3 on one side and 0 on the other -- B=256 on the
    # reference's own init -- the ratio says nothing, and (3)/(4) above are the rigorous statement).
    if min(flips, flips_cpu) >= 5:
        assert rep["bwd_free"]["hip_pooled"] <= 2.0 * rep["bwd_free"]["cpu32_pooled"] + 1e-6, rep["bwd_free"]
    assert rep["bwd_free"]["hip_pooled"] <= 1e-2, rep["bwd_free"]


def test_stack_with_one_outlier_sample_against_f64_referee(cuda, monkeypatch):
    """One image of the batch 2^20 brighter than its companions (its activations set every per-tensor scale of the f16-pair
    kernels, the others sit 2^20 below the tensors' maxima) and its incoming gradient 2^-20 of theirs (so that every sample
    still carries a comparable share of dW): the whole stack, forward per sample on each sample's own scale -- the
    gradient-carrying per-layer kernels and the no-grad layer chains -- and dW / db with the gates of the HIP run forced, against
    the float64 referee.  (Single layers on degenerate inputs: test_encoder_gpu.py; the numerical method's range limits:
    test_f16_pair_numerics.py.)"""
    monkeypatch.setattr(encoder_ops, "PRECISION", "f32")
    B, k = 8, 3
    _, cs = make_states("he", 6, 11)
    ids = so.conv_layer_ids(11)
    ws = [cs[f"encoder.shared_cnn.layers.{i}.weight"] for i in ids]
    bs = [torch.zeros_like(cs[f"encoder.shared_cnn.layers.{i}.bias"]) for i in ids]  # no bias: a sample's activations scale with its image
    rs = np.random.RandomState(77)
    x = torch.from_numpy(rs.randint(0, 256, (B, 9, 84, 84)).astype(np.float32))
    gout = torch.from_numpy(rs.standard_normal((B, 32, 21, 21)).astype(np.float32))
    x[k] *= 2.0 ** 20
    gout[k] *= 2.0 ** -20
    rdev = ref_device(cuda)
    out64, pre64, _, _ = ref_run(x, ws, bs, gout, torch.float64, device=rdev)
    xd, gd = x.to(cuda), gout.to(cuda)
    wg = [w.to(cuda).clone().requires_grad_(True) for w in ws]
    bg = [b.to(cuda).clone().requires_grad_(True) for b in bs]
    out = encoder_ops.conv_stack(xd, wg, bg, first_stride=2, input_scale=255.0)
    # the gates of the HIP run: the post-ReLU maps its autograd node kept (read before backward frees them)
    gates_hip = [(a > 0).cpu() for a in node_acts(out, [torch.empty((B, 32, 41 - 2 * i, 41 - 2 * i), device="meta") for i in range(10)])]
    (out * gd).sum().backward()
    with torch.no_grad():
        chain = encoder_ops.conv_stack(xd, [w.detach() for w in wg], [b.detach() for b in bg], first_stride=2, input_scale=255.0)
    torch.cuda.synchronize()
    rep = {"per_sample_fwd": [rel_l2(out[i].detach(), out64[i]) for i in range(B)],
           "per_sample_fwd_chains": [rel_l2(chain[i], out64[i]) for i in range(B)]}
    assert max(rep["per_sample_fwd"]) <= STRICT and max(rep["per_sample_fwd_chains"]) <= STRICT, rep
    _, _, dw64g, db64g = ref_run(x, ws, bs, gout, torch.float64, gates=gates_hip, device=rdev)
    rep["dW"] = [rel_l2(a.grad, b) for a, b in zip(wg, dw64g)]
    rep["db"] = [rel_l2(a.grad, b) for a, b in zip(bg, db64g)]
    dump("encoder_outlier_sample_B8_he_f32", rep)
    assert max(rep["dW"] + rep["db"]) <= STRICT, rep


def test_bf16_encoder_against_f64_referee(cuda):
    """The bf16 PERFORMANCE mode of the conv stack (BASELINE config 2's bf16 leg; not the parity path) at the config's batch
    size, He-scaled weights, against the float64 referee: every layer's stored activation (bf16 NHWC) and, with the gates of
    the bf16 run forced in the referee (bf16 noise flips near-zero gates; what is measured is the arithmetic), every dW / db.
    Measured (MI355X): relative L2 of the stored activations 2.8e-3 (layer 0) growing to 1.04e-2 (output) -- bf16 operands,
    f32 accumulation, one bf16 rounding of the activation per layer; dW 5.6e-3 .. 9.5e-3, db 1.0e-3 .. 6.0e-3; 0.15 % of the
    86.6 M ReLU gates differ from the referee's.  Required: <= 2e-2 everywhere."""
    B = 256
    _, cs = make_states("he", 6, 11)
    ids = so.conv_layer_ids(11)
    ws = [cs[f"encoder.shared_cnn.layers.{i}.weight"] for i in ids]
    bs = [cs[f"encoder.shared_cnn.layers.{i}.bias"] for i in ids]
    rs = np.random.RandomState(1000 + B)
    x = torch.from_numpy(rs.randint(0, 256, (B, 9, 84, 84)).astype(np.float32))
    gout = torch.from_numpy(rs.standard_normal((B, 32, 21, 21)).astype(np.float32))
    rdev = ref_device(cuda)
    out64, pre64, _, _ = ref_run(x, ws, bs, gout, torch.float64, device=rdev)
    xd, gd = x.to(cuda), gout.to(cuda)
    wd, bd = [w.to(cuda) for w in ws], [b.to(cuda) for b in bs]
    # per-layer activations through the C ABI (precision code 1): bf16 NHWC for layers 0..9, f32 NCHW for the last
    n = len(wd)
    hs = [41 - 2 * i for i in range(n)]
    acts = [torch.empty((B, h, h, 32), dtype=torch.bfloat16, device=cuda) for h in hs[:-1]] + [torch.empty((B, 32, 21, 21), device=cuda)]
    _lib.call("augcl_enc_stack_forward", xd.data_ptr(), _ptrs(wd), _ptrs(bd), _ptrs(acts), n, B, 9, 84, 84, 2, 255.0, 1,
              None, 0, None, None, _lib.current_stream(cuda))
    torch.cuda.synchronize()
    fwd = [rel_l2(a.permute(0, 3, 1, 2).float(), torch.relu(z)) for a, z in zip(acts[:-1], pre64)] + [rel_l2(acts[-1], out64)]
    gates_hip = [(a.permute(0, 3, 1, 2) > 0).cpu() for a in acts[:-1]]
    # the product path (autograd node) in bf16 mode
    wg = [w.clone().requires_grad_(True) for w in wd]
    bg = [b.clone().requires_grad_(True) for b in bd]
    out = encoder_ops.conv_stack(xd, wg, bg, first_stride=2, input_scale=255.0, precision="bf16")
    (out * gd).sum().backward()
    torch.cuda.synchronize()
    assert torch.equal(out.detach(), acts[-1])
    _, _, dw64g, db64g = ref_run(x, ws, bs, gout, torch.float64, gates=gates_hip, device=rdev)
    dW = [rel_l2(a.grad, b) for a, b in zip(wg, dw64g)]
    dB = [rel_l2(a.grad, b) for a, b in zip(bg, db64g)]
    flips = sum(int((g != (z > 0)).sum()) for g, z in zip(gates_hip, pre64))
    rep = {"B": B, "fwd_rel_l2_per_layer": fwd, "dW_vs_forced_referee": dW, "db_vs_forced_referee": dB,
           "gates_differ": flips, "of": sum(z.numel() for z in pre64)}
    dump("encoder_bf16_B256_he", rep)
    assert max(fwd) <= 2e-2, rep
    assert max(dW + dB) <= 2e-2, rep
    assert flips <= 5e-3 * rep["of"], rep


# ------------------------------------------------------------------------------------------------ update()
class RecLogger:
    def __init__(self):
        self.rows = {}

    def log(self, key, value, step, n=1):
        self.rows[key] = float(value.item() if torch.is_tensor(value) else value)


CASES = [  # (algo, data_aug, A, B, hp overrides, init): BASELINE configs 2, 3, 4 (config 4's global batch on one GPU) + DrQ
    # K=M=2; "delta" = the reference's real initial state, "he" = He-scaled weights (a net in mid-training)
    ("svea", "conv", 6, 256, {}, "he"),
    ("svea", "conv", 6, 256, {}, "delta"),
    ("drq", "identity", 2, 512, {}, "he"),
    ("drq", "identity", 2, 512, {}, "delta"),
    ("augcl", "conv-overlay", 6, 1024, {}, "he"),
    ("drq", "identity", 2, 128, dict(drq_k=2, drq_m=2), "he"),
]


def group_of(name):
    if so.is_cnn(name):
        return "cnn"
    if "projection" in name:
        return "projection"
    return name.split(".")[0]  # Q1 / Q2 / mlp


@pytest.mark.parametrize("algo,data_aug,A,B,hpo,init", CASES)
def test_update_against_f64_referee(cuda, monkeypatch, algo, data_aug, A, B, hpo, init):
    """One full update() (critic, actor + alpha, target EMA) at the configs' batch sizes, on real-init / He-scaled nets.
    The referee is run twice: FREE (its own ReLU gates) -- reported next to the float32 oracle's deviation, which shows
    how far two correct float32 evaluations drift apart through a handful of near-zero gates -- and with the gates of
    the HIP run FORCED, where every logged loss, every gradient tensor and the target network must agree to 1e-5."""
    t0 = time.time()
    hp = dict(so.DEFAULTS, **hpo)
    K, M = (hp["drq_k"], hp["drq_m"]) if algo == "drq" else (1, 1)
    as_, cs = make_states(init, A, 21)
    rs = np.random.RandomState(B + A)
    obs = rs.randint(0, 256, (B, *OBS)).astype(np.float32)
    nobs = rs.randint(0, 256, (B, *OBS)).astype(np.float32)
    act = rs.uniform(-1, 1, (B, A)).astype(np.float32)
    rew = rs.uniform(0, 4, (B, 1)).astype(np.float32)
    nd = np.ones((B, 1), np.float32)
    noises = [torch.from_numpy(rs.standard_normal((B, A)).astype(np.float32)) for _ in range(M + 1)]
    draws = [(rs.randint(0, 9, B).astype(np.int32), rs.randint(0, 9, B).astype(np.int32)) for _ in range(K + M)]
    cw = rs.standard_normal((B, 3, 3, 3, 3)).astype(np.float32)
    ov = rs.random_sample((B, 3, 84, 84)).astype(np.float32)
    tt = torch.from_numpy
    d_obs, d_nobs = tt(obs).to(cuda), tt(nobs).to(cuda)

    # the augmented batches are produced ONCE by the HIP operators (they have their own parity tests) and handed to
    # every evaluation, so that what is compared is the update and nothing else
    def strong(x):
        for k in data_aug.split("-"):
            if k == "conv":
                x = aug.random_conv(x, weights=cw)
            elif k == "overlay":
                x = aug.random_overlay(x, imgs=tt(ov).to(cuda))
        return x

    shifted = [aug.random_shift(d_obs, oy=oy, ox=ox) for oy, ox in draws[:K]]
    shifted_next = [aug.random_shift(d_nobs, oy=oy, ox=ox) for oy, ox in draws[K:]] if algo == "drq" else []
    strong_d = strong(shifted[0]) if algo == "svea" else strong(d_obs) if algo == "augcl" else None  # augcl.py:72: un-shifted
    torch.cuda.synchronize()
    # inputs of the gradient-carrying critic passes, in the oracle's order
    critic_inputs = {"svea": [shifted[0], strong_d], "drq": shifted, "augcl": [strong_d]}[algo]
    weak_inputs = [shifted[0]] if algo == "augcl" else []
    rdev = ref_device(cuda)

    def oracle_run(dtype, forced=None, after_critic=None):
        dev = rdev if dtype == torch.float64 else torch.device("cpu")
        c = lambda t: t.detach().to(device=dev, dtype=dtype)  # noqa: E731
        mk = lambda t: (lambda x: c(t))  # noqa: E731
        if algo == "drq":
            kw = dict(shift=[mk(t) for t in shifted], shift2=[mk(t) for t in shifted_next])
        else:
            kw = dict(shift=mk(shifted[0]), aug=mk(strong_d))
        ag = so.OracleAgent(algo, OBS, A, as_, cs, hp=hp, dtype=dtype, device=dev)
        batch = tuple(c(tt(v)) for v in (obs, act, rew, nobs, nd))
        nt = [c(n) for n in noises[:M]]
        log = ag.update(batch, 0, nt if algo == "drq" and M > 1 else nt[0], c(noises[M]), forced=forced,
                        after_critic=after_critic, **kw)
        for d in (ag.critic, ag.actor, ag.critic_target, getattr(ag, "critic_weak", {})):  # results back on the host
            for k in d:
                d[k] = d[k].detach().cpu()
        ag.rec = {k: ({n: t.cpu() for n, t in v.items()} if isinstance(v, dict) else v.cpu()) for k, v in ag.rec.items()}
        return ag, log

    # ---- the product
    args = au.dotdict(algorithm=algo, data_aug=data_aug, aug_params=None, **hp)

    def load(agent):
        csd = {k: v.to(cuda) for k, v in cs.items()}
        agent.critic.load_state_dict(csd)
        agent.actor.load_state_dict({k: (csd[k] if so.is_cnn(k) else v.to(cuda)) for k, v in as_.items()})
        agent.critic_target.load_state_dict(csd)
        return agent

    if algo == "augcl":
        agent = make_agent(OBS, (A,), args)
        agent.load_pretrained_agent(load(make_agent(OBS, (A,), au.dotdict(dict(args, algorithm="non_naive_rad", data_aug="shift")))))
    else:
        agent = load(make_agent(OBS, (A,), args))
    agent.fuse_front_end = False

    # gates of the HIP run, CNN part: the encoder forward through the C ABI on the very tensors update() will see, batched as
    # update() batches them (SVEA / DrQ run their passes as ONE batch; the f16-pair kernels' last bits depend on an image's
    # position in the launch, see tests/test_encoder_gpu.py::test_full_batch_linearity_property) -- which of the two batchings
    # update() used is decided below by the features it computed, bit for bit
    ids = so.conv_layer_ids(11)
    cnn_w = [cs[f"encoder.shared_cnn.layers.{i}.weight"].to(cuda) for i in ids]
    cnn_b = [cs[f"encoder.shared_cnn.layers.{i}.bias"].to(cuda) for i in ids]

    def cnn_gates(x):
        acts = hip_forward(x, cnn_w, cnn_b)
        return [(a > 0).cpu() for a in acts[:-1]], acts[-1]

    def cnn_variants(inputs):
        out = [[cnn_gates(x) for x in inputs]]
        if len(inputs) > 1:
            g, f = cnn_gates(torch.cat(inputs))
            n0 = inputs[0].shape[0]
            out.append([([gl[i * n0:(i + 1) * n0] for gl in g], f[i * n0:(i + 1) * n0]) for i in range(len(inputs))])
        return out

    crit_variants = cnn_variants(critic_inputs)
    weak_cnn = [cnn_gates(x) for x in weak_inputs]
    seen_feat = []
    watched = [agent.critic.encoder.shared_cnn] + ([agent.critic_weak.encoder.shared_cnn] if algo == "augcl" else [])
    hooks = [mod.register_forward_hook(lambda _m, _i, out: seen_feat.append(out.detach())) for mod in watched]
    # The MLP heads' ReLUs run inside GEMM epilogues and HIP kernels (head_ops._TwinQ / _Mlp3): head_ops.gate_hook hands
    # over the post-ReLU activations of every fused forward, in call order.  twin: a1 (B, 2H) = [Q1 | Q2], a2 (2, B, H).
    from augcl_amd import head_ops
    twin_calls, mlp_calls = [], []
    monkeypatch.setattr(head_ops, "gate_hook", lambda kind, a1, a2: (twin_calls if kind == "twin" else mlp_calls).append(
        ((a1 > 0).cpu(), (a2 > 0).cpu())))

    calls = {"shift": 0}

    def shift_given(x, pad=4, **kw):
        i = calls["shift"]
        calls["shift"] += 1
        return (shifted + shifted_next)[i].clone()

    import augcl_amd.algorithms.augcl as m_augcl
    import augcl_amd.algorithms.drq as m_drq
    import augcl_amd.algorithms.svea as m_svea
    for mod in (m_svea, m_drq, m_augcl):
        monkeypatch.setattr(mod.augmentations, "random_shift", shift_given)
    if hasattr(agent, "aug_funcs"):
        keys = list(agent.aug_funcs)
        for k in keys:
            agent.aug_funcs[k]["func"] = (lambda x, **kw: strong_d.clone()) if k == keys[0] else (lambda x, **kw: x)
    queue = [n.to(cuda) for n in noises]
    monkeypatch.setattr(torch, "randn_like", lambda t, *a, **k: queue.pop(0).clone())
    before = {"critic": agent.critic_group.param.clone(), "actor": agent.actor_group.param.clone()}
    if algo == "augcl":
        before["weak"] = agent._weak_group.param.clone()

    class Buf:
        def sample(self, n=None, idxs=None):
            return d_obs.clone(), tt(act).to(cuda), tt(rew).to(cuda), d_nobs.clone(), tt(nd).to(cuda)

    L = RecLogger()
    agent.update(Buf(), L, 0)
    torch.cuda.synchronize()
    for h in hooks:
        h.remove()
    assert not queue

    # the features update() computed are the ones the gates were taken from (bit for bit)
    crit_cnn = None
    for cand in crit_variants:
        want_feats = ([f for _, f in weak_cnn] if algo == "augcl" else []) + [f for _, f in cand]
        got = torch.cat([f.reshape(-1, 32, 21, 21) for f in seen_feat])[:sum(f.shape[0] for f in want_feats)]
        if torch.equal(got, torch.cat(want_feats)):
            crit_cnn = cand
    assert crit_cnn is not None, "encoder features inside update() differ from the C-ABI forward"

    # assemble the gate sets per oracle pass.  twin_q calls in update() order: the target pass(es) (no gradient: not
    # forced), then the critic passes (AugCL: weak first), last the actor step's critic call.
    Hd = hp["hidden_dim"]
    twin = []
    for l1, l2 in twin_calls:
        rows = l1.shape[0]
        twin.append(([l1.view(rows, 2, Hd)[:, t] for t in (0, 1)], [l2[t] for t in (0, 1)]))  # per trunk: layer-1, layer-2 gates

    def q_gates(call, rows=None):
        l1, l2 = twin[call]
        cut = (lambda t: t) if rows is None else (lambda t: t[rows[0]:rows[1]])
        return {"Q1": [cut(l1[0]), cut(l2[0])], "Q2": [cut(l1[1]), cut(l2[1])]}

    n_pass = len(critic_inputs)
    first = M if algo == "drq" else 1               # index of the first gradient-carrying twin_q call
    forced = {"critic": [], "actor": {"mlp": list(mlp_calls[-1])}, "actor_q": q_gates(-1)}
    if algo == "augcl":
        forced["weak"] = [dict(cnn=weak_cnn[0][0], **q_gates(first))]
        first += 1
    for k in range(n_pass):
        if algo == "svea":  # SVEA's two halves go through the critic as ONE 2B batch
            forced["critic"].append(dict(cnn=crit_cnn[k][0], **q_gates(first, (k * B, (k + 1) * B))))
        else:
            forced["critic"].append(dict(cnn=crit_cnn[k][0], **q_gates(first + k)))
    assert len(twin) == first + (1 if algo == "svea" else n_pass) + 1, len(twin)

    # Between the critic and the actor phase the forced referee takes over the product's critic parameters: Adam's
    # first step is lr*sign(g), so the handful of elements whose gradient is ~0 land 2*lr apart in ANY two float32/64
    # evaluations -- Adam itself is checked exactly below (HIP's step == Adam's step of HIP's gradient).
    def take_hip_critic(ag):
        with torch.no_grad():
            for grp, params in ((agent.critic_group, ag.critic),) + (((agent._weak_group, ag.critic_weak),) if algo == "augcl" else ()):
                for n in grp.names:
                    a, b = grp.slices[n]
                    params[n].copy_(grp.param[a:b].view(grp.shapes[n]).to(params[n]))

    ref_free, log_free = oracle_run(torch.float64)
    o32, log32 = oracle_run(torch.float32)
    # AUDIT of every gate set handed to the referee (round 2 only audited the CNN's, in test_encoder_against_f64_referee): the
    # forced float64 run records, per ReLU, how many imposed gates differ from the sign of ITS OWN float64 pre-activation
    # (computed under the same upstream gates) and how far from zero the worst of them sits.  A wrong-but-consistent gate
    # in the fused MLP heads (head_ops._TwinQ / _Mlp3) would be inherited by a referee that just takes the gates; here it
    # would show as a differing gate with |z| / rms far above rounding.
    audits = []
    for gset in forced["critic"] + forced.get("weak", []) + [forced["actor"], forced["actor_q"]]:
        gset["audit"] = []
        audits.append(gset["audit"])
    ref, log64 = oracle_run(torch.float64, forced=forced, after_critic=take_hip_critic)
    audit_rows = [r for a in audits for r in a]
    for gset in forced["critic"] + forced.get("weak", []) + [forced["actor"], forced["actor_q"]]:
        del gset["audit"]
    head_rows = [r for r in audit_rows if r[0] != "cnn"]
    assert head_rows and any(r[0] == "mlp" for r in head_rows) and any(r[0] == "Q1" for r in head_rows)
    n_head = sum(r[3] for r in head_rows)
    flips_head = sum(r[2] for r in head_rows)
    worst_head = max(r[4] for r in head_rows)
    worst_cnn = max([r[4] for r in audit_rows if r[0] == "cnn"], default=0.0)
    gate_audit = {"head_gates": n_head, "head_gates_differ": flips_head, "head_max_|z|/rms_at_differing": worst_head,
                  "cnn_gates_differ": sum(r[2] for r in audit_rows if r[0] == "cnn"), "cnn_max_|z|/rms_at_differing": worst_cnn}
    assert worst_head <= 1e-5 and flips_head <= 5e-6 * n_head + 10, gate_audit
    assert worst_cnn <= 1e-5, gate_audit
    # the float32 oracle under the SAME forced gates: what float32 arithmetic alone costs on this batch (a loss made of
    # cancelling differences, a tanh that saturates to exactly 1 in float32 but not in float64, ...) -- the yardstick
    # wherever that exceeds 1e-5
    y32, ylog32 = oracle_run(torch.float32, forced=forced, after_critic=take_hip_critic)

    rep = {"algo": algo, "B": B, "init": init, "hp": hpo, "referee_device": str(rdev), "logs": {}, "tensors": {}, "groups": {},
           "gate_audit": gate_audit}
    # ---- every logged quantity against the gate-forced referee
    for k, want in log64.items():
        got_v = L.rows[k]
        scale = max(1.0, abs(want)) + (T.ENT_CONST(A) if k.endswith("mean_entropy") else 0.0)
        rep["logs"][k] = {"f64_forced": want, "hip": got_v, "cpu32_forced": ylog32[k], "f64_free": log_free[k], "cpu32_free": log32[k]}
        assert abs(got_v - want) <= max(STRICT * scale, 2.0 * abs(ylog32[k] - want)), (k, rep["logs"][k])

    # ---- gradients per tensor; Adam's step from HIP's own gradient
    def collect(group, rec_key, tag, lr):
        rows = []
        for n in group.names:
            a, b = group.slices[n]
            g64 = ref.rec[rec_key].get(n)
            if g64 is None:
                continue
            g_hip = group.grad[a:b].view(group.shapes[n]).cpu()
            d_hip = (group.param[a:b] - before[tag][a:b]).view(group.shapes[n]).cpu()
            # Adam, free of gate noise: the parameter step must be Adam's first step of HIP's own gradient
            g = g_hip.double()
            want_step = -lr * g / (g.abs() + 1e-8)
            p_before = before[tag][a:b].view(group.shapes[n]).cpu().double()  # d_hip = fl(p_after) - p_before: 1 ulp of p
            assert bool(((d_hip.double() - want_step).abs() <= 4e-6 * lr + 1.2e-7 * p_before.abs()).all()), (tag, n)
            rows.append((tag, n, g_hip, g64, ref_free.rec[rec_key][n], o32.rec[rec_key][n], y32.rec[rec_key][n]))
        return rows

    rows = collect(agent.critic_group, "critic_grads", "critic", hp["critic_lr"])
    rows += collect(agent.actor_group, "actor_grads", "actor", hp["actor_lr"])
    if algo == "augcl":
        rows += collect(agent._weak_group, "weak_grads", "weak", hp["critic_lr"])
    groups = {}
    for tag, n, g_hip, g64, g_free, g32, gy in rows:
        groups.setdefault(f"{tag}/{group_of(n)}", []).append((n, g_hip, g64, g_free, g32, gy))
    worst, bad = (0.0, None), []
    for gname, items in groups.items():
        gp = {"hip_vs_forced": pooled([(i[1], i[2]) for i in items]), "cpu32_vs_forced": pooled([(i[5], i[2]) for i in items]),
              "hip_vs_free": pooled([(i[1], i[3]) for i in items]), "cpu32_vs_free": pooled([(i[4], i[3]) for i in items])}
        rep["groups"][gname] = gp
        for n, g_hip, g64, g_free, g32, gy in items:
            row = {"hip_vs_forced": rel_l2(g_hip, g64), "cpu32_vs_forced": rel_l2(gy, g64), "hip_vs_free": rel_l2(g_hip, g_free),
                   "cpu32_vs_free": rel_l2(g32, g_free)}
            rep["tensors"][f"{gname.split('/')[0]}/{n}"] = row
            worst = max(worst, (row["hip_vs_forced"], f"{gname.split('/')[0]}/{n}"))
            # identical gates everywhere: 1e-5, or as close to the referee as the float32 oracle gets (on this tensor or
            # on its module as a whole)
            if row["hip_vs_forced"] > max(STRICT, 2.0 * max(row["cpu32_vs_forced"], gp["cpu32_vs_forced"])):
                bad.append((gname, n, row, gp))
    rep["worst_gradient_vs_forced_referee"] = worst
    rep["violations"] = [str(b[:2]) for b in bad]
    # ---- target network after the EMA, log_alpha (float64 scalar)
    tg = agent.target_group
    pairs = [(tg.param[tg.slices[n][0]:tg.slices[n][1]].view(tg.shapes[n]).cpu(), ref.critic_target[n]) for n in tg.names]
    rep["target_vs_forced"] = pooled(pairs)
    rep["log_alpha"] = {"hip": float(agent.log_alpha.detach()), "f64_forced": float(ref.log_alpha.detach())}
    rep["seconds"] = time.time() - t0
    dump(f"update_{algo}_{data_aug}_B{B}_k{K}m{M}_{init}", rep)
    assert not bad, bad[:3]
    assert rep["target_vs_forced"] <= STRICT, rep["target_vs_forced"]
    assert abs(rep["log_alpha"]["hip"] - rep["log_alpha"]["f64_forced"]) <= 1e-7 * abs(rep["log_alpha"]["f64_forced"]), rep["log_alpha"]
    # sanity bound on the un-forced comparison (gate noise included): both float32 evaluations stay within 1e-2
    assert max(g["hip_vs_free"] for g in rep["groups"].values()) <= 1e-2, rep["groups"]
