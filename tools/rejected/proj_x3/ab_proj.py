"""Projection forward (augcl_proj_ln_tanh_forward): split-K partial sums on the bf16 pipe (proj_x3.hip) vs the float32 pipe, us per
call (hipGraph replay) and the difference of the results against float64.  python tools/ab_proj.py"""
import ctypes, sys
import torch
sys.path.insert(0, '.')
from augcl_amd import _lib, head_ops  # noqa: E402

dev = torch.device('cuda:0')
raw = ctypes.CDLL(_lib.LIB_PATH)
K, N = 14112, 100


def graph_time(fn, n=10, reps=5):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n):
            fn()
    best = 1e9
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / n * 1e3)
    return best


lin = torch.nn.Linear(K, N).to(dev)
ln = torch.nn.LayerNorm(N).to(dev)
for M in (64, 256, 512, 1024):
    x = torch.relu(torch.randn(M, K, device=dev))
    with torch.no_grad():
        want = torch.tanh(torch.nn.functional.layer_norm(x.double() @ lin.weight.double().t() + lin.bias.double(), (N,), ln.weight.double(), ln.bias.double(), ln.eps))
        row = []
        for on in (0, 1):
            raw.augcl_internal_proj_x3(on)
            z = head_ops.proj_ln_tanh(x, lin, ln)
            err = float((z.double() - want).abs().max())
            t = graph_time(lambda: head_ops.proj_ln_tanh(x, lin, ln))
            row.append((t, err))
    print(f"M={M}: f32 pipe {row[0][0]:6.1f} us (err {row[0][1]:.1e})   bf16 pieces {row[1][0]:6.1f} us (err {row[1][1]:.1e})   [both launches: partial + reduce/LN/tanh]", flush=True)
raw.augcl_internal_proj_x3(1)
# ---- backward, input side: dx = dy W
P = ctypes.c_void_p
w = lin.weight.detach()
for M in (64, 256, 512, 1024):
    dy = torch.randn(M, N, device=dev) * 1e-3
    dx = torch.empty(M, K, device=dev)
    want = dy.double() @ w.double()
    row = []
    for on in (0, 1):
        raw.augcl_internal_proj_x3(on)
        fn = lambda: _lib.call("augcl_proj_dx_f32", P(dy.data_ptr()), P(w.data_ptr()), P(dx.data_ptr()), M, N, K, _lib.current_stream(dev))
        fn()
        err = float((dx.double() - want).abs().max() / want.abs().max())
        row.append((graph_time(fn), err))
    print(f"dx M={M}: f32 pipe {row[0][0]:6.1f} us (rel err {row[0][1]:.1e})   bf16 pieces {row[1][0]:6.1f} us (rel err {row[1][1]:.1e})", flush=True)
raw.augcl_internal_proj_x3(1)
