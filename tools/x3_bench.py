"""Per-layer timing of the 32->32 layer kernels: native f32 MFMA vs split-bf16 (x3).  python tools/x3_bench.py [B]"""
import sys, torch
sys.path.insert(0, '.')
from augcl_amd import _lib
L = _lib.load()
dev = torch.device('cuda:0')
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
st = torch.cuda.current_stream().cuda_stream
p = lambda t: t.data_ptr()
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): fn()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / n * 1e3)
    return best
w = torch.randn(32, 32, 3, 3, device=dev) * 0.05
bias = torch.zeros(32, device=dev)
ws = torch.empty(L.augcl_enc_wgrad_workspace_bytes(32) // 4, device=dev)
dw, db = torch.empty(32, 32, 3, 3, device=dev), torch.empty(32, device=dev)
tot = {}
print("H    fwd f32 -> x3      bwd f32 -> x3      wgrad f32 -> x3   (us)   x3: fwd TB/s  MFMA-eq TF")
for H in range(41, 21, -2):
    x = torch.rand(B, 32, H, H, device=dev)
    y = torch.empty(B, 32, H - 2, H - 2, device=dev)
    dz = torch.randn(B, 32, H - 2, H - 2, device=dev)
    dx = torch.empty(B, 32, H, H, device=dev)
    row = []
    for kind in ("forward", "backward_data", "backward_weights"):
        for pre in ("augcl_enc_conv32_", "augcl_enc_x3_conv32_"):
            if kind == "forward": fn = lambda: _lib.call(pre + kind, p(x), p(w), p(bias), p(y), B, H, H, 1, st)
            elif kind == "backward_data": fn = lambda: _lib.call(pre + kind, p(dz), p(w), p(x), p(dx), B, H - 2, H - 2, st)
            else: fn = lambda: _lib.call(pre + kind, p(x), p(dz), p(dw), p(db), p(ws), ws.numel() * 4, B, H, H, st)
            t = timeit(fn)
            row.append(t)
            tot[(kind, pre)] = tot.get((kind, pre), 0.0) + t
    byt = B * 32 * 4 * (H * H + (H - 2) * (H - 2))
    fl = 2 * 288 * 32 * B * (H - 2) * (H - 2)
    print("%-4d %6.1f -> %6.1f   %6.1f -> %6.1f   %6.1f -> %6.1f        %5.2f   %6.1f" % (H, *row, byt / row[1] / 1e6, fl / row[1] / 1e6))
print("sum  " + "   ".join("%6.1f -> %6.1f" % (tot[(k, "augcl_enc_conv32_")], tot[(k, "augcl_enc_x3_conv32_")]) for k in ("forward", "backward_data", "backward_weights")))
