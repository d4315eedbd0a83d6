"""Per-layer A/B of two library builds on one box: python tools/ab_kernels.py LIB_A LIB_B"""
import ctypes, sys, os, torch
sys.path.insert(0, '.')
from augcl_amd import _lib
libs = [os.path.abspath(p) for p in sys.argv[1:3]]
dev = torch.device('cuda:0')
B = int(sys.argv[3]) if len(sys.argv) > 3 else 512
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
res = {}
for li, path in enumerate(libs):
    L = ctypes.CDLL(path)
    for name, sig in _lib._SIGS.items():
        if hasattr(L, name):
            getattr(L, name).argtypes = sig
            getattr(L, name).restype = ctypes.c_int
    L.augcl_enc_wgrad_workspace_bytes.restype = ctypes.c_size_t
    st = torch.cuda.current_stream().cuda_stream
    w = torch.randn(32, 32, 3, 3, device=dev) * 0.05
    bias = torch.zeros(32, device=dev)
    for H in range(41, 21, -2):
        x = torch.rand(B, 32, H, H, device=dev)
        y = torch.empty(B, 32, H - 2, H - 2, device=dev)
        dz = torch.rand(B, 32, H - 2, H - 2, device=dev)
        dx = torch.empty(B, 32, H, H, device=dev)
        xb = torch.rand(B, H, H, 32, device=dev).bfloat16()
        yb = torch.empty(B, H - 2, H - 2, 32, device=dev, dtype=torch.bfloat16)
        dzb = torch.rand(B, H - 2, H - 2, 32, device=dev).bfloat16()
        dxb = torch.empty(B, H, H, 32, device=dev, dtype=torch.bfloat16)
        p = lambda t: ctypes.c_void_p(t.data_ptr())
        res[(li, H, 'f32 fwd')] = timeit(lambda: L.augcl_enc_conv32_forward(p(x), p(w), p(bias), p(y), B, H, H, 1, st))
        res[(li, H, 'f32 bwd')] = timeit(lambda: L.augcl_enc_conv32_backward_data(p(dz), p(w), p(x), p(dx), B, H - 2, H - 2, st))
        res[(li, H, 'bf16 fwd')] = timeit(lambda: L.augcl_enc_bf16_conv32_forward(p(xb), p(w), p(bias), p(yb), B, H, H, 1, 0, st))
        res[(li, H, 'bf16 bwd')] = timeit(lambda: L.augcl_enc_bf16_conv32_backward_data(p(dzb), p(w), p(xb), p(dxb), B, H - 2, H - 2, st))
kinds = ['f32 fwd', 'f32 bwd', 'bf16 fwd', 'bf16 bwd']
print("H    " + "   ".join("%-17s" % k for k in kinds))
tot = {(li, k): 0.0 for li in (0, 1) for k in kinds}
for H in range(41, 21, -2):
    row = []
    for k in kinds:
        a, b = res[(0, H, k)], res[(1, H, k)]
        tot[(0, k)] += a; tot[(1, k)] += b
        row.append("%6.1f -> %6.1f" % (a, b))
    print("%-4d " % H + "   ".join("%-17s" % r for r in row))
print("sum  " + "   ".join("%6.1f -> %6.1f  " % (tot[(0, k)], tot[(1, k)]) for k in kinds))
