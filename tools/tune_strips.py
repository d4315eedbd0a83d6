"""Sweep the number of strips per image (AUGCL_EXTRA_STRIPS) for the 32-channel conv kernels over the encoder's
layer sizes: python tools/tune_strips.py [B]"""
import ctypes, sys, os, torch
sys.path.insert(0, '.')
from augcl_amd import _lib
# tuning build (reads the AUGCL_* overrides): make -B -C augcl_amd/csrc EXTRA_FLAGS=-DAUGCL_TUNING OUT=../../tools/libaugcl_tuning.so; then make -B -C augcl_amd/csrc
_lib.LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'libaugcl_tuning.so')
B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
dev = torch.device('cuda:0')
L = _lib.load()
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
st = torch.cuda.current_stream().cuda_stream
w = torch.randn(32, 32, 3, 3, device=dev) * 0.05
bias = torch.zeros(32, device=dev)
p = lambda t: ctypes.c_void_p(t.data_ptr())
kinds = ["f32 fwd", "f32 bwd"]
print("B=%d   columns: extra strips 0..4 (us), best" % B)
for H in range(41, 21, -2):
    x = torch.rand(B, 32, H, H, device=dev) - 0.3
    y = torch.empty(B, 32, H - 2, H - 2, device=dev)
    dz = torch.rand(B, 32, H - 2, H - 2, device=dev)
    dx = torch.empty(B, 32, H, H, device=dev)
    xb = (torch.rand(B, H, H, 32, device=dev) - 0.3).bfloat16()
    yb = torch.empty(B, H - 2, H - 2, 32, device=dev, dtype=torch.bfloat16)
    dzb = torch.rand(B, H - 2, H - 2, 32, device=dev).bfloat16()
    dxb = torch.empty(B, H, H, 32, device=dev, dtype=torch.bfloat16)
    fns = {'f32 fwd': lambda: L.augcl_enc_conv32_forward(p(x), p(w), p(bias), p(y), B, H, H, 1, st),
           'f32 bwd': lambda: L.augcl_enc_conv32_backward_data(p(dz), p(w), p(x), p(dx), B, H - 2, H - 2, st),
           'bf16 fwd': lambda: L.augcl_enc_bf16_conv32_forward(p(xb), p(w), p(bias), p(yb), B, H, H, 1, 0, st),
           'bf16 bwd': lambda: L.augcl_enc_bf16_conv32_backward_data(p(dzb), p(w), p(xb), p(dxb), B, H - 2, H - 2, st)}
    for k in kinds:
        ts = []
        for extra in range(5):
            os.environ["AUGCL_EXTRA_STRIPS"] = str(extra)
            ts.append(timeit(fns[k]))
        best = min(range(5), key=lambda i: ts[i])
        print("H=%d %-8s " % (H, k) + " ".join("%6.1f" % t for t in ts) + "  best +%d (%.0f%%)" % (best, 100 * (1 - ts[best] / ts[0])))
os.environ.pop("AUGCL_EXTRA_STRIPS", None)
