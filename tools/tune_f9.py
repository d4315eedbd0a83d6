import ctypes, sys, os, torch
sys.path.insert(0, '.')
from augcl_amd import _lib
# tuning build (reads the AUGCL_* overrides): make -B -C augcl_amd/csrc EXTRA_FLAGS=-DAUGCL_TUNING OUT=../../tools/libaugcl_tuning.so; then make -B -C augcl_amd/csrc
_lib.LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'libaugcl_tuning.so')
L = _lib.load(); dev = torch.device('cuda:0')
st = torch.cuda.current_stream().cuda_stream
p = lambda t: ctypes.c_void_p(t.data_ptr())
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): fn()
        e1.record(); torch.cuda.synchronize(); best = min(best, e0.elapsed_time(e1) / n * 1e3)
    return best
w = torch.randn(32, 9, 3, 3, device=dev) * 0.1; b = torch.zeros(32, device=dev)
for B in (512, 256):
    x = torch.randint(0, 256, (B, 9, 84, 84), device=dev).float()
    y = torch.empty(B, 32, 41, 41, device=dev); yb = torch.empty(B, 41, 41, 32, device=dev, dtype=torch.bfloat16)
    for rt in (0, 3, 4, 5, 6, 7):
        if rt: os.environ["AUGCL_F9_RT"] = str(rt)
        else: os.environ.pop("AUGCL_F9_RT", None)
        t1 = timeit(lambda: L.augcl_enc_conv_first_forward(p(x), p(w), p(b), p(y), B, 9, 84, 84, 2, 255.0, 1, st))
        t2 = timeit(lambda: L.augcl_enc_bf16_conv_first_forward(p(x), p(w), p(b), p(yb), B, 9, 84, 84, 2, 255.0, 1, st))
        print("B=%d RT=%s  f32 out %.1f us   bf16 out %.1f us" % (B, rt or "auto", t1, t2))
