"""A/B two library builds on the headline random_conv launch: python tools/ab_conv.py LIB_A LIB_B"""
import ctypes, sys, os, torch
sys.path.insert(0, '.')
from augcl_amd import _lib
dev = torch.device('cuda:0'); B = 256
xs = [torch.randint(0, 256, (B, 9, 84, 84), device=dev).float() for _ in range(8)]
outs = [torch.empty_like(xs[0]) for _ in range(8)]
w = torch.randn(B, 81, device=dev)
st = torch.cuda.current_stream().cuda_stream
p = lambda t: ctypes.c_void_p(t.data_ptr())
for rep in range(2):
    for path in sys.argv[1:3]:
        L = ctypes.CDLL(os.path.abspath(path))
        L.augcl_random_conv_f32.argtypes = _lib._SIGS["augcl_random_conv_f32"]; L.augcl_random_conv_f32.restype = ctypes.c_int
        run = lambda i: L.augcl_random_conv_f32(p(xs[i % 8]), p(outs[i % 8]), B, 3, 84, 84, p(w), 0, 0, 0, st)
        for i in range(200): run(i)
        torch.cuda.synchronize(); best = 1e9
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for i in range(1000): run(i)
            e1.record(); torch.cuda.synchronize(); best = min(best, e0.elapsed_time(e1) / 1000 * 1e3)
        print(os.path.basename(path), "%.2f us  %.2f M obs/s  %.0f GB/s" % (best, B / best, B * 508356 / best / 1e3))
