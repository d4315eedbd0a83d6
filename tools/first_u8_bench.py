"""First layer / no-grad stack: uint8 observations vs float32 ones (us per call, hipGraph replay).  python tools/first_u8_bench.py"""
import ctypes, sys
import torch
sys.path.insert(0, '.')
from augcl_amd import _lib, encoder_hip  # noqa: E402
import os  # noqa: E402
if len(sys.argv) > 1 and sys.argv[1] == "tuning":  # tools/build_tuning.sh; AUGCL_U8_ABL is read by that build only
    _lib.LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'libaugcl_tuning.so')

dev = torch.device('cuda:0')
P = ctypes.c_void_p


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


w = torch.randn(32, 9, 3, 3, device=dev) * 0.15
b = torch.randn(32, device=dev) * 0.05
st = lambda: _lib.current_stream(dev)
for B in (256, 512):
    x8 = torch.randint(0, 256, (B, 9, 84, 84), device=dev, dtype=torch.uint8)
    xf = x8.float()
    out = torch.empty((B, 32, 41, 41), device=dev)
    t8 = graph_time(lambda: _lib.call("augcl_enc_conv_first_forward_u8", P(x8.data_ptr()), P(w.data_ptr()), P(b.data_ptr()), P(out.data_ptr()), B, 9, 84, 84, 2, 255.0, 1, st()))
    tf = graph_time(lambda: _lib.call("augcl_enc_conv_first_forward", P(xf.data_ptr()), P(w.data_ptr()), P(b.data_ptr()), P(out.data_ptr()), B, 9, 84, 84, 2, 255.0, 1, st()))
    raw = ctypes.CDLL(_lib.LIB_PATH)
    raw.augcl_internal_first_layer_x3(0)  # the float32 matrix pipe (rounds 1-4) for comparison
    t32 = graph_time(lambda: _lib.call("augcl_enc_conv_first_forward", P(xf.data_ptr()), P(w.data_ptr()), P(b.data_ptr()), P(out.data_ptr()), B, 9, 84, 84, 2, 255.0, 1, st()))
    raw.augcl_internal_first_layer_x3(1)
    print(f"first layer B={B}: f32 in, f32 MFMA pipe {t32:6.1f} us ({B * (63504 * 4 + 32 * 1681 * 4) / t32 / 1e6:5.2f} TB/s)", flush=True)
    by8, byf = B * (63504 + 32 * 1681 * 4), B * (63504 * 4 + 32 * 1681 * 4)
    print(f"first layer B={B}: f32 in {tf:6.1f} us ({byf / tf / 1e6:5.2f} TB/s)   uint8 in {t8:6.1f} us ({by8 / t8 / 1e6:5.2f} TB/s)", flush=True)
ws = [w] + [torch.randn(32, 32, 3, 3, device=dev) * (2.0 / 288) ** 0.5 for _ in range(10)]
bs = [torch.randn(32, device=dev) * 0.05 for _ in range(11)]
ws2 = [t + 0.01 for t in ws]
B = 256
x8 = torch.randint(0, 256, (B, 9, 84, 84), device=dev, dtype=torch.uint8)
xf = x8.float()
for sets in (1, 2):
    wb, bb = (ws2, bs) if sets == 2 else (None, None)
    t8 = graph_time(lambda: encoder_hip.conv_stack_infer(x8, ws, bs, wb, bb, 2, 255.0, "f32"))
    tf = graph_time(lambda: encoder_hip.conv_stack_infer(xf, ws, bs, wb, bb, 2, 255.0, "f32"))
    print(f"no-grad stack B={B} x {sets} set(s): f32 in {tf:6.1f} us   uint8 in {t8:6.1f} us", flush=True)
