"""First layer's backward-weights (augcl_enc_conv_first_backward_weights, float32 dZ): 2-row against 3-row strips, hipGraph replay.
python tools/wgrad_first_bench.py [B ...]"""
import ctypes, sys
import torch
sys.path.insert(0, '.')
from augcl_amd import _lib, encoder_hip  # noqa: E402

dev = torch.device('cuda:0')
raw = ctypes.CDLL(_lib.LIB_PATH)
lib = _lib.load()
for B in [int(a) for a in sys.argv[1:]] or [512, 256]:
    x = torch.randint(0, 256, (B, 9, 84, 84), device=dev).float()
    dz = torch.randn(B, 32, 41, 41, device=dev)
    dw = torch.empty(32, 9, 3, 3, device=dev); db = torch.empty(32, device=dev)
    ws = encoder_hip._workspace(dev, 9)
    res = {}
    for rep in range(2):
        for on in (0, 1):
            raw.augcl_internal_wgrad_first_rows3(on)
            def run():
                _lib.check(lib.augcl_enc_conv_first_backward_weights(x.data_ptr(), dz.data_ptr(), dw.data_ptr(), db.data_ptr(), ws.data_ptr(),
                                                                     ws.numel() * 4, B, 9, 84, 84, 2, 255.0, _lib.current_stream(dev)), "wgrad_first")
            s = torch.cuda.Stream()
            with torch.cuda.stream(s):
                for _ in range(3): run()
                torch.cuda.synchronize()
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, stream=s):
                    for _ in range(20): run()
                g.replay(); torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                best = 1e9
                for _ in range(5):
                    e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
                    best = min(best, e0.elapsed_time(e1) / 20 * 1e3)
            res[on] = dw.clone()
            print(f"B={B} rows3={on}: {best:.1f} us per call (kernel + slab reduction)", flush=True)
    ref = torch.nn.grad.conv2d_weight((x / 255.0).double(), (32, 9, 3, 3), dz.double(), stride=2)
    for on in (0, 1):
        print(f"   rows3={on}: max |dW - f64| / max |dW| = {float((res[on].double() - ref).abs().max() / ref.abs().max()):.2e}")
raw.augcl_internal_wgrad_first_rows3(1)
