"""conv stack forward + backward (one autograd node), bf16 mode: per-layer kernels vs chains (forward only / forward + backward-data).
python tools/stack_fb_bench.py [B]"""
import sys
import numpy as np
import torch
sys.path.insert(0, '.')
from augcl_amd import encoder_hip, encoder_ops  # noqa: E402

dev = torch.device('cuda:0')
B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
prec = sys.argv[2] if len(sys.argv) > 2 else "bf16"
rs = np.random.RandomState(1)
ws = [torch.from_numpy((rs.standard_normal((32, 9 if i == 0 else 32, 3, 3)) * np.sqrt(2.0 / ((9 if i == 0 else 32) * 9))).astype(np.float32)).to(dev).requires_grad_(True) for i in range(11)]
bs = [torch.from_numpy((rs.standard_normal(32) * 0.05).astype(np.float32)).to(dev).requires_grad_(True) for i in range(11)]
x = torch.randint(0, 256, (B, 9, 84, 84), device=dev).float()
g = torch.randn(B, 32, 21, 21, device=dev)


def run():
    out = encoder_ops.conv_stack(x, ws, bs, precision=prec)
    out.backward(g)


def timeit(n=20):
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            run()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / n * 1e3)
    return best


for rep in range(2):
    for fwd, bwd in ((False, False), (True, False), (True, True)):
        encoder_hip._CHAINS, encoder_hip._CHAINS_BWD = fwd, bwd
        print(f"{prec} B={B} chains fwd={fwd} bwd={bwd}: {timeit():.1f} us per forward + backward", flush=True)
encoder_hip._CHAINS = encoder_hip._CHAINS_BWD = True
