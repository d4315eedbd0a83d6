"""Gradient-carrying conv stack, precision "f32": per-layer f16-pair kernels vs layer chains that store every layer's output
(augcl_enc_stack_forward_chained, ABI v6).  Forward alone and forward + backward, eager launches (the queue stays full).
python tools/keep_bench.py [B ...]          then an update-level A/B:  python tools/keep_bench.py update [sac|rad|drq|svea] [B]"""
import sys
import numpy as np
import torch
sys.path.insert(0, '.')
from augcl_amd import encoder_hip, encoder_ops  # noqa: E402

dev = torch.device('cuda:0')

if len(sys.argv) > 1 and sys.argv[1] == "update":
    from augcl_amd import bench_update
    algo = sys.argv[2] if len(sys.argv) > 2 else "sac"
    B = int(sys.argv[3]) if len(sys.argv) > 3 else 256
    aug = "conv" if algo in ("svea", "augcl") else "shift"
    for rep in range(2):
        for on in (False, True):
            encoder_hip._CHAINS_GRAD = on
            r = bench_update.run(dev, None, 1, 0, steps=100, B=B, algo=algo, data_aug=aug, precision="f32", graph=True, min_s=1.0)
            print(f"f32 {algo}+{aug} B={B} kept-forward chains={on}: {r['ms_per_update']:.4f} ms/update  ({r['grad_steps_per_s']:.1f} grad-steps/s)", flush=True)
    encoder_hip._CHAINS_GRAD = True
    sys.exit(0)

rs = np.random.RandomState(1)
ws = [torch.from_numpy((rs.standard_normal((32, 9 if i == 0 else 32, 3, 3)) * np.sqrt(2.0 / ((9 if i == 0 else 32) * 9))).astype(np.float32)).to(dev).requires_grad_(True) for i in range(11)]
bs = [torch.from_numpy((rs.standard_normal(32) * 0.05).astype(np.float32)).to(dev).requires_grad_(True) for i in range(11)]


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / n * 1e3)
    return best


for B in [int(a) for a in sys.argv[1:]] or [512, 256, 128]:
    x = torch.randint(0, 256, (B, 9, 84, 84), device=dev).float()
    g = torch.randn(B, 32, 21, 21, device=dev)

    def fwd():
        return encoder_ops.conv_stack(x, ws, bs, precision="f32")

    def fwd_bwd():
        fwd().backward(g)

    encoder_hip._CHAINS_GRAD_MAX_B = 1 << 30  # (the product takes the chains up to one image per CU: measured here beyond it)
    for rep in range(2):
        for on in (False, True):
            encoder_hip._CHAINS_GRAD = on
            print(f"f32 B={B} kept-forward chains={on}: forward {timeit(fwd):.1f} us, forward + backward {timeit(fwd_bwd):.1f} us", flush=True)
encoder_hip._CHAINS_GRAD = True
