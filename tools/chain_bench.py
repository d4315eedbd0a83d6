"""No-grad conv stack: LDS-resident layer chains (augcl_enc_stack_infer) vs the per-layer f16-pair kernels.
python tools/chain_bench.py [B ...]   -> us per pass, single stack and two weight sets (the dual forward of calculate_target_Q)"""
import sys
import numpy as np
import torch
sys.path.insert(0, '.')
from augcl_amd import encoder_hip, encoder_ops  # noqa: E402

dev = torch.device('cuda:0')
prec = "bf16" if "bf16" in sys.argv[1:] else "f32"
Bs = [int(a) for a in sys.argv[1:] if a.isdigit()] or [256, 512]


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


def stack(seed):
    rs = np.random.RandomState(seed)
    ws = [torch.from_numpy((rs.standard_normal((32, 9 if i == 0 else 32, 3, 3)) * np.sqrt(2.0 / ((9 if i == 0 else 32) * 9))).astype(np.float32)).to(dev) for i in range(11)]
    bs = [torch.from_numpy((rs.standard_normal(32) * 0.05).astype(np.float32)).to(dev) for i in range(11)]
    return ws, bs


(wa, ba), (wb, bb) = stack(1), stack(2)
print("precision", prec)
print("B     single: layerwise -> chains (us)     dual: layerwise -> chains (us)")
for B in Bs:
    x = torch.randint(0, 256, (B, 9, 84, 84), device=dev).float()
    res = []
    with torch.no_grad():
        for dual in (False, True):
            for chains in (False, True):
                encoder_hip._CHAINS = chains
                if dual:
                    fn = lambda: encoder_ops.conv_stack_dual(x, wa, ba, wb, bb, precision=prec)  # noqa: E731
                else:
                    fn = lambda: encoder_ops.conv_stack(x, wa, ba, precision=prec)  # noqa: E731
                res.append(timeit(fn))
    encoder_hip._CHAINS = True
    print("%-5d %9.1f -> %7.1f                    %9.1f -> %7.1f" % (B, *res))
