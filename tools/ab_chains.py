"""A/B on ONE box: SVEA update() with the no-grad passes as LDS-resident chains vs per-layer kernels.
python tools/ab_chains.py [weights=init|dense] [B] [f32|bf16]"""
import sys
import torch
sys.path.insert(0, '.')
from augcl_amd import bench_update, encoder_hip  # noqa: E402

w = sys.argv[1] if len(sys.argv) > 1 else "init"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 256
prec = sys.argv[3] if len(sys.argv) > 3 else "f32"
dev = torch.device('cuda:0')
for rep in range(2):
    for chains in (False, True):
        encoder_hip._CHAINS = chains
        r = bench_update.run(dev, None, 1, 0, steps=100, B=B, precision=prec, graph=True, min_s=1.0, weights=w)
        print(f"{prec} chains={chains} weights={w} B={B}: {r['ms_per_update']:.4f} ms/update  ({r['grad_steps_per_s']:.1f} grad-steps/s)", flush=True)
