"""One process running the gradient-carrying conv stack (precision "f32", forward + backward) N times at batch B, for rocprofv3:
python tools/prof_keep.py N B   (B <= 256: the 32 -> 32 layers of the forward run as the storing chains, k_conv32_chain<true>)"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from augcl_amd import encoder_ops  # noqa: E402

dev = torch.device("cuda:0")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 20
B = int(sys.argv[2]) if len(sys.argv) > 2 else 256
rs = np.random.RandomState(1)
ws = [torch.from_numpy((rs.standard_normal((32, 9 if i == 0 else 32, 3, 3)) * np.sqrt(2.0 / ((9 if i == 0 else 32) * 9))).astype(np.float32)).to(dev).requires_grad_(True) for i in range(11)]
bs = [torch.from_numpy((rs.standard_normal(32) * 0.05).astype(np.float32)).to(dev).requires_grad_(True) for i in range(11)]
x = torch.randint(0, 256, (B, 9, 84, 84), device=dev).float()
g = torch.randn(B, 32, 21, 21, device=dev)
for _ in range(N):
    encoder_ops.conv_stack(x, ws, bs, precision="f32").backward(g)
torch.cuda.synchronize()
print("done", N, B)
