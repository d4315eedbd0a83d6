"""Conv-stack passes as update() issues them, timed inside ONE source tree (argv[1]): the no-grad dual forward (B = 256 images,
two weight sets), the no-grad single forward (B = 256), forward + backward (B = 512).  tools/ab_trees.py-style A/B:
  for t in _prev .; do python tools/ab_stack.py $t; done"""
import sys, time, numpy as np, torch
tree = sys.argv[1] if len(sys.argv) > 1 else "."
sys.path.insert(0, tree.split(":")[0])
import os
from augcl_amd import _lib, encoder_ops, encoder_hip
if os.environ.get("AUGCL_LIB"):  # A/B of two library builds of the SAME tree (same ABI): AUGCL_LIB=tools/libaugcl_variant.so
    _lib.LIB_PATH = os.path.abspath(os.environ["AUGCL_LIB"])
if tree.endswith(":nobits"):
    encoder_hip._GATE_BITS = False
dev = torch.device("cuda:0")
rs = np.random.RandomState(0)
ws = [torch.from_numpy((rs.standard_normal((32, 9 if i == 0 else 32, 3, 3)) * (0.2 if i == 0 else 0.08)).astype(np.float32)).to(dev) for i in range(11)]
bs = [torch.zeros(32, device=dev) for _ in range(11)]
w2 = [w * 1.05 for w in ws]
x256 = torch.from_numpy(rs.randint(0, 256, (256, 9, 84, 84)).astype(np.float32)).to(dev)
x512 = torch.cat([x256, x256.flip(0)])
g = torch.randn(512, 32, 21, 21, device=dev)
wg = [w.clone().requires_grad_(True) for w in ws]
bg = [b.clone().requires_grad_(True) for b in bs]


def t(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / n * 1e6)
    return best


def fb():
    out = encoder_ops.conv_stack(x512, wg, bg)
    out.backward(g)


def fwd_only():
    return encoder_ops.conv_stack(x512, wg, bg)


with torch.no_grad():
    d = t(lambda: encoder_ops.conv_stack_dual(x256, ws, bs, w2, bs))
    s = t(lambda: encoder_ops.conv_stack(x256, ws, bs))
    s5 = t(lambda: encoder_ops.conv_stack(x512, ws, bs))
f = t(fwd_only)
a = t(fb)
print(f"{tree:8s} dual fwd {d:7.1f} us | no-grad fwd B=256 {s:7.1f} | no-grad fwd B=512 {s5:7.1f} | grad fwd B=512 {f:7.1f} | fwd+bwd B=512 {a:7.1f}")
