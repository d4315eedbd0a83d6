import os, sys
import numpy as np, torch, torch.nn.functional as F
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, ROOT + "/tests"); sys.path.insert(0, ROOT + "/tests/golden")
from augcl_amd import encoder_ops
import test_encoder_gpu as E
cuda = torch.device("cuda:0")
for B in (5, 16, 5, 16, 5, 16, 32, 32):
    rs = np.random.RandomState(B)
    x = torch.from_numpy(rs.randint(0, 256, (B, 9, 84, 84)).astype(np.float32))
    n_layers = 11
    ws = [torch.from_numpy((rs.standard_normal((32, 9 if i == 0 else 32, 3, 3)) * np.sqrt(2.0 / (9 * (9 if i == 0 else 32)))).astype(np.float32)) for i in range(n_layers)]
    bs = [torch.from_numpy((0.05 * rs.standard_normal(32)).astype(np.float32)) for _ in range(n_layers)]
    wr = [w.clone().requires_grad_(True) for w in ws]; br = [b.clone().requires_grad_(True) for b in bs]
    ref = E.ref_stack(x, wr, br, 2, 255.0)
    gout = torch.from_numpy(rs.standard_normal(tuple(ref.shape)).astype(np.float32))
    (ref * gout).sum().backward()
    wg = [w.clone().to(cuda).requires_grad_(True) for w in ws]; bg = [b.clone().to(cuda).requires_grad_(True) for b in bs]
    out = encoder_ops.conv_stack(x.to(cuda), wg, bg)
    (out * gout.to(cuda)).sum().backward()
    errs = [E.rel_err(wg[i].grad.cpu(), wr[i].grad) for i in range(n_layers)]
    print(B, "fwd %.1e" % E.rel_err(out.detach().cpu(), ref.detach()), " ".join("%.1e" % e for e in errs))
