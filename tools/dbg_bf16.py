import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, ROOT + "/tests"); sys.path.insert(0, ROOT + "/tests/golden")
from augcl_amd import encoder_ops
import test_encoder_gpu as E
cuda = torch.device("cuda:0")
for (B, C, H, n_layers) in [(3, 9, 84, 11), (2, 9, 84, 3), (5, 3, 44, 4), (8, 9, 84, 11)]:
    rs = np.random.RandomState(B + H)
    x = torch.from_numpy(rs.randint(0, 256, (B, C, H, H)).astype(np.float32))
    ws = [torch.from_numpy((rs.standard_normal((32, C if i == 0 else 32, 3, 3)) * np.sqrt(2.0 / (9 * (C if i == 0 else 32)))).astype(np.float32)) for i in range(n_layers)]
    bs = [torch.from_numpy((0.05 * rs.standard_normal(32)).astype(np.float32)) for _ in range(n_layers)]
    wr = [w.clone().requires_grad_(True) for w in ws]; br = [b.clone().requires_grad_(True) for b in bs]
    ref = E.ref_stack(x, wr, br, 2, 255.0)
    gout = torch.from_numpy(rs.standard_normal(tuple(ref.shape)).astype(np.float32))
    (ref * gout).sum().backward()
    wg = [w.clone().to(cuda).requires_grad_(True) for w in ws]; bg = [b.clone().to(cuda).requires_grad_(True) for b in bs]
    encoder_ops.set_precision("bf16")
    out = encoder_ops.conv_stack(x.to(cuda), wg, bg)
    (out * gout.to(cuda)).sum().backward()
    encoder_ops.set_precision("f32")
    l2 = lambda a, b: float((a.double() - b.double()).norm() / b.double().norm())
    print(B, C, H, n_layers, "fwd max %.1e l2 %.1e" % (E.rel_err(out.detach().cpu(), ref.detach()), l2(out.detach().cpu(), ref.detach())))
    print("  dW max:", " ".join("%.1e" % E.rel_err(wg[i].grad.cpu(), wr[i].grad) for i in range(n_layers)))
    print("  dW l2 :", " ".join("%.1e" % l2(wg[i].grad.cpu(), wr[i].grad) for i in range(n_layers)))
    print("  db max:", " ".join("%.1e" % E.rel_err(bg[i].grad.cpu(), br[i].grad) for i in range(n_layers)))
print("gate-safe weights (no ReLU gate within bf16 noise of zero):")
from oracle import sac_oracle as so
hp = dict(so.DEFAULTS)
st = so.synth_state(5, so.critic_shapes((9, 84, 84), 6, hp))
ids = so.conv_layer_ids(11)
for B in (3, 8):
    rs = np.random.RandomState(B)
    x = torch.from_numpy(rs.randint(0, 256, (B, 9, 84, 84)).astype(np.float32))
    ws = [st[f"encoder.shared_cnn.layers.{i}.weight"] for i in ids]; bs = [st[f"encoder.shared_cnn.layers.{i}.bias"] for i in ids]
    wr = [w.clone().requires_grad_(True) for w in ws]; br = [b.clone().requires_grad_(True) for b in bs]
    ref = E.ref_stack(x, wr, br, 2, 255.0)
    gout = torch.from_numpy(rs.standard_normal(tuple(ref.shape)).astype(np.float32))
    (ref * gout).sum().backward()
    wg = [w.clone().to(cuda).requires_grad_(True) for w in ws]; bg = [b.clone().to(cuda).requires_grad_(True) for b in bs]
    encoder_ops.set_precision("bf16")
    out = encoder_ops.conv_stack(x.to(cuda), wg, bg)
    (out * gout.to(cuda)).sum().backward()
    encoder_ops.set_precision("f32")
    l2 = lambda a, b: float((a.double() - b.double()).norm() / b.double().norm())
    print(B, "fwd l2 %.1e" % l2(out.detach().cpu(), ref.detach()), " dW l2:", " ".join("%.1e" % l2(wg[i].grad.cpu(), wr[i].grad) for i in range(11)))
