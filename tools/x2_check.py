"""The whole conv stack (forward + backward through augcl_enc_stack_*) in every f32-class precision against a float64
referee on the same inputs: output error, per-layer dW / db error, and the time of one forward + backward.
usage: python tools/x2_check.py [B] [grad_scale]"""
import sys, time, numpy as np, torch, torch.nn.functional as F
sys.path.insert(0, '.'); sys.path.insert(0, 'tests'); sys.path.insert(0, 'tests/golden')
from augcl_amd import _lib, encoder_ops
from oracle import sac_oracle as so
cuda = torch.device('cuda:0')
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
gscale = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
hp = dict(so.DEFAULTS)
cs = so.synth_state(11, so.critic_shapes((9, 84, 84), 6, hp), gate_safe=False)
ids = so.conv_layer_ids(11)
ws = [cs[f"encoder.shared_cnn.layers.{i}.weight"].to(cuda) for i in ids]
bs = [cs[f"encoder.shared_cnn.layers.{i}.bias"].to(cuda) for i in ids]
rs = np.random.RandomState(1000 + B)
x = torch.from_numpy(rs.randint(0, 256, (B, 9, 84, 84)).astype(np.float32)).to(cuda)
gout = torch.from_numpy((rs.standard_normal((B, 32, 21, 21)) * gscale).astype(np.float32)).to(cuda)


def rl2(a, b):
    return float((a.double() - b).norm() / b.norm().clamp_min(1e-300))


def run(prec, gates=None):
    wl = [w.clone().requires_grad_(True) for w in ws]
    bl = [b.clone().requires_grad_(True) for b in bs]
    out = encoder_ops.conv_stack(x, wl, bl, precision=prec)
    out.backward(gout)
    return out.detach(), [w.grad for w in wl], [b.grad for b in bl]


# float64 referee with its own gates (random He weights: gate flips between two float32-class evaluations are rare and show up
# as isolated larger errors; tests/test_referee_gpu.py forces the gates, this tool is for a quick look and for timing)
wd = [w.double().clone().requires_grad_(True) for w in ws]
bd = [b.double().clone().requires_grad_(True) for b in bs]
h = x.double() / 255.0
for i, (w, b) in enumerate(zip(wd, bd)):
    if i:
        h = torch.relu(h)
    h = F.conv2d(h, w, b, stride=2 if i == 0 else 1)
h.backward(gout.double())
ok = True
for prec in ("f32_mfma", "f32_x3", "f32"):
    out, dws, dbs = run(prec)
    torch.cuda.synchronize()
    e_out = rl2(out, h.detach())
    e_w = [rl2(a, b.grad) for a, b in zip(dws, wd)]
    e_b = [rl2(a, b.grad) for a, b in zip(dbs, bd)]
    fin = all(bool(torch.isfinite(t).all()) for t in [out] + dws + dbs)
    ok &= fin and e_out < 2e-6 and max(e_w) < 2e-5 and max(e_b) < 2e-5
    print(f"{prec:9s} out {e_out:.2e} | dW max {max(e_w):.2e} (layer {int(np.argmax(e_w))}) first {e_w[0]:.2e} | db max {max(e_b):.2e} "
          f"(layer {int(np.argmax(e_b))}) | finite {fin}")
    print("           dW per layer: " + " ".join(f"{v:.1e}" for v in e_w))
    for _ in range(3):
        run(prec)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        run(prec)
    torch.cuda.synchronize()
    print(f"           forward + backward: {(time.perf_counter() - t0) / 10 * 1e3:.3f} ms")
# no-grad dual forward: two weight sets over one batch
w2 = [w * 1.1 for w in ws]
with torch.no_grad():
    for prec in ("f32_x3", "f32"):
        a, b = encoder_ops.conv_stack_dual(x, ws, bs, w2, bs, precision=prec)
        ra = encoder_ops.conv_stack(x, ws, bs, precision="f32_mfma")
        rb = encoder_ops.conv_stack(x, w2, bs, precision="f32_mfma")
        print(f"dual {prec}: set A vs f32_mfma {rl2(a, ra.double()):.2e}  set B {rl2(b, rb.double()):.2e}")
        ok &= rl2(a, ra.double()) < 2e-6 and rl2(b, rb.double()) < 2e-6
print("ALL OK" if ok else "FAILED")
sys.exit(0 if ok else 1)
