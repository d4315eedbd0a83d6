"""Walk the conv stack backward layer by layer with the native f32 MFMA and the split-bf16 (x3) kernels on the SAME inputs
(the float64 referee's activations) and report every layer's dz / dW error against float64."""
import sys, numpy as np, torch, torch.nn.functional as F
sys.path.insert(0, '.'); sys.path.insert(0, 'tests'); sys.path.insert(0, 'tests/golden')
from augcl_amd import _lib
from oracle import sac_oracle as so
L = _lib.load()
cuda = torch.device('cuda:0')
st = _lib.current_stream(cuda)
p = lambda t: None if t is None else t.data_ptr()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
hp = dict(so.DEFAULTS)
cs = so.synth_state(11, so.critic_shapes((9, 84, 84), 6, hp), gate_safe=False)
ids = so.conv_layer_ids(11)
ws = [cs[f"encoder.shared_cnn.layers.{i}.weight"].to(cuda) for i in ids]
bs = [cs[f"encoder.shared_cnn.layers.{i}.bias"].to(cuda) for i in ids]
rs = np.random.RandomState(1000 + B)
x = torch.from_numpy(rs.randint(0, 256, (B, 9, 84, 84)).astype(np.float32)).to(cuda)
gout = torch.from_numpy(rs.standard_normal((B, 32, 21, 21)).astype(np.float32)).to(cuda)
# float64 referee: activations and per-layer dz
h = x.double() / 255.0
acts, pre = [], []
for i, (w, b) in enumerate(zip(ws, bs)):
    if i:
        pre.append(h); h = torch.relu(h)
    acts.append(h)  # input of layer i (post-ReLU for i >= 1; acts[0] = x/255)
    h = F.conv2d(h, w.double(), b.double(), stride=2 if i == 0 else 1)
dz64 = [None] * 11
dz64[10] = gout.double()
for i in range(10, 0, -1):
    dz64[i - 1] = F.conv_transpose2d(dz64[i], ws[i].double()) * (acts[i] > 0)
def rl2(a, b): return float((a.double() - b).norm() / b.norm().clamp_min(1e-300))
ws_buf = torch.empty(L.augcl_enc_wgrad_workspace_bytes(32) // 4, device=cuda)
for pre_ in ("augcl_enc_conv32", "augcl_enc_x3_conv32"):
    print(pre_)
    dz = gout.clone()
    for i in range(10, 0, -1):
        a_prev = acts[i].float().contiguous()
        H = a_prev.shape[2]
        dw, db = torch.empty(32, 32, 3, 3, device=cuda), torch.empty(32, device=cuda)
        _lib.call(pre_ + "_backward_weights", p(a_prev), p(dz), p(dw), p(db), p(ws_buf), ws_buf.numel() * 4, B, H, H, st)
        # exact-input variant: dW from the float64 dz rounded to f32 (isolates this layer's own error)
        dw1, db1 = torch.empty(32, 32, 3, 3, device=cuda), torch.empty(32, device=cuda)
        dzr = dz64[i].float().contiguous()
        _lib.call(pre_ + "_backward_weights", p(a_prev), p(dzr), p(dw1), p(db1), p(ws_buf), ws_buf.numel() * 4, B, H, H, st)
        wd = ws[i].double().clone().requires_grad_(True)
        F.conv2d(acts[i], wd).backward(dz64[i])
        dx = torch.empty(B, 32, H, H, device=cuda)
        _lib.call(pre_ + "_backward_data", p(dz), p(ws[i]), p(a_prev), p(dx), B, H - 2, H - 2, st)
        dx1 = torch.empty(B, 32, H, H, device=cuda)
        _lib.call(pre_ + "_backward_data", p(dzr), p(ws[i]), p(a_prev), p(dx1), B, H - 2, H - 2, st)
        torch.cuda.synchronize()
        print(f"  layer {i:2d} H={H}: dz_in err {rl2(dz, dz64[i]):.2e} | dW chain {rl2(dw, wd.grad):.2e} own {rl2(dw1, wd.grad):.2e} | db chain {rl2(db, dz64[i].sum((0,2,3))):.2e} own {rl2(db1, dz64[i].sum((0,2,3))):.2e} | dx chain {rl2(dx, dz64[i-1]):.2e} own {rl2(dx1, dz64[i-1]):.2e}  |dz| rms {float(dz64[i].square().mean().sqrt()):.2e} zeros {float((dz64[i]==0).double().mean()):.2f}")
        dz = dx
