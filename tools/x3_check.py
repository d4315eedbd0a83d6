"""Ad-hoc check of the x3 (split-bf16) 32->32 layer kernels against float64 ATen and the native f32 MFMA kernels."""
import sys, torch, torch.nn.functional as F
sys.path.insert(0, '.')
from augcl_amd import _lib
L = _lib.load()
cuda = torch.device('cuda:0')
st = _lib.current_stream(cuda)
p = lambda t: None if t is None else t.data_ptr()
def rl2(a, b): return float((a.double() - b).norm() / b.norm().clamp_min(1e-300))
def mx(a, b): return float((a.double() - b).abs().max() / b.abs().max().clamp_min(1e-300))
shapes = [(2, 12, 70), (3, 70, 12), (1, 10, 100), (2, 33, 35), (5, 11, 11), (1, 3, 9), (1, 5, 6), (4, 41, 41), (256, 41, 41), (256, 23, 23), (64, 39, 39), (7, 31, 31), (1, 41, 41)]
if len(sys.argv) > 1: shapes = [tuple(int(v) for v in a.split(',')) for a in sys.argv[1:]]
bad = 0
for (B, H, W) in shapes:
    g = torch.Generator(device=cuda).manual_seed(B * 1000 + H * 10 + W)
    x = torch.relu(torch.randn(B, 32, H, W, device=cuda, generator=g))
    w = torch.randn(32, 32, 3, 3, device=cuda, generator=g) * 0.06
    bias = torch.randn(32, device=cuda, generator=g) * 0.1
    dz = torch.randn(B, 32, H - 2, W - 2, device=cuda, generator=g)
    res = {}
    for pre in ("augcl_enc_conv32", "augcl_enc_x3_conv32"):
        y = torch.full((B, 32, H - 2, W - 2), float('nan'), device=cuda)
        _lib.call(pre + "_forward", p(x), p(w), p(bias), p(y), B, H, W, 1, st)
        dx = torch.full((B, 32, H, W), float('nan'), device=cuda)
        _lib.call(pre + "_backward_data", p(dz), p(w), p(x), p(dx), B, H - 2, W - 2, st)
        dx0 = torch.full((B, 32, H, W), float('nan'), device=cuda)
        _lib.call(pre + "_backward_data", p(dz), p(w), None, p(dx0), B, H - 2, W - 2, st)
        out = [y, dx, dx0]
        if W - 2 >= 8:
            ws = torch.empty(L.augcl_enc_wgrad_workspace_bytes(32) // 4, device=cuda)
            dw, db = torch.full((32, 32, 3, 3), float('nan'), device=cuda), torch.full((32,), float('nan'), device=cuda)
            _lib.call(pre + "_backward_weights", p(x), p(dz), p(dw), p(db), p(ws), ws.numel() * 4, B, H, W, st)
            out += [dw, db]
        torch.cuda.synchronize()
        res[pre] = out
    xd, wd = x.double(), w.double().clone().requires_grad_(True)
    want_y = torch.relu(F.conv2d(xd, wd, bias.double()))
    want_dx0 = F.conv_transpose2d(dz.double(), w.double())
    want_dx = want_dx0 * (x > 0)
    F.conv2d(xd, wd).backward(dz.double())
    wants = [want_y.detach(), want_dx, want_dx0, wd.grad, dz.double().sum((0, 2, 3))]
    names = ["fwd", "bwd-data(gated)", "bwd-data", "dW", "db"]
    line = f"B={B} H={H} W={W}: "
    for i, nm in enumerate(names):
        if i >= len(res["augcl_enc_conv32"]): break
        a, b = res["augcl_enc_conv32"][i], res["augcl_enc_x3_conv32"][i]
        ea, eb = rl2(a, wants[i]), rl2(b, wants[i])
        ma, mb = mx(a, wants[i]), mx(b, wants[i])
        ok = eb <= max(2 * ea, 1e-6) and mb <= 3e-5 and not torch.isnan(b).any()
        bad += (not ok)
        line += f"{nm} f32 {ea:.1e}/{ma:.1e} x3 {eb:.1e}/{mb:.1e}{'' if ok else ' <-- BAD'} | "
    print(line)
print("BAD" if bad else "ALL OK", bad)
sys.exit(1 if bad else 0)
