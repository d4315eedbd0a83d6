import os, sys
import numpy as np, torch, torch.nn.functional as F
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from augcl_amd import _lib
cuda = torch.device("cuda:0")
call, p = _lib.call, _lib.ptr
st = None
rs = np.random.RandomState(0)
for (B, Ho) in [(5, 33), (16, 25), (4, 21), (16, 37), (64, 31)]:
    dz = torch.from_numpy(rs.standard_normal((B, 32, Ho, Ho)).astype(np.float32)).to(cuda)
    w = torch.from_numpy((rs.standard_normal((32, 32, 3, 3)) * 0.08).astype(np.float32)).to(cuda)
    a_prev = torch.relu(torch.from_numpy(rs.standard_normal((B, 32, Ho + 2, Ho + 2)).astype(np.float32))).to(cuda)
    ref = F.conv_transpose2d(dz.cpu(), w.cpu()) * (a_prev.cpu() > 0)
    bad = 0; first = None
    for it in range(30):
        out = torch.full_like(a_prev, float("nan"))
        call("augcl_enc_conv32_backward_data", p(dz), p(w), p(a_prev), p(out), B, Ho, Ho, torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        err = (out.cpu() - ref).abs().max().item() / ref.abs().max().item()
        if not (err < 1e-5):
            bad += 1
            if first is None:
                d = (out.cpu() - ref).abs(); idx = (d > 1e-3 * ref.abs().max()).nonzero()
                first = (err, idx.shape[0], idx[:6].tolist(), bool(torch.isnan(out).any()))
    print("bwd_data", B, Ho, "bad runs", bad, first)
    # wgrad
    act = a_prev
    dzz = torch.from_numpy(rs.standard_normal((B, 32, Ho, Ho)).astype(np.float32)).to(cuda)
    wr = w.cpu().clone().requires_grad_(True); br = torch.zeros(32, requires_grad=True)
    (F.conv2d(act.cpu(), wr, br) * dzz.cpu()).sum().backward()
    nws = _lib.load().augcl_enc_wgrad_workspace_bytes(32)
    ws = torch.empty(nws // 4, device=cuda)
    bad = 0; first = None
    for it in range(30):
        dw = torch.full_like(w, float("nan")); db = torch.full((32,), float("nan"), device=cuda)
        call("augcl_enc_conv32_backward_weights", p(act), p(dzz), p(dw), p(db), p(ws), nws, B, Ho + 2, Ho + 2, torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        e1 = (dw.cpu() - wr.grad).abs().max().item() / wr.grad.abs().max().item()
        e2 = (db.cpu() - br.grad).abs().max().item() / br.grad.abs().max().item()
        if not (e1 < 2e-5 and e2 < 2e-5):
            bad += 1; first = first or (e1, e2)
    print("wgrad   ", B, Ho, "bad runs", bad, first)
