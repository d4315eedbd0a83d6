import sys
import numpy as np, torch, torch.nn.functional as F
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from augcl_amd import encoder_hip, encoder_ops
from test_chain_gpu import _stack, _rel, _kept
cuda = torch.device('cuda:0')
n_layers, H, W = 11, 84, 84
for B in [int(a) for a in sys.argv[1:]] or [300]:
    x = torch.from_numpy(np.random.RandomState(B + H).randint(0, 256, (B, 9, H, W)).astype(np.float32)).to(cuda)
    ws, bs = _stack(43, n_layers, dev=cuda)
    hs = [41 - 2 * i for i in range(n_layers)]
    w64 = [w.double().requires_grad_(True) for w in ws]; b64 = [b.double().requires_grad_(True) for b in bs]
    y = x.double() / 255.0
    for i in range(n_layers):
        y = F.conv2d(y, w64[i], b64[i], stride=2 if i == 0 else 1)
        if i < n_layers - 1: y = torch.relu(y)
    g = torch.from_numpy(np.random.RandomState(5).standard_normal(tuple(y.shape)).astype(np.float32)).to(cuda)
    (y * g.double()).sum().backward()
    for chains in (True, False, True):
        encoder_hip._CHAINS_GRAD = chains
        wg = [w.clone().requires_grad_(True) for w in ws]; bg = [b.clone().requires_grad_(True) for b in bs]
        out = encoder_ops.conv_stack(x, wg, bg, precision="f32")
        acts, amax, words = _kept(out, B, hs, hs)
        amax0 = amax.clone(); acts0 = [a.clone() for a in acts]; words0 = [w_.clone() for w_ in words]
        (out * g).sum().backward(); torch.cuda.synchronize()
        acts1, amax1, words1 = _kept(out, B, hs, hs) if False else (acts, None, words)
        print(f"B={B} chains={chains}: arena changed by backward: acts {[bool((a != b).any()) for a, b in zip(acts, acts0)]}")
        print("   dW err", [f"{_rel(w.grad, r.grad):.1e}" for w, r in zip(wg, w64)])
        print("   db err", [f"{_rel(b.grad, r.grad):.1e}" for b, r in zip(bg, b64)])
        print("   amax", [f"{float(v):.4f}" for v in amax0], flush=True)
