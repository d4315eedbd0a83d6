"""Where a compute wave's cycles go inside one chain launch (tuning build: tools/build_tuning.sh).
python tools/chain_stamps.py [B] [keep]  -> per launch of the two chains of the 84x84 stack: cycles per round by phase
(keep: the gradient-carrying launch, k_conv32_chain<true>, which also stores every layer, its gate words and maxima)"""
import ctypes, os, sys
import numpy as np
import torch
sys.path.insert(0, '.')
from augcl_amd import _lib
_lib.LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'libaugcl_tuning.so')
L = _lib.load()
dev = torch.device('cuda:0')
B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
KEEP = len(sys.argv) > 2 and sys.argv[2] == "keep"
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
L.augcl_internal_chain_forward.argtypes = [ctypes.c_void_p] * 6 + [ctypes.c_int] * 6 + [ctypes.c_void_p] * 3
L.augcl_internal_chain_forward_keep.argtypes = [ctypes.c_void_p] * 4 + [ctypes.c_int] * 6 + [ctypes.c_void_p] * 6
fb = L.augcl_enc_frag_workspace_bytes(1, 1)
for (Hi, nl, parts) in ((41, 4, 2), (33, 6, 1), (41, 1, 1), (41, 2, 1), (33, 1, 1)):
    if L.augcl_internal_chain_rounds(Hi, Hi, nl, parts) < 0:
        print(Hi, nl, parts, "does not fit"); continue
    x = torch.relu(torch.randn(B, 32, Hi, Hi, device=dev))
    ws = [torch.randn(32, 32, 3, 3, device=dev) * 0.083 for _ in range(nl)]
    bs = [torch.randn(32, device=dev) * 0.1 for _ in range(nl)]
    frag = torch.empty(nl * fb, dtype=torch.uint8, device=dev)
    warr = (ctypes.c_void_p * nl)(*[w.data_ptr() for w in ws])
    assert L.augcl_internal_x2_prep_weights(warr, nl, 2, ctypes.c_void_p(frag.data_ptr()), None, 0, st) == 0  # mode 2: forward fragments, unscaled residuals
    am = torch.zeros(4, device=dev); am[0] = x.abs().max()
    out = torch.empty(B, 32, Hi - 2 * nl, Hi - 2 * nl, device=dev)
    fa = (ctypes.c_void_p * nl)(*[frag.data_ptr() + j * fb for j in range(nl)])
    ba = (ctypes.c_void_p * nl)(*[b.data_ptr() for b in bs])
    acts = [torch.empty(B, 32, Hi - 2 * (j + 1), Hi - 2 * (j + 1), device=dev) for j in range(nl - 1)]
    bits = [torch.empty(B, Hi - 2 * (j + 1), Hi - 2 * (j + 1), dtype=torch.int32, device=dev) for j in range(nl)]
    aa = (ctypes.c_void_p * max(nl - 1, 1))(*[t.data_ptr() for t in acts]) if nl > 1 else (ctypes.c_void_p * 1)(None)
    bb = (ctypes.c_void_p * nl)(*[t.data_ptr() for t in bits])
    amk = torch.zeros(8, device=dev)
    def run():
        if KEEP:
            rc = L.augcl_internal_chain_forward_keep(x.data_ptr(), out.data_ptr(), fa, ba, nl, B, Hi, Hi, parts, 1, am.data_ptr(), amk.data_ptr() + 28,
                                                     aa, bb, amk.data_ptr(), st)
        else:
            rc = L.augcl_internal_chain_forward(x.data_ptr(), out.data_ptr(), fa, ba, None, None, nl, B, Hi, Hi, parts, 0, am.data_ptr(), None, st)
        assert rc == 0, L.augcl_last_error()
    for _ in range(5): run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): run()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    c = (ctypes.c_ulonglong * 8)()
    L.augcl_tuning_chain_read_clk(c)
    r = max(c[5], 1)
    print(f"{Hi}x{Hi} nl={nl} parts={parts} B={B}: {us:.1f} us/launch; wave 0 of WG 0: {c[6]} cycles = {c[7] / 100:.1f} us ({c[6] / max(c[7], 1) / 10:.2f} GHz), {c[5]} rounds;"
          f" per round: mma {c[0] / r:.0f}  wait {c[1] / r:.0f}  epilogue {c[2] / r:.0f}  layer {c[3] / r:.0f}  item {c[4] / r:.0f}  total {c[6] / r:.0f}")
