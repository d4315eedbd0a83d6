"""Per-layer timing of the 32->32 layer kernels: split-bf16 (x3) vs f16-pair (x2).  python tools/x2_bench.py [B] [tuning]
(tuning: load tools/libaugcl_tuning.so -- AUGCL_X2_ABL / AUGCL_X2_RT are read by that build only)"""
import ctypes, os, sys, torch
sys.path.insert(0, '.')
from augcl_amd import _lib
if len(sys.argv) > 2 and sys.argv[2] == "tuning":
    _lib.LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'libaugcl_tuning.so')
L = _lib.load()
dev = torch.device('cuda:0')
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
P = ctypes.c_void_p
p = lambda t: P(t.data_ptr())
def chk(rc, what):
    if rc != 0: raise RuntimeError(f"{what}: rc {rc} {L.augcl_last_error()}")
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): fn()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / n * 1e3)
    return best
w = torch.randn(32, 32, 3, 3, device=dev) * 0.05
bias = torch.zeros(32, device=dev)
ws = torch.empty(L.augcl_enc_wgrad_workspace_bytes(32) // 4, device=dev)
dw, db = torch.empty(32, 32, 3, 3, device=dev), torch.empty(32, device=dev)
frag = torch.empty(2 * L.augcl_enc_frag_workspace_bytes(1, 1), dtype=torch.uint8, device=dev)
fb = L.augcl_enc_frag_workspace_bytes(1, 1)
warr = (ctypes.c_void_p * 1)(w.data_ptr())
chk(L.augcl_internal_x2_prep_weights(warr, 1, 0, p(frag), None, 0, st), "prep0")
chk(L.augcl_internal_x2_prep_weights(warr, 1, 1, P(frag.data_ptr() + fb), None, 0, st), "prep1")
am = torch.zeros(8, device=dev)
def amax(t, slot):
    am[slot] = 0
    chk(L.augcl_internal_amax_f32(p(t), ctypes.c_size_t(t.numel()), P(am.data_ptr() + 4 * slot), st), "amax")
tot = {}
print("H    fwd x3 -> x2       bwd x3 -> x2       wgrad x3 -> x2    (us)   x2: fwd TB/s")
for H in range(41, 21, -2):
    x = torch.relu(torch.randn(B, 32, H, H, device=dev))  # post-ReLU statistics (half zeros): all-dense operands cost the
    # matrix pipe more power and hide what the kernels are bound by in the real stack
    y = torch.empty(B, 32, H - 2, H - 2, device=dev)
    dz = torch.randn(B, 32, H - 2, H - 2, device=dev) * 1e-3 * (torch.rand(B, 32, H - 2, H - 2, device=dev) > 0.5)  # gated gradient
    dx = torch.empty(B, 32, H, H, device=dev)
    amax(x, 0); amax(dz, 1)
    gbits = torch.empty(B * H * H, dtype=torch.int32, device=dev)  # gate bits of x (the backward-data mask), as the forward pass of
    chk(L.augcl_internal_gate_bits(p(x), p(gbits), B, H, H, st), "bits")  # the layer below would have left them
    ybits = torch.empty(B * (H - 2) * (H - 2), dtype=torch.int32, device=dev)
    row = []
    for kind in ("forward", "backward_data", "backward_weights"):
        for x2 in (False, True):
            if not x2:
                pre = "augcl_enc_x3_conv32_"
                if kind == "forward": fn = lambda: _lib.call(pre + kind, p(x), p(w), p(bias), p(y), B, H, H, 1, st)
                elif kind == "backward_data": fn = lambda: _lib.call(pre + kind, p(dz), p(w), p(x), p(dx), B, H - 2, H - 2, st)
                else: fn = lambda: _lib.call(pre + kind, p(x), p(dz), p(dw), p(db), p(ws), ctypes.c_size_t(ws.numel() * 4), B, H, H, st)
            else:
                if kind == "forward":
                    fn = lambda: chk(L.augcl_internal_x2_conv32_forward(p(x), p(bias), None, p(y), B, H, H, 1, p(frag), None, P(am.data_ptr()), P(am.data_ptr() + 8), p(ybits), st), "fwd")
                elif kind == "backward_data":
                    fn = lambda: chk(L.augcl_internal_x2_conv32_backward_data(p(dz), p(x), p(dx), B, H - 2, H - 2, P(frag.data_ptr() + fb), P(am.data_ptr() + 4), P(am.data_ptr() + 12), p(gbits), st), "bwd")
                else:
                    fn = lambda: chk(L.augcl_internal_x2_conv32_backward_weights(p(x), p(dz), p(dw), p(db), p(ws), ctypes.c_size_t(ws.numel() * 4), B, H, H, P(am.data_ptr()), P(am.data_ptr() + 4), st), "wg")
            t = timeit(fn)
            if x2 and kind != "backward_weights" and hasattr(L, "augcl_tuning_x2_read_clk"):
                c = (ctypes.c_ulonglong * 4)()
                L.augcl_tuning_x2_read_clk(c)
                print(f"      {kind}: wave 0 of WG 0: {c[0]} cycles, {c[1] / 100:.1f} us -> {c[0] / max(c[1], 1) / 10:.2f} GHz, {c[2]} tiles, {c[0] / max(c[2], 1):.0f} cycles / tile")
            row.append(t)
            tot[(kind, x2)] = tot.get((kind, x2), 0.0) + t
    byt = B * 32 * 4 * (H * H + (H - 2) * (H - 2))
    print("%-4d %6.1f -> %6.1f   %6.1f -> %6.1f   %6.1f -> %6.1f        %5.2f" % (H, *row, byt / row[1] / 1e6))
print("sum  " + "   ".join("%6.1f -> %6.1f" % (tot[(k, False)], tot[(k, True)]) for k in ("forward", "backward_data", "backward_weights")))
