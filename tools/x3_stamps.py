"""Phase breakdown of one x3 conv launch from in-kernel s_memtime stamps (tuning build: tools/build_tuning.sh).
python tools/x3_stamps.py [B] [H] [mode: fwd|bwd]"""
import ctypes, os, sys, collections, torch
sys.path.insert(0, '.')
from augcl_amd import _lib
_lib.LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'libaugcl_tuning.so')
L = _lib.load()
L.augcl_tuning_x3_set_stamps.argtypes = [ctypes.c_void_p]
dev = torch.device('cuda:0')
B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
H = int(sys.argv[2]) if len(sys.argv) > 2 else 41
mode = sys.argv[3] if len(sys.argv) > 3 else "fwd"
st = torch.cuda.current_stream().cuda_stream
p = lambda t: t.data_ptr()
w = torch.randn(32, 32, 3, 3, device=dev) * 0.05
bias = torch.zeros(32, device=dev)
x = torch.rand(B, 32, H, H, device=dev)
y = torch.empty(B, 32, H - 2, H - 2, device=dev)
dz = torch.randn(B, 32, H - 2, H - 2, device=dev)
dx = torch.empty(B, 32, H, H, device=dev)
dbg = torch.zeros(8192, dtype=torch.int64, device=dev)
def run():
    if mode == "fwd": _lib.call("augcl_enc_x3_conv32_forward", p(x), p(w), p(bias), p(y), B, H, H, 1, st)
    else: _lib.call("augcl_enc_x3_conv32_backward_data", p(dz), p(w), p(x), p(dx), B, H - 2, H - 2, st)
for _ in range(30): run()          # warm: clocks settle under load
L.augcl_tuning_x3_set_stamps(p(dbg))
for _ in range(5): run()
L.augcl_tuning_x3_set_stamps(None)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): run()
e1.record(); torch.cuda.synchronize()
print(f"B={B} H={H} {mode}: {e0.elapsed_time(e1) / 20 * 1e3:.1f} us per launch")
d = dbg.cpu().numpy().astype('uint64')
n = int(d[8189])
rt0, rt1 = int(d[8190]), int(d[8191])
rec = [(int(v) >> 8, int(v) & 255) for v in d[:n]]
t0, t1 = rec[0][0], rec[-1][0]
clk = (t1 - t0) / max(1, (rt1 - rt0)) * 100e6
print(f"compute wave 0 of WG 0: {n} stamps, {t1 - t0} cycles, {(rt1 - rt0) / 100:.1f} us => shader clock {clk / 1e9:.2f} GHz")
names = {(0, 1): "stage f32 weights + barrier", (1, 2): "weight fragments (split)", (2, 3): "wait: first strip staged", (3, 11): "tile setup", (15, 11): "tile setup",
         (13, 11): "tile setup", (11, 12): "MFMA chain (108) + operand reads", (12, 13): "epilogue (gate, stores)", (13, 14): "strip end", (14, 15): "barrier wait", (3, 14): "no tile", (15, 14): "no tile"}
agg = collections.defaultdict(lambda: [0, 0])
for (ta, ia), (tb, ib) in zip(rec[:-1], rec[1:]):
    k = names.get((ia, ib), f"{ia}->{ib}")
    agg[k][0] += tb - ta; agg[k][1] += 1
for k, (c, m) in sorted(agg.items(), key=lambda kv: -kv[1][0]):
    print(f"  {k:38s} {c:9d} cycles {100.0 * c / (t1 - t0):5.1f} %  x{m:4d}  avg {c / m:8.0f}")
# loader wave
recl = [(int(v) >> 8, int(v) & 255) for v in d[4096:8000] if int(v)]
if recl:
    namesl = {(20, 24): "wait for the strip's loads", (24, 25): "wait for the NEXT strip's loads (vmcnt 0)", (25, 21): "commit (split + LDS writes)", (24, 21): "commit (split + LDS writes)", (20, 21): "commit (split + LDS writes)", (21, 22): "issue loads", (22, 23): "barrier wait", (23, 20): "loop", (0, 1): "stage f32 weights + barrier", (1, 20): "first issue"}
    aggl = collections.defaultdict(lambda: [0, 0])
    for (ta, ia), (tb, ib) in zip(recl[:-1], recl[1:]):
        k = namesl.get((ia, ib), f"{ia}->{ib}")
        aggl[k][0] += tb - ta; aggl[k][1] += 1
    tot = recl[-1][0] - recl[0][0]
    print(f"loader wave 4 of WG 0: {len(recl)} stamps, {tot} cycles")
    for k, (c, m) in sorted(aggl.items(), key=lambda kv: -kv[1][0]):
        print(f"  {k:38s} {c:9d} cycles {100.0 * c / tot:5.1f} %  x{m:4d}  avg {c / m:8.0f}")
