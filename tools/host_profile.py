"""cProfile of the HOST side of eager update() calls (bf16 mode is launch-bound: host time is what limits it):
python tools/host_profile.py [bf16|f32] [steps]"""
import cProfile, pstats, sys, os, io, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from augcl_amd import bench_update
prec = sys.argv[1] if len(sys.argv) > 1 else "bf16"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 60
dev = torch.device("cuda:0")
bench_update.run(dev, None, 1, 0, steps=8, warmup=3, precision=prec, graph=False, min_s=0.0)  # warm everything up
pr = cProfile.Profile()
pr.enable()
res = bench_update.run(dev, None, 1, 0, steps=steps, warmup=3, precision=prec, graph=False, min_s=0.0)
pr.disable()
print(res["ms_per_update"], "ms per update (eager, profiled)")
out = io.StringIO()
pstats.Stats(pr, stream=out).sort_stats("tottime").print_stats(28)
print(out.getvalue()[:6000])
