"""update() fed by the synthetic float batches vs by the product's uint8 replay ring (graph replay, f32): ms per update"""
import sys, torch
sys.path.insert(0, ".")
from augcl_amd import bench_update
for rp in ("synthetic", "ring", "synthetic", "ring"):
    r = bench_update.run(torch.device("cuda:0"), None, 1, 0, steps=100, warmup=10, B=256, precision="f32", graph=True, replay=rp)
    print(rp, round(r["ms_per_update"], 4), flush=True)
