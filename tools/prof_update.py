"""One process running `steps` SVEA update() calls (+8 warm-up) for rocprofv3: python tools/prof_update.py STEPS [f32|bf16] [graph]"""
import sys, os, json, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from augcl_amd import bench_update
print(json.dumps(bench_update.run(torch.device("cuda:0"), None, 1, 0, steps=int(sys.argv[1]) if len(sys.argv) > 1 else 10, warmup=3,
                                  precision=sys.argv[2] if len(sys.argv) > 2 else "f32", graph=len(sys.argv) > 3 and sys.argv[3] == "graph", min_s=0.0)))
