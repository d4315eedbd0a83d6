"""A/B on ONE box: SVEA update() with one of the library's internal kernel switches off / on.
python tools/ab_update_switch.py <augcl_internal_proj_x3|augcl_internal_first_layer_x3> [synthetic|ring]"""
import ctypes, sys
import torch
sys.path.insert(0, '.')
from augcl_amd import _lib, bench_update  # noqa: E402

switch = sys.argv[1]
replay = sys.argv[2] if len(sys.argv) > 2 else "synthetic"
raw = ctypes.CDLL(_lib.LIB_PATH)
dev = torch.device('cuda:0')
for rep in range(2):
    for on in (0, 1):
        getattr(raw, switch)(on)
        r = bench_update.run(dev, None, 1, 0, steps=100, B=256, precision="f32", graph=True, min_s=1.0, replay=replay)
        print(f"f32 {switch}={on} replay={replay}: {r['ms_per_update']:.4f} ms/update  ({r['grad_steps_per_s']:.1f} grad-steps/s)", flush=True)
getattr(raw, switch)(1)
