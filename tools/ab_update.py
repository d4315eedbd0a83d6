"""A/B two builds of the library on the same box: python tools/ab_update.py LIB_A LIB_B [precision]"""
import subprocess, sys, json, os
prec = sys.argv[3] if len(sys.argv) > 3 else "f32"
code = ("import sys,json,torch; sys.path.insert(0,'.'); from augcl_amd import _lib; _lib.LIB_PATH=%r; "
        "from augcl_amd import bench_update; r=bench_update.run(torch.device('cuda:0'),None,1,0,steps=100,warmup=8,precision=%r,graph=True); print(json.dumps(r['ms_per_update']))")
for rep in range(2):
    for lib in sys.argv[1:3]:
        out = subprocess.run([sys.executable, "-c", code % (os.path.abspath(lib), prec)], capture_output=True, text=True)
        print(prec, os.path.basename(lib), out.stdout.strip().splitlines()[-1] if out.stdout.strip() else out.stderr[-300:])
