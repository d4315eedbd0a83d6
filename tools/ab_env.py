"""A/B an environment switch on one box: python tools/ab_env.py VAR precision"""
import subprocess, sys, os
var, prec = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "f32")
code = ("import sys,json,torch; sys.path.insert(0,'.'); from augcl_amd import bench_update; "
        "r=bench_update.run(torch.device('cuda:0'),None,1,0,steps=100,warmup=8,precision=%r,graph=True); print(json.dumps(r['ms_per_update']))")
for rep in range(2):
    for val in ("0", "1"):
        env = dict(os.environ); env[var] = val
        out = subprocess.run([sys.executable, "-c", code % prec], capture_output=True, text=True, env=env)
        print(prec, var, val, out.stdout.strip().splitlines()[-1] if out.stdout.strip() else out.stderr[-400:])
