"""A/B of update() between two source trees on ONE box: runs augcl_amd.bench_update of each tree in its own process,
alternating, and prints ms per update.  usage: python tools/ab_trees.py <treeA> <treeB> [rounds] [precision]"""
import json, subprocess, sys
trees = sys.argv[1:3]
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 3
prec = sys.argv[4] if len(sys.argv) > 4 else "f32"
code = ("import sys, json, os, torch; sys.path.insert(0, sys.argv[1].split(':')[0]); from augcl_amd import bench_update, encoder_hip; "
        "encoder_hip._GATE_BITS = not sys.argv[1].endswith(':nobits'); "
        "r = bench_update.run(torch.device('cuda:0'), None, 1, 0, steps=200, warmup=10, B=256, precision=sys.argv[2], graph=True); "
        "print('RES', json.dumps(r['ms_per_update']))")
res = {t: [] for t in trees}
for _ in range(rounds):
    for t in trees:
        out = subprocess.run([sys.executable, "-c", code, t, prec], capture_output=True, text=True)
        line = [l for l in out.stdout.splitlines() if l.startswith("RES")]
        if not line:
            print(out.stdout[-500:], out.stderr[-1500:])
            sys.exit(1)
        res[t].append(float(line[0][4:]))
        print(t, res[t][-1], flush=True)
for t in trees:
    print(t, "min %.4f  mean %.4f" % (min(res[t]), sum(res[t]) / len(res[t])))
