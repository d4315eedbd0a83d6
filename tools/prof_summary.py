#!/usr/bin/env python3
"""Run ON the GPU box after `rocprofv3 --kernel-trace --stats -d DIR -- ...`: write DIR.csv (top kernels: name [launch grid],
calls, total/avg/min/max us, %) + DIR.json (launches and GPU-busy ms) and delete the raw trace (tens of MiB).  Durations are
aggregated from the kernel TRACE per (kernel, grid size): the same kernel runs at several batch sizes in one command (the
bench's B=256 headline and its B=1024 extras), and one average over both says nothing.
usage: tools/prof_summary.py DIR n_updates "title" """
import collections, csv, glob, json, os, shutil, sys
d, n_upd, title = sys.argv[1], int(sys.argv[2]), sys.argv[3]
f = glob.glob(os.path.join(d, "*", "*_kernel_trace.csv"))[0]
agg = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    g = r.get("Grid_Size") or "x".join(r.get(k, "1") for k in ("Grid_Size_X", "Grid_Size_Y", "Grid_Size_Z"))
    agg[(r["Kernel_Name"], g)].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
calls = sum(len(v) for v in agg.values())
busy = sum(sum(v) for v in agg.values())
rows = sorted(agg.items(), key=lambda kv: -sum(kv[1]))
with open(d.rstrip("/") + ".csv", "w") as o:
    o.write(f"# rocprofv3 --kernel-trace --stats (top kernels, per launch grid), {title}\n")
    o.write(f"# whole trace (warm-up and graph capture included): {calls} kernel launches, GPU busy {busy / 1e3:.1f} ms, {n_upd} timed update() calls\n")
    o.write("name,grid,calls,total_us,avg_us,median_us,min_us,max_us,pct\n")
    for (name, grid), v in rows[:50]:
        v = sorted(v)
        o.write('"%s",%s,%d,%.1f,%.2f,%.2f,%.2f,%.2f,%.2f\n' % (name[:110], grid, len(v), sum(v), sum(v) / len(v), v[len(v) // 2], v[0], v[-1],
                                                            100.0 * sum(v) / busy))
json.dump({"title": title, "launches_total": calls, "gpu_busy_ms_total": busy / 1e3, "updates_timed": n_upd}, open(d.rstrip("/") + ".json", "w"))
shutil.rmtree(d)
