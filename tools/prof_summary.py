#!/usr/bin/env python3
"""Run ON the GPU box after `rocprofv3 --kernel-trace --stats -d DIR -- ...`: write DIR.csv (top kernels: name, calls,
total/avg/min/max us, %) + DIR.json (launches and GPU-busy ms per update) and delete the raw trace (tens of MiB).
usage: tools/prof_summary.py DIR n_updates "title" """
import csv, glob, json, os, shutil, sys
d, n_upd, title = sys.argv[1], int(sys.argv[2]), sys.argv[3]
f = glob.glob(os.path.join(d, "*", "*_kernel_stats.csv"))[0]
rows = list(csv.DictReader(open(f)))
calls = sum(int(r["Calls"]) for r in rows)
busy = sum(float(r["TotalDurationNs"]) for r in rows) / 1e6
with open(d.rstrip("/") + ".csv", "w") as o:
    o.write(f"# rocprofv3 --kernel-trace --stats (top kernels), {title}\n")
    o.write(f"# whole trace (warm-up and graph capture included): {calls} kernel launches, GPU busy {busy:.1f} ms, {n_upd} timed update() calls\n")
    o.write("name,calls,total_us,avg_us,min_us,max_us,pct\n")
    for r in rows[:45]:
        o.write('"%s",%s,%.1f,%.2f,%.2f,%.2f,%s\n' % (r["Name"][:110], r["Calls"], float(r["TotalDurationNs"]) / 1e3, float(r["AverageNs"]) / 1e3,
                                                  float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3, r["Percentage"]))
json.dump({"title": title, "launches_total": calls, "gpu_busy_ms_total": busy, "updates_timed": n_upd}, open(d.rstrip("/") + ".json", "w"))
shutil.rmtree(d)
