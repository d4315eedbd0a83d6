#!/usr/bin/env python3
"""Run ON the GPU box after `rocprofv3 --kernel-trace --output-format csv -d DIR -- python tools/prof_update.py N f32 graph`:
write DIR_sequence.txt = the kernels of the LAST update() of the trace in launch order (start offset, duration, gap to
the previous kernel's end, name) and delete the raw trace.   usage: tools/prof_sequence.py DIR launches_per_update"""
import csv, glob, os, shutil, sys
d, n = sys.argv[1], int(sys.argv[2])
f = glob.glob(os.path.join(d, "*", "*_kernel_trace.csv"))[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
last = rows[-n:]
t0 = int(last[0]["Start_Timestamp"])
with open(d.rstrip("/") + "_sequence.txt", "w") as o:
    o.write(f"# last {n} kernel launches of the trace (one graph-replayed update()); us from the first one\n")
    o.write("#  i   start_us    dur_us    gap_us  kernel\n")
    prev_end, busy = None, 0.0
    for i, r in enumerate(last):
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        gap = 0.0 if prev_end is None else (s - prev_end) / 1e3
        busy += (e - s) / 1e3
        o.write("%4d %9.1f %9.2f %9.2f  %s\n" % (i, (s - t0) / 1e3, (e - s) / 1e3, gap, r["Kernel_Name"][:120]))
        prev_end = e
    o.write("# span %.1f us, kernels busy %.1f us\n" % ((prev_end - t0) / 1e3, busy))
shutil.rmtree(d)
