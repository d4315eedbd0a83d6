#!/bin/bash
# per-kernel time of the whole-stack check (tools/x2_check.py: f32_mfma, f32_x3, f32 precisions back to back).  Run ON the GPU box.
set -e
B=${1:-512}
ROOT=$(pwd)
cd /tmp && export TMPDIR=/tmp && cd "$ROOT"
O=gpurun_out
rocprofv3 --kernel-trace --output-format csv -d $O/x2_trace -- python tools/x2_check.py $B > /dev/null 2>&1 || true
python - <<PY
import csv, glob, collections
rows = []
for f in glob.glob("$O/x2_trace/**/*kernel_trace.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the LAST forward+backward of each precision: take the last launches of each kernel family in order
seq = [(r["Kernel_Name"].split("(")[0][:60], (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3) for r in rows]
agg = collections.OrderedDict()
for n, d in seq:
    agg.setdefault(n, []).append(d)
for n, v in agg.items():
    if "conv" in n or "wgrad" in n or "prep" in n or "amax" in n or "reduce" in n:
        print(f"{n:62s} n={len(v):4d} avg {sum(v)/len(v):7.2f} min {min(v):7.2f}")
def last_pass(name, cnt):
    v = [d for n, d in seq if n.startswith(name)]
    return v[-cnt:]
for nm in ("void augcl::k_conv32_x3<0>", "void augcl::k_conv32_x2<0>", "void augcl::k_conv32_x3<1>", "void augcl::k_conv32_x2<1>", "augcl::k_wgrad32_x3", "augcl::k_wgrad32_x2"):
    v = last_pass(nm, 10)
    print(f"{nm:32s} last pass: " + " ".join(f"{d:6.1f}" for d in v) + f"  sum {sum(v):7.1f}")
PY
