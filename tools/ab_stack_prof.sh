#!/bin/bash
# kernel trace of tools/ab_stack.py in two trees (run ON the GPU box): per-kernel average durations side by side
ROOT=$(pwd); cd /tmp && export TMPDIR=/tmp && cd "$ROOT"
for t in "$@"; do
  tag=$(echo $t | tr -c 'a-zA-Z0-9\n' '_')
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/abp_$tag -- python tools/ab_stack.py $t > /dev/null 2>&1
  echo "== $t"
  python - <<PY
import csv, glob
rows=[]
for f in glob.glob("gpurun_out/abp_$tag/**/*kernel_stats.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
for r in rows[:9]:
    print(f'{r["Name"][:70]:70s} calls {r["Calls"]:>6s} avg {float(r["AverageNs"])/1e3:8.2f} us total {float(r["TotalDurationNs"])/1e6:8.2f} ms')
PY
  rm -rf gpurun_out/abp_$tag
done
