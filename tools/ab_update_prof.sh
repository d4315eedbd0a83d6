#!/bin/bash
# kernel-trace stats of the graph-replayed SVEA update() in two source trees on ONE box (run ON the GPU box)
ROOT=$(pwd); cd /tmp && export TMPDIR=/tmp && cd "$ROOT"
for t in "$@"; do
  tag=$(echo $t | tr -c 'a-zA-Z0-9\n' '_')
  ( cd $t && rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/abu_$tag -- python tools/prof_update.py 40 f32 graph > /dev/null 2>&1 )
  echo "== $t"
  python - <<PY
import csv, glob
rows=[]
for f in glob.glob("gpurun_out/abu_$tag/**/*kernel_stats.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total GPU busy %.2f ms" % (tot / 1e6))
for r in rows[:8]:
    print(f'{r["Name"][:60]:60s} calls {r["Calls"]:>6s} avg {float(r["AverageNs"])/1e3:8.2f} us total {float(r["TotalDurationNs"])/1e6:8.2f} ms')
PY
  rm -rf gpurun_out/abu_$tag
done
