#!/bin/bash
# shader clock / power while the graph-replayed SVEA update() runs for a few seconds:  bash tools/update_power_probe.sh [precision]
P=${1:-f32}
python -c "
import sys, torch; sys.path.insert(0, '.')
from augcl_amd import bench_update
r = bench_update.run(torch.device('cuda:0'), None, 1, 0, steps=2500, warmup=10, B=256, precision='$P', graph=True)
print('ms_per_update', r['ms_per_update'])
" &
PID=$!
sleep 4
for i in 1 2 3 4 5 6; do
    rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power|Average" | head -3
    sleep 0.4
done
wait $PID
