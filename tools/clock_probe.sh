#!/bin/bash
# Sample the GPU's shader clock / power while ONE ebench variant runs for a few seconds:  bash tools/clock_probe.sh VARIANT_INDEX [B H]
V=$1; B=${2:-512}; H=${3:-37}
./tools/ebench $B $H $V 4 &
PID=$!
sleep 1.2
for i in 1 2 3 4 5; do
    rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power|Average" | head -4
    sleep 0.4
done
wait $PID
