#!/bin/bash
# forward / backward-data time of every layer size under forced strip heights (tuning build reads AUGCL_X2_RT); run ON the GPU box
B=${1:-512}
for rt in 2 3 4 5 6 7 8 9 10 11 12 14; do
  echo "== RT $rt"; AUGCL_X2_RT=$rt python tools/x2_bench.py $B tuning 2>&1 | grep -E "^[0-9]+ |^sum" | awk '{print $1, $4, $7}' | tr '\n' ';'; echo
done
echo "== default"; python tools/x2_bench.py $B tuning 2>&1 | grep -E "^[0-9]+ |^sum" | awk '{print $1, $4, $7}' | tr '\n' ';'; echo
