set -e
for abl in ${ABLS:-18 22 26 30}; do
  echo "== ABL $abl"; AUGCL_X2_ABL=$abl python tools/x2_bench.py 512 tuning 2>&1 | tail -11
done
