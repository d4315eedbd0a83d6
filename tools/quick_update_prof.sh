set -e
ROOT=$(pwd); cd /tmp && export TMPDIR=/tmp && cd "$ROOT"
O=gpurun_out
rocprofv3 --kernel-trace --stats --output-format csv -d $O/q_update_f32 -- python tools/prof_update.py 40 f32 graph > $O/q_update_f32.log 2>&1
python tools/prof_summary.py $O/q_update_f32 48 "SVEA update() B=256 f32 graph (quick look)"
