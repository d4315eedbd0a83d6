#!/bin/bash
# rocprofv3 passes over the storing chains (k_conv32_chain<true>) at B = 256: durations, HBM traffic, MFMA busy -- on the GPU box
set -e
ROOT=$(pwd); cd /tmp && export TMPDIR=/tmp && cd "$ROOT"
O=gpurun_out
rocprofv3 --kernel-trace --stats --output-format csv -d $O/keep_stats -- python tools/prof_keep.py 30 256 > $O/keep_stats.log 2>&1
python tools/prof_summary.py $O/keep_stats 30 "gradient-carrying conv stack forward + backward, f32, B=256 (tools/prof_keep.py 30 256)"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/keep_fetch -- python tools/prof_keep.py 12 256 > /dev/null 2>&1
python tools/pmc_summary.py $O/keep_fetch
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/keep_write -- python tools/prof_keep.py 12 256 > /dev/null 2>&1
python tools/pmc_summary.py $O/keep_write
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/keep_mfma -- python tools/prof_keep.py 12 256 > /dev/null 2>&1
python tools/pmc_summary.py $O/keep_mfma
echo "[prof_keep] done"
