#!/bin/bash
# PMC passes over tools/x3_bench.py (per-layer x3 vs f32 kernels): MFMA-busy / clocks / waits / LDS.  Run ON the GPU box.
set -e
B=${1:-512}
ROOT=$(pwd)
cd /tmp && export TMPDIR=/tmp && cd "$ROOT"
O=gpurun_out
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $O/x3_pmc_a -- python tools/x3_bench.py $B > /dev/null 2>&1
python tools/pmc_summary.py $O/x3_pmc_a
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $O/x3_pmc_b -- python tools/x3_bench.py $B > /dev/null 2>&1
python tools/pmc_summary.py $O/x3_pmc_b
rocprofv3 --kernel-trace --stats --output-format csv -d $O/x3_trace -- python tools/x3_bench.py $B > /dev/null 2>&1
python tools/prof_summary.py $O/x3_trace 1 "tools/x3_bench.py $B"
