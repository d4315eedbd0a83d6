#!/bin/bash
# The rocprofv3 passes behind profiles/ -- run ON the GPU box from the repo root:  bash tools/profile_round.sh r4
# then, back in the container:  python tools/summarize_profiles.py r4   (copies / condenses gpurun_out/r3_* into profiles/).
# One counter group per pass (MI355X_MICROARCH.md); --pmc is never combined with the hip / sys traces; every raw
# trace directory is condensed on the box (tools/prof_summary.py, tools/pmc_summary.py) and deleted.
# The hash of the source tree the binary was built from goes next to the results: bench.py only quotes counters whose
# hash matches the tree it runs on.
set -e
R=${1:-r4}
ROOT=$(pwd)
cd /tmp && export TMPDIR=/tmp && cd "$ROOT"
O=gpurun_out
python -c "import sys; sys.path.insert(0, '.'); from augcl_amd import _lib; print(_lib.source_hash())" > $O/${R}_source_hash.txt
BENCH="python bench.py --steps 2000 --warmup 200 --no-sac --no-cpu-baseline --min-timed-s 0"

# 1. per-kernel durations of the augmentation / replay bench (headline kernel k_conv<...> and the extras)
rocprofv3 --kernel-trace --stats --output-format csv -d $O/${R}_bench -- $BENCH > $O/${R}_bench.out 2>&1
python tools/prof_summary.py $O/${R}_bench 1 "bench.py --steps 2000 --warmup 200 --no-sac (B=256 augmentation / replay kernels)"
echo "[profile_round] bench stats done"

# 2. HBM traffic of the same kernels: FETCH_SIZE and WRITE_SIZE in separate passes
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/${R}_fetch -- $BENCH > /dev/null 2>&1
python tools/pmc_summary.py $O/${R}_fetch
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/${R}_write -- $BENCH > /dev/null 2>&1
python tools/pmc_summary.py $O/${R}_write
echo "[profile_round] traffic done"

# 3. SVEA update(): per-kernel durations (hipGraph replay), MFMA-busy counters and HBM traffic of the encoder kernels,
#    f32 parity path (f16-pair kernels) and bf16 mode; durations only for the split-bf16 and native f32 MFMA forms of the parity path
for P in f32 bf16 f32_x3 f32_mfma; do
    rocprofv3 --kernel-trace --stats --output-format csv -d $O/${R}_update_$P -- python tools/prof_update.py 40 $P graph > $O/${R}_update_$P.log 2>&1
    python tools/prof_summary.py $O/${R}_update_$P 48 "SVEA update() B=256, $P, hipGraph replay (tools/prof_update.py 40 $P graph: 8 warm-up + 40 timed)"
    echo "[profile_round] update $P stats done"
    [ "$P" = "f32_mfma" ] && continue
    [ "$P" = "f32_x3" ] && continue
    rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/${R}_mfma_$P -- python tools/prof_update.py 12 $P > /dev/null 2>&1
    python tools/pmc_summary.py $O/${R}_mfma_$P
    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/${R}_ufetch_$P -- python tools/prof_update.py 12 $P > /dev/null 2>&1
    python tools/pmc_summary.py $O/${R}_ufetch_$P
    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/${R}_uwrite_$P -- python tools/prof_update.py 12 $P > /dev/null 2>&1
    python tools/pmc_summary.py $O/${R}_uwrite_$P
    echo "[profile_round] update $P counters done"
done

# 4. the launch sequence of one graph-replayed update (launch count, per-kernel durations in order)
rocprofv3 --kernel-trace --output-format csv -d $O/${R}_seq_f32 -- python tools/prof_update.py 6 f32 graph > /dev/null 2>&1
python tools/prof_sequence.py $O/${R}_seq_f32 420
echo "[profile_round] all done"
