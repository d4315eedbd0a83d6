#!/bin/bash
# PMC passes over the conv stack in precision "f32" (tools/ab_stack.py: dual / no-grad / grad forward and forward + backward at
# B = 256 / 512): where the f16-pair kernels' cycles go.  Run ON the GPU box; writes gpurun_out/x2_pmc_{a,b}.json.
set -e
ROOT=$(pwd)
cd /tmp && export TMPDIR=/tmp && cd "$ROOT"
O=gpurun_out
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $O/x2_pmc_a -- python tools/ab_stack.py . > /dev/null 2>&1
python tools/pmc_summary.py $O/x2_pmc_a
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $O/x2_pmc_b -- python tools/ab_stack.py . > /dev/null 2>&1
python tools/pmc_summary.py $O/x2_pmc_b
python - <<PY
import json
a = json.load(open("$O/x2_pmc_a.json")); b = json.load(open("$O/x2_pmc_b.json"))
def merge(d):
    out = {}
    for k, v in d.items():
        n = k.partition(" [grid ")[0]
        if "x2" not in n and "first" not in n: continue
        e = out.setdefault(n, {})
        for c, x in v.items():
            t = e.setdefault(c, [0.0, 0]); t[0] += x["avg"] * x["n"]; t[1] += x["n"]
    return {n: {c: t[0] / t[1] for c, t in e.items()} for n, e in out.items()}
A, Bm = merge(a), merge(b)
print("%-48s %9s %9s %9s %9s | %9s %9s %9s %9s" % ("kernel (all launch sizes pooled)", "mfma/CUbusy", "wait/wave", "waitinst", "active", "VALU/wave", "LDS insts", "bankconf", "LDS idx"))
for n in A:
    x, y = A[n], Bm.get(n, {})
    print("%-48s %9.3f %9.3f %9.3f %9.3f | %9.0f %9.0f %9.3f %9.3f" % (
        n[:48], x["SQ_VALU_MFMA_BUSY_CYCLES"] / (4 * x["SQ_BUSY_CU_CYCLES"]), x["SQ_WAIT_ANY"] / x["SQ_WAVE_CYCLES"],
        x["SQ_WAIT_INST_ANY"] / x["SQ_WAVE_CYCLES"], x["SQ_ACTIVE_INST_ANY"] / x["SQ_WAVE_CYCLES"],
        y.get("SQ_INSTS_VALU", 0), y.get("SQ_INSTS_LDS", 0),
        y.get("SQ_LDS_BANK_CONFLICT", 0) / max(y.get("SQ_LDS_IDX_ACTIVE", 1), 1), y.get("SQ_LDS_IDX_ACTIVE", 0) / max(x["SQ_BUSY_CU_CYCLES"], 1)))
PY
