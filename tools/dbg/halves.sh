#!/bin/bash
# Issue fraction of the two halves of a pairing on their own: k_miller_loop and k_final_exp at 2^20 (tools/fe_only.py), one counter pass.
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
OUT=gpurun_out/halves; rm -rf $OUT; mkdir -p $OUT
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VALU_INT64 GRBM_GUI_ACTIVE SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d $OUT/pmc -o p -- python3 tools/fe_only.py > $OUT/p.log 2>&1
python3 - <<'PY'
import csv, collections, glob
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(glob.glob("gpurun_out/halves/pmc/*counter_collection.csv")[0])):
    for k in ("k_miller_loop", "k_final_exp"):
        if k in r["Kernel_Name"]: acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
n = 1 << 20
for k, d in acc.items():
    m = {c: sum(v) / len(v) / n for c, v in d.items()}
    ideal = m["SQ_INSTS_VALU_INT64"] * 4 + (m["SQ_INSTS_VALU"] - m["SQ_INSTS_VALU_INT64"]) * 2
    simd_cycles = m["GRBM_GUI_ACTIVE"] / 8 * 1024
    print(k, {c: round(v, 1) for c, v in m.items()}, "issue frac %.3f" % (ideal / simd_cycles), "int64 share %.2f" % (m["SQ_INSTS_VALU_INT64"] / m["SQ_INSTS_VALU"]))
PY
