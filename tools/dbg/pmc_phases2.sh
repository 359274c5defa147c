#!/bin/bash
# Parked / issuing split of the two phases of a pairing as standalone kernels (k_miller_loop, k_final_exp at 2^19) -- where does SQ_WAIT_ANY come from?
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/pmc_phases2; rm -rf $O; mkdir -p $O
cat > $O/drv.py <<'PY'
import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, sylow_amd
from bench import make_points, SEED
eng = sylow_amd.Engine(0)
n = 1 << 19
p, q, ka, kb = make_points(eng, n, SEED + 3)
f, g = eng.empty((48, n)), eng.empty((48, n))
for _ in range(2):
    eng._call("sylow_hip_miller_loop_batch", p.ptr, q.ptr, f.ptr, n)
    eng._call("sylow_hip_final_exp_batch", f.ptr, g.ptr, n)
eng.sync()
PY
for grp in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VALU_INT64 GRBM_GUI_ACTIVE" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_FLAT SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVES"; do
  name=$(echo $grp | cut -d' ' -f1)
  rocprofv3 --pmc $grp --output-format csv -d $O/$name -o p -- python3 $O/drv.py > $O/$name.log 2>&1
done
python3 - $O <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/**/p_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        if "k_miller_loop" in k or "k_final_exp" in k:
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
n = 1 << 19
for k, c in acc.items():
    v = {a: sum(b) / 2 for a, b in c.items()}      # two launches each; counters of one dispatch come as several rows (per XCD): sum / launches
    wc = v.get("SQ_WAVE_CYCLES", 1)
    print(k)
    print("   per pairing: VALU %.0f (INT64 %.0f)  scratch rd %.1f wr %.1f  flat %.1f  LDS %.1f  SALU %.0f" % (v.get("SQ_INSTS_VALU", 0) / n, v.get("SQ_INSTS_VALU_INT64", 0) / n, v.get("SQ_INSTS_VMEM_RD", 0) / n, v.get("SQ_INSTS_VMEM_WR", 0) / n, v.get("SQ_INSTS_FLAT", 0) / n, v.get("SQ_INSTS_LDS", 0) / n, v.get("SQ_INSTS_SALU", 0) / n))
    print("   wave cycles: issuing %.4f  issue-stalled %.4f  parked %.4f ;  issue frac (ideal) %.4f" % (v.get("SQ_ACTIVE_INST_ANY", 0) / wc, v.get("SQ_WAIT_INST_ANY", 0) / wc, v.get("SQ_WAIT_ANY", 0) / wc,
          (4 * v.get("SQ_INSTS_VALU_INT64", 0) + 2 * (v.get("SQ_INSTS_VALU", 0) - v.get("SQ_INSTS_VALU_INT64", 0))) / (v.get("GRBM_GUI_ACTIVE", 1) / 8 * 1024)))
PY
