#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
OUT=gpurun_out/pmc_phases; rm -rf $OUT; mkdir -p $OUT
cat > /tmp/phases.py <<PY
import os, sys
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import numpy as np, sylow_amd
from bench import make_points
eng = sylow_amd.Engine(0)
n = 1 << 20
p, q, ka, kb = make_points(eng, n, 5)
f = eng.empty((48, n)); gt = eng.empty((48, n))
for _ in range(2):
    eng._call("sylow_hip_miller_loop_batch", p.ptr, q.ptr, f.ptr, n)
    eng._call("sylow_hip_final_exp_batch", f.ptr, gt.ptr, n)
    eng._call("sylow_hip_pairing_batch", p.ptr, None, q.ptr, None, gt.ptr, n)
eng.sync()
PY
rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE --output-format csv -d $OUT/a -o p -- python3 /tmp/phases.py > $OUT/a.log 2>&1
rocprofv3 --pmc SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INST_LEVEL_VMEM SQ_INSTS_FLAT TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/b -o p -- python3 /tmp/phases.py > $OUT/b.log 2>&1
python3 - <<PY
import csv, collections, glob
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('$OUT/*/p_counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name'].split('(')[0]
        if k.startswith('plk::k_'): agg[k][r['Counter_Name']].append(float(r['Counter_Value']))
for k,v in sorted(agg.items()):
    print(k, {c: round(sum(x)/len(x)/2**20,1) for c,x in sorted(v.items())})
PY
