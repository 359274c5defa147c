#!/bin/bash
# usage: ab_icache.sh libA.so libB.so -- instruction-cache and instruction-fetch-stall counters of k_pairing (per pairing) for two builds
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for L in "$@"; do
  for set in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQC_TC_INST_REQ GRBM_GUI_ACTIVE" "SQ_IFETCH SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU"; do
    O=gpurun_out/ab_icache/$(basename $L .so); rm -rf $O; mkdir -p $O
    rocprofv3 --pmc $set --output-format csv -d $O -o p -- python3 tools/dbg/ab_raw.py $L 1 > $O/log 2>&1
    python3 - $O $L <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(glob.glob(sys.argv[1] + "/*counter_collection.csv")[0])):
    for k in ("k_pairing", "k_bls_verify_fused"):
        if k in r["Kernel_Name"]: acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
n = 1 << 20
for k, d in acc.items():
    print(sys.argv[2], k, {c: round(sum(v) / len(v) / n, 3) for c, v in d.items()})
PY
  done
done
