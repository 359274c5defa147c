#!/bin/bash
# usage: ab_raw.sh libA.so libB.so -- alternate two builds on ONE box (time), then SQ_INSTS_VALU of k_pairing for each
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for r in 1 2 3; do for L in "$@"; do python3 tools/dbg/ab_raw.py $L 3 2>&1 | grep -E 'pairing|verify'; done; done
for L in "$@"; do
  O=gpurun_out/ab_raw/$(basename $L .so); rm -rf $O; mkdir -p $O
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VALU_INT64 GRBM_GUI_ACTIVE --output-format csv -d $O -o p -- python3 tools/dbg/ab_raw.py $L 1 > $O/log 2>&1
  python3 - $O $L <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(list)
for r in csv.DictReader(open(glob.glob(sys.argv[1] + "/*counter_collection.csv")[0])):
    if "k_pairing" in r["Kernel_Name"]: acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
n = 1 << 20
print(sys.argv[2], {k: sum(v) / len(v) / n for k, v in acc.items()})
PY
done
