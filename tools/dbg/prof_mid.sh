#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/prof_mid; rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --output-format csv -d $OUT/t -o p -- python3 tools/dbg/time_mid.py > $OUT/t.log 2>&1
python3 - <<PY
import csv, glob, collections
f=glob.glob('$OUT/t/**/p_kernel_trace.csv', recursive=True)[0]
agg=collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    n=r['Kernel_Name']
    if 'wide_batch' in n:
        agg[(n.split('(')[0][-28:], int(r['Grid_Size_X'])//64)].append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e6)
for k in sorted(agg): print(k, 'ms %.3f' % (sum(agg[k])/len(agg[k])))
PY
