#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
OUT=gpurun_out/pmc_multi; rm -rf $OUT; mkdir -p $OUT
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VALU_INT64 SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES --output-format csv -d $OUT/a -o p -- python3 tools/dbg/time_multi.py > $OUT/a.log 2>&1
python3 - <<PY
import csv, collections
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open('$OUT/a/p_counter_collection.csv')):
    k=r['Kernel_Name'].split('(')[0]
    if 'pairing' in k: agg[(k, r['Grid_Size'])][r['Counter_Name']].append(float(r['Counter_Value']))
for (k,g),v in sorted(agg.items()):
    jobs=int(g)//2
    print(k[:40], 'jobs', jobs, {c: round(sum(x)/len(x)/jobs,1) for c,x in v.items()})
PY
