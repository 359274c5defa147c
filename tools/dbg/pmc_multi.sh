#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
OUT=gpurun_out/pmc_multi; rm -rf $OUT; mkdir -p $OUT
export LOG2N=20
rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_INSTS_SALU GRBM_GUI_ACTIVE --output-format csv -d $OUT/a -o p -- python3 tools/dbg/time_multi.py > $OUT/a.log 2>&1
rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_BRANCH --output-format csv -d $OUT/b -o p -- python3 tools/dbg/time_multi.py > $OUT/b.log 2>&1
python3 - <<PY
import csv, collections, glob
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('$OUT/*/p_counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name'].split('(')[0]
        if 'pairing' in k: agg[(k, r['Grid_Size'])][r['Counter_Name']].append(float(r['Counter_Value']))
for (k,g),v in sorted(agg.items()):
    jobs=int(g)//2
    print(k[:40], 'jobs', jobs, {c: round(sum(x)/len(x)/jobs,1) for c,x in sorted(v.items())})
PY
