#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/pmc_hash; rm -rf $OUT; mkdir -p $OUT
rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d $OUT/a -o p -- python3 tools/dbg/time_hash.py > $OUT/a.log 2>&1
python3 - <<PY
import csv, collections, glob
agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.defaultdict(collections.Counter)
for f in glob.glob('$OUT/a/**/p_counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name'].split('(')[0]
        agg[k][r['Counter_Name']]+=float(r['Counter_Value']); cnt[k][r['Counter_Name']]+=1
for k in agg:
    d={c: v/cnt[k][c] for c,v in agg[k].items()}
    w=d.get('SQ_WAVES',1)
    print(k[:30], {c: round(v/w) for c,v in d.items() if c!='SQ_WAVES'}, 'waves', w)
PY
