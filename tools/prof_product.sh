#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/prof_product; rm -rf $OUT; mkdir -p $OUT
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_SALU --output-format csv -d $OUT/sq -o p -- python3 tools/prof_product.py > $OUT/sq.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o p -- python3 tools/prof_product.py > $OUT/trace.log 2>&1
python3 - <<PY
import csv, collections
agg=collections.defaultdict(lambda: collections.defaultdict(float))
for r in csv.DictReader(open('$OUT/sq/p_counter_collection.csv')):
    agg[r['Kernel_Name'].split('(')[0]][r['Counter_Name']]+=float(r['Counter_Value'])
for k,v in agg.items():
    if 'miller' in k or 'multi_pairing' in k or 'tree' in k: print(k, dict(v))
PY
grep -E "miller|multi_pairing|tree_level|final_exp" $OUT/trace/p_kernel_stats.csv | cut -d, -f1-4 | cut -c1-160
