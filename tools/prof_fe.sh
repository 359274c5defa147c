#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for v in A B; do
  OUT=gpurun_out/prof_fe_$v; rm -rf $OUT; mkdir -p $OUT
  export SYLOW_HIP_LIB=$GRAFT_REPO_ROOT/tools/ab/lib_$v.so
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o p -- python3 tools/fe_only.py > $OUT/t.log 2>&1
  rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU --output-format csv -d $OUT/pmc -o p -- python3 tools/fe_only.py > $OUT/p.log 2>&1
  rocprofv3 --pmc SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_INSTS_FLAT SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc2 -o p -- python3 tools/fe_only.py > $OUT/p2.log 2>&1
  python3 - <<PY
import csv, collections, glob
agg=collections.defaultdict(float); cnt=collections.Counter()
for f in glob.glob('$OUT/pmc*/p_counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        if 'k_final_exp' in r['Kernel_Name']: agg[r['Counter_Name']]+=float(r['Counter_Value']); cnt[r['Counter_Name']]+=1
print('$v', {k: v/cnt[k] for k,v in agg.items()})
for r in csv.DictReader(open('$OUT/trace/p_kernel_stats.csv')):
    if 'final_exp' in r['Name']: print('$v', r['Name'][:20], r['AverageNs'])
PY
done
