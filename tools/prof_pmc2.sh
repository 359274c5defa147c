#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/prof_pmc2
rm -rf $OUT; mkdir -p $OUT
run() { name=$1; shift; rocprofv3 --pmc "$@" --output-format csv -d $OUT/$name -o p -- python3 bench.py --steps 1 --warmup 0 --no-cpu --no-aux --log2n 18 > $OUT/$name.log 2>&1; }
run stall SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT
run icache SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQ_INSTS_FLAT SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INST_LEVEL_VMEM
run tcc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum
run tcp TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum
python3 - <<'PY'
import csv, collections, glob
for f in sorted(glob.glob('gpurun_out/prof_pmc2/*/p_counter_collection.csv')):
    agg=collections.defaultdict(float)
    for r in csv.DictReader(open(f)):
        if 'k_pairing' in r['Kernel_Name']: agg[r['Counter_Name']]+=float(r['Counter_Value'])
    print(f.split('/')[2], dict(agg))
PY
grep -h "error\|Error\|invalid" $OUT/*.log | head
