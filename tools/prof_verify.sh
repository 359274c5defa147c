#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/prof_verify; rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o p -- python3 tools/prof_verify.py > $OUT/trace.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_LDS --output-format csv -d $OUT/sq -o p -- python3 tools/prof_verify.py > $OUT/sq.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_INT32 GRBM_GUI_ACTIVE --output-format csv -d $OUT/mix -o p -- python3 tools/prof_verify.py > $OUT/mix.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -o p -- python3 tools/prof_verify.py > $OUT/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -o p -- python3 tools/prof_verify.py > $OUT/write.log 2>&1
python3 - <<PY
import csv, collections, json, glob
agg=collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob('$OUT/*/p_counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name'].split('(')[0]
        if 'bls_verify' in k: agg[k][r['Counter_Name']]+=float(r['Counter_Value'])
ms={}
for r in csv.DictReader(open('$OUT/trace/p_kernel_stats.csv')):
    if 'bls_verify' in r['Name']: ms[r['Name'].split('(')[0]]=float(r['AverageNs'])/1e6
issue={}
for k,v in agg.items():
    if k in ms and 'SQ_INSTS_VALU_INT64' in v and 'GRBM_GUI_ACTIVE' in v:
        n=1<<20
        i64=v['SQ_INSTS_VALU_INT64']; oth=v['SQ_INSTS_VALU']-i64
        clock=v['GRBM_GUI_ACTIVE']/8/(ms[k]*1e-3)            # GRBM_GUI_ACTIVE is summed over the 8 XCDs
        avail=ms[k]*1e-3*clock*1024
        issue[k]={"clock_ghz":clock/1e9,"valu_per_verify":v['SQ_INSTS_VALU']/n,"int64_class_frac":i64/v['SQ_INSTS_VALU'],
                  "issue_frac_ideal":(4*i64+2*oth)/avail,"issue_frac_measured_rates":(4.19*i64+2.31*oth)/avail}
out={"note":"tools/prof_verify.sh: one launch each at n = 2^20; FETCH_SIZE/WRITE_SIZE in KiB; issue fractions as bench.py's issue_roofline (quarter-rate 64-bit class x 4 cycles + other VALU x 2, over kernel time x clock x 1024 SIMDs)", "kernel_ms":ms, "issue":issue, "pmc":{k:dict(v) for k,v in agg.items()}}
json.dump(out, open('$OUT/summary.json','w'), indent=1); print(json.dumps(out))
PY
