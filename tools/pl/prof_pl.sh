#!/bin/bash
# PMC passes over the dev harness's pairing kernels (lane-pair vs single-lane), n = 2^20
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/prof_pl
rm -rf $OUT; mkdir -p $OUT
run() { name=$1; shift; rocprofv3 --pmc "$@" --output-format csv -d $OUT/$name -o p -- python3 tools/pl/run_pl.py bench > $OUT/$name.log 2>&1; }
run sq SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE



python3 - <<PY
import csv, collections, glob, json
out=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.defaultdict(lambda: collections.defaultdict(int))
for f in sorted(glob.glob('$OUT/*/p_counter_collection.csv')):
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name'].split('(')[0]
        out[k][r['Counter_Name']]+=float(r['Counter_Value']); cnt[k][r['Counter_Name']]+=1
res={}
for k in out:
    res[k]={c: out[k][c]/max(1,cnt[k][c]/ (8 if False else 1)) for c in out[k]}
    res[k]['_dispatches']=max(cnt[k].values())
json.dump(res, open('$OUT/pmc.json','w'), indent=1)
for k in res: print(k, json.dumps(res[k]))
PY
tail -3 $OUT/sq.log
