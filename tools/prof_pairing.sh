#!/bin/bash
# usage: prof_pairing.sh <tag>   -- rocprofv3 kernel-trace stats + PMC passes for the bench's dominant kernel
TAG=${1:-r01_v2}
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/prof_$TAG
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o p -- python3 bench.py --steps 3 --warmup 1 --no-cpu > $OUT/bench_trace.log 2>&1
grep '^{' $OUT/bench_trace.log > $OUT/bench_line.json
run() { name=$1; shift; rocprofv3 --pmc "$@" --output-format csv -d $OUT/$name -o p -- python3 bench.py --steps 1 --warmup 0 --no-cpu --no-aux > $OUT/$name.log 2>&1; }
run sq SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE
run stall SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_FLAT
run fetch FETCH_SIZE
run write WRITE_SIZE
run tcc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum
python3 - <<PY
import csv, collections, glob, json
out={}
for f in sorted(glob.glob('$OUT/*/p_counter_collection.csv')):
    agg=collections.defaultdict(float)
    for r in csv.DictReader(open(f)):
        if 'k_pairing' in r['Kernel_Name']: agg[r['Counter_Name']]+=float(r['Counter_Value'])
    out.update(agg)
out['note']='plk::k_pairing (lane pairs), one launch, n=2^20 (32768 waves). FETCH_SIZE/WRITE_SIZE in KiB as reported by rocprofv3 (gfx950: FETCH_SIZE under-reports wide streaming reads 2x, MI355X_MICROARCH.md)'
json.dump(out, open('$OUT/pmc_k_pairing.json','w'), indent=1)
n = 1 << 20
json.dump({"source": "profiles/r01_pairing_$TAG".replace("r01_pairing_r01_", "r01_pairing_"), "n": n,
           "valu_instr_per_pairing": out["SQ_INSTS_VALU"] / n,
           "hbm_bytes_per_pairing": (2 * out["FETCH_SIZE"] + out["WRITE_SIZE"]) * 1024 / n}, open('$OUT/pmc_current.json', 'w'), indent=1)
print(json.dumps(out))
PY
cat $OUT/trace/p_kernel_stats.csv | head -4
cat $OUT/bench_line.json | cut -c1-200
