#!/bin/bash
# usage: prof_pairing.sh <tag>   -- rocprofv3 kernel-trace stats + PMC passes for the bench's dominant kernel (plk::k_pairing).
# One --kernel-trace --stats run, then one --pmc run per counter group (no trace domain is ever combined with --pmc).
# Writes gpurun_out/prof_<tag>/{kernel_stats.csv, pmc_k_pairing.json, pmc_current.json, bench_line.json}; copy them to profiles/<tag>/
# and pmc_current.json to profiles/ (bench.py reads it; it carries the hash of the kernel's sources).
TAG=${1:-r02_pairing}
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/prof_$TAG
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o p -- python3 bench.py --steps 3 --warmup 1 --no-cpu --no-aux > $OUT/bench_trace.log 2>&1
grep '^{' $OUT/bench_trace.log > $OUT/bench_line.json
cp $OUT/trace/p_kernel_stats.csv $OUT/kernel_stats.csv 2>/dev/null
run() { name=$1; shift; rocprofv3 --pmc "$@" --output-format csv -d $OUT/$name -o p -- python3 bench.py --steps 1 --warmup 0 --no-cpu --no-aux > $OUT/$name.log 2>&1; }
run sq SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE
run mix SQ_INSTS_VALU SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_LDS SQ_INSTS_SMEM GRBM_GUI_ACTIVE
run stall SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_FLAT
run fetch FETCH_SIZE
run write WRITE_SIZE
run tcc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum
python3 - <<PY
import csv, collections, glob, json, sys
sys.path.insert(0, '.')
import bench
out = {}
for f in sorted(glob.glob('$OUT/*/p_counter_collection.csv')):
    agg = collections.defaultdict(float)
    for r in csv.DictReader(open(f)):
        if 'k_pairing' in r['Kernel_Name']: agg[r['Counter_Name']] += float(r['Counter_Value'])
    name = f.split('/')[-2]
    for k, v in agg.items():
        out[k if k not in out else k + '@' + name] = v
kern_ms = None
for r in csv.DictReader(open('$OUT/kernel_stats.csv')):
    if 'k_pairing' in r['Name']: kern_ms = float(r['AverageNs']) / 1e6
out['kernel_ms_stats'] = kern_ms
out['note'] = ('plk::k_pairing (lane pairs), one launch, n=2^20 (32768 waves). FETCH_SIZE/WRITE_SIZE in KiB as reported by rocprofv3 '
               '(gfx950: FETCH_SIZE under-reports wide streaming reads 2x, MI355X_MICROARCH.md). INT64 = v_mad_[iu]64_[iu]32, 64-bit shifts/adds '
               '(quarter rate: 4 issue cycles); INT32 = every other integer VALU op incl. v_mul_lo_u32 (calibrated with tools/ubench/issue_rate under --pmc)')
json.dump(out, open('$OUT/pmc_k_pairing.json', 'w'), indent=1)
n = 1 << 20
valu, i64 = out['SQ_INSTS_VALU'], out['SQ_INSTS_VALU_INT64']
# clock of the PMC run that counted the mix: GRBM_GUI_ACTIVE is summed over the 8 XCDs; wall time of that run = cycles / clock is not
# known, so the clock is taken from the un-profiled kernel time of the stats run (profiled passes clock slightly lower: conservative)
clock_ghz = out['GRBM_GUI_ACTIVE'] / 8 / (kern_ms * 1e-3) / 1e9
cur = {"source": "profiles/$TAG", "n": n, "kernel_source_hash": bench.kernel_source_hash(),
       "valu_instr_per_pairing": valu / n, "valu_int64_per_pairing": i64 / n, "valu_int32_per_pairing": out['SQ_INSTS_VALU_INT32'] / n,
       "hbm_bytes_per_pairing": (2 * out["FETCH_SIZE"] + out["WRITE_SIZE"]) * 1024 / n,
       "clock_ghz": clock_ghz, "kernel_ms": kern_ms,
       # issue cycles per SIMD: quarter-rate class 4 cycles, everything else 2 (MI355X_MICROARCH.md: a wave64 VALU op issues over 2 cycles)
       "valu_cycles_ideal_per_pairing": (4 * i64 + 2 * (valu - i64)) / n,
       # the same with the issue rates this chip measured for pure streams at 2 waves/SIMD (profiles/r01_issue_rate_ubench.txt)
       "valu_cycles_ubench_per_pairing": (4.19 * i64 + 2.31 * (valu - i64)) / n,
       "mix": {"int64_class_frac": i64 / valu, "other_frac": 1 - i64 / valu},
       "issue_note": "frac = sum over instruction classes of (SQ_INSTS_VALU_INT64 x 4 + other VALU x 2 issue cycles) / (kernel time x clock x 1024 SIMDs); "
                     "classes from rocprofv3 PMC on this kernel, clock = GRBM_GUI_ACTIVE / 8 / kernel time; v_mul_lo_u32 (4 cycles, ~2 % of the stream) is "
                     "counted in the 2-cycle class, so the fraction is a slight under-estimate"}
json.dump(cur, open('$OUT/pmc_current.json', 'w'), indent=1)
print(json.dumps(cur))
PY
head -4 $OUT/kernel_stats.csv
cut -c1-300 $OUT/bench_line.json
