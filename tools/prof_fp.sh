#!/bin/bash
# usage: prof_fp.sh <tag>  -- the HBM-bound kernels of the path (k_fp_binop<OP,0>: 96 algorithmic bytes per element): kernel-trace stats,
# then FETCH_SIZE and WRITE_SIZE in separate --pmc passes.  Output: gpurun_out/prof_<tag>/{kernel_stats.csv, summary.json}
TAG=${1:-r02_fp_batch}
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/prof_$TAG
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o p -- python3 tools/prof_fp.py > $OUT/trace.log 2>&1
cp $OUT/trace/p_kernel_stats.csv $OUT/kernel_stats.csv
cp $OUT/trace/p_kernel_trace.csv $OUT/kernel_trace.csv 2>/dev/null
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -o p -- python3 tools/prof_fp.py > $OUT/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -o p -- python3 tools/prof_fp.py > $OUT/write.log 2>&1
python3 - <<PY
import csv, collections, json
# per (kernel, grid size) : durations from the kernel trace, counters from the PMC passes
dur = collections.defaultdict(list)
for r in csv.DictReader(open('$OUT/kernel_trace.csv')):
    if 'k_fp_binop' in r['Kernel_Name']:
        dur[(r['Kernel_Name'], int(r.get('Grid_Size') or r['Grid_Size_X']))].append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
cnt = collections.defaultdict(lambda: collections.defaultdict(list))
for which in ('fetch', 'write'):
    for r in csv.DictReader(open('$OUT/%s/p_counter_collection.csv' % which)):
        if 'k_fp_binop' in r['Kernel_Name']:
            cnt[(r['Kernel_Name'], int(r['Grid_Size']))][r['Counter_Name']].append(float(r['Counter_Value']))
out = {}
for key, ds in sorted(dur.items()):
    name, grid = key
    n = grid * 2                      # two elements per lane
    ds = sorted(ds)[: max(1, len(ds) - 1)]          # drop the slowest (first, cold) launch
    t = sum(ds) / len(ds) * 1e-9
    c = cnt[key]
    # one counter row per dispatch (KiB as reported)
    fetch = sum(c['FETCH_SIZE']) / len(c['FETCH_SIZE']) * 1024 if c['FETCH_SIZE'] else None
    write = sum(c['WRITE_SIZE']) / len(c['WRITE_SIZE']) * 1024 if c['WRITE_SIZE'] else None
    out[f"{name.split('(')[0]} n=2^{n.bit_length() - 1}"] = {
        "avg_launch_us": t * 1e6, "elements": n, "algorithmic_bytes": 96 * n, "algorithmic_GBps": 96 * n / t / 1e9, "frac_of_8TBps": 96 * n / t / 8e12,
        "FETCH_SIZE_bytes_as_reported": fetch, "WRITE_SIZE_bytes": write,
        "hbm_bytes_corrected": (2 * fetch + write) if fetch is not None and write is not None else None,
        "traffic_over_algorithmic": ((2 * fetch + write) / (96 * n)) if fetch is not None and write is not None else None}
out["note"] = ("k_fp_binop<OP,FR>: OP 0 add, 1 sub, 2 mul; 2 elements per lane, 16-byte accesses, 64 B read + 32 B written per element. FETCH_SIZE doubled per the "
               "gfx950 correction for wide streaming reads (MI355X_MICROARCH.md §HBM); 2^20-element launches (96 MB) mostly hit the 256 MiB Infinity Cache")
json.dump(out, open('$OUT/summary.json', 'w'), indent=1)
print(json.dumps(out, indent=1))
PY
