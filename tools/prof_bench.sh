#!/bin/bash
# Per-kernel time of EVERY launch of the default bench.py run (timed loop + aux configs): rocprofv3 kernel trace + stats.
# usage: tools/prof_bench.sh <tag>   -> gpurun_out/prof_<tag>/kernel_stats.csv
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/prof_$1; rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o p -- python3 bench.py --no-cpu > $OUT/bench_line.json 2> $OUT/trace.log
cp $OUT/trace/p_kernel_stats.csv $OUT/kernel_stats.csv
python3 - <<PY
import csv
for r in csv.DictReader(open('$OUT/kernel_stats.csv')):
    print("%-64s calls %5s avg %10.3f ms total %9.1f ms" % (r['Name'].split('(')[0][:64], r['Calls'], float(r['AverageNs'])/1e6, float(r['TotalDurationNs'])/1e6))
PY
