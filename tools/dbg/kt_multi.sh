#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
export SYLOW_HIP_MULTI_TABLES=1 LOG2N=18
for K in 4 2; do
  export K
  rm -rf gpurun_out/kt_multi_$K; rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kt_multi_$K -o t -- python3 tools/dbg/time_multi_k.py 2>&1 | grep "^multi\|^raw"
  python3 - <<PY
import csv, glob
for f in glob.glob('gpurun_out/kt_multi_$K/**/t_kernel_stats.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if any(k in r['Name'] for k in ('pair_lines', 'glued_from', 'multi_pairing', 'k_pairing')):
            print(' ', r['Name'][:60], 'calls', r['Calls'], 'avg_ms', round(float(r['AverageNs'])/1e6, 3), 'min', round(float(r['MinNs'])/1e6,3), 'max', round(float(r['MaxNs'])/1e6,3))
PY
done
