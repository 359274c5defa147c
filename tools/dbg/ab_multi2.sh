#!/bin/bash
# table route with / without the one-operation-ahead line prefetch, then a kernel trace of the table route
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for rep in 1 2; do
  for pf in 0 1; do
    echo "== tables, SYLOW_HIP_MULTI_PREFETCH=$pf rep $rep"
    SYLOW_HIP_MULTI_TABLES=1 SYLOW_HIP_MULTI_PREFETCH=$pf LOG2N=${LOG2N:-18} python3 tools/dbg/time_multi.py 2>&1 | grep -v "^W\|^E\|amdgpu.ids"
  done
done
export SYLOW_HIP_MULTI_TABLES=1 LOG2N=18
rm -rf gpurun_out/kt_multi; rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kt_multi -o t -- python3 tools/dbg/time_multi.py > /dev/null 2>&1
python3 - <<PY
import csv, glob
for f in glob.glob('gpurun_out/kt_multi/**/t_kernel_stats.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if any(k in r['Name'] for k in ('pair_lines', 'glued_from', 'multi_pairing', 'k_pairing')):
            print(r['Name'][:60], 'calls', r['Calls'], 'avg_ms', round(float(r['AverageNs'])/1e6, 3), 'min', round(float(r['MinNs'])/1e6,3), 'max', round(float(r['MaxNs'])/1e6,3))
PY
