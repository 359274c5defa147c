#!/bin/bash
# usage: c5_pmc_hunt2.sh [processes] -- the profile driver itself, restricted to the configuration that failed, under the counter set of the pass that failed
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
N=${1:-30}; mkdir -p gpurun_out/hunt
for i in $(seq 1 $N); do
  rm -rf gpurun_out/hunt/q; mkdir -p gpurun_out/hunt
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_SALU SQ_INSTS_LDS GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/hunt/q -o p -- python3 tools/prof_configs.py --only "${ONLY:-C5_ecpairing_bytes_2^16_k2}" > gpurun_out/hunt/q.log 2>&1
  echo "pass $i: rc=$? $(grep -c WRONG gpurun_out/hunt/q.log) wrong $(grep -c Traceback gpurun_out/hunt/q.log) tracebacks"; grep -A8 Traceback gpurun_out/hunt/q.log | tail -9
  grep -A12 "WRONG" gpurun_out/hunt/q.log
done
rm -rf gpurun_out/hunt/q
