#!/bin/bash
# usage: c5_pmc_hunt.sh [processes] [rounds] -- tools/dbg/c5_pmc_hunt.py once plain, then `processes` times under the counter set of the pass that failed
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
N=${1:-4}; R=${2:-150}
mkdir -p gpurun_out/hunt
echo "== plain"; python3 tools/dbg/c5_pmc_hunt.py $R 2>&1 | tail -12
for i in $(seq 1 $N); do
  echo "== pmc mix $i"
  rm -rf gpurun_out/hunt/p$i
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_SALU SQ_INSTS_LDS GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/hunt/p$i -o p -- python3 tools/dbg/c5_pmc_hunt.py $R 2>&1 | grep -v "^[WEI]2026" | tail -12
  rm -rf gpurun_out/hunt/p$i
done
