#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
SYLOW_HIP_MULTI_TABLES=1 python -m pytest tests/test_gpu_multi_pairing.py tests/test_gpu_evm.py tests/test_gpu_precomputed.py -m gpu -q -x 2>&1 | tail -3
python -m pytest tests/test_gpu_multi_pairing.py tests/test_gpu_evm.py tests/test_gpu_aggregate.py -m gpu -q -x 2>&1 | tail -3
for rep in 1 2; do
  echo "== in-register rep $rep"; SYLOW_HIP_MULTI_TABLES=0 LOG2N=18 python3 tools/dbg/time_multi.py 2>&1 | grep "^multi\|^pairing"
  echo "== tables rep $rep";      SYLOW_HIP_MULTI_TABLES=1 LOG2N=18 python3 tools/dbg/time_multi.py 2>&1 | grep "^multi\|^pairing"
done
echo "== default routing, 2^20 pairs"; LOG2N=20 python3 tools/dbg/time_multi.py 2>&1 | grep "^multi\|^pairing"
tools/dbg/kt_multi.sh
