#!/bin/bash
cd $GRAFT_REPO_ROOT
SYLOW_HIP_MULTI_TABLES=1 python -m pytest tests/test_gpu_aggregate.py tests/test_gpu_multi_pairing.py -m gpu -q -x 2>&1 | grep -E "passed|failed"
python -m pytest tests/test_gpu_aggregate.py -m gpu -q -x 2>&1 | grep -E "passed|failed"
for rep in 1 2; do
  echo "== in-register rep $rep"; SYLOW_HIP_MULTI_TABLES=0 python3 tools/dbg/time_agg.py 2>&1 | grep "^pairing_product\|^aggregate"
  echo "== default rep $rep";     python3 tools/dbg/time_agg.py 2>&1 | grep "^pairing_product\|^aggregate"
done
