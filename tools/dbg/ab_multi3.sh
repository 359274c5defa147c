#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for rep in 1 2; do
  echo "== in-register rep $rep";        SYLOW_HIP_MULTI_TABLES=0 LOG2N=18 python3 tools/dbg/time_multi.py 2>&1 | grep "^multi\|^pairing"
  echo "== tables, no prefetch rep $rep"; SYLOW_HIP_MULTI_TABLES=1 SYLOW_HIP_MULTI_PREFETCH=0 LOG2N=18 python3 tools/dbg/time_multi.py 2>&1 | grep "^multi\|^pairing"
  echo "== tables, prefetch rep $rep";    SYLOW_HIP_MULTI_TABLES=1 SYLOW_HIP_MULTI_PREFETCH=1 LOG2N=18 python3 tools/dbg/time_multi.py 2>&1 | grep "^multi\|^pairing"
done
SYLOW_HIP_MULTI_PREFETCH=0 tools/dbg/kt_multi.sh
