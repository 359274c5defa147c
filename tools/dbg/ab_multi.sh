#!/bin/bash
# A/B of the two multi-pairing routes (in-register shared squarings vs lines-to-HBM + table-driven loop), alternating processes on one box
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
  for mode in 0 1; do
    echo "== SYLOW_HIP_MULTI_TABLES=$mode rep $rep"
    SYLOW_HIP_MULTI_TABLES=$mode LOG2N=${LOG2N:-18} python3 tools/dbg/time_multi.py 2>&1 | grep -v "^W\|^E\|amdgpu.ids"
  done
done
