#!/bin/bash
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_groups.py tests/test_gpu_full_size.py tests/test_gpu_api.py -m gpu -q -x 2>&1 | grep -E "passed|failed"
for rep in 1 2 3; do for v in 0 1; do echo -n "SYLOW_HIP_G1_TABLES=$v: "; SYLOW_HIP_G1_TABLES=$v python3 tools/dbg/time_g2.py 2>&1 | grep "g2_scalarmul\|g1_scalarmul\|g2_mul_subgr\|subgroup" | head -4 | tr '\n' ' '; echo; done; done
