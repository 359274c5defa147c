#!/bin/bash
cd $GRAFT_REPO_ROOT
SYLOW_HIP_LIB=$PWD/tools/ab/lib_B.so python -m pytest tests/test_gpu_hash_bls.py tests/test_gpu_evm.py -m gpu -q 2>&1 | grep -E "passed|failed"
for rep in 1 2 3; do for v in A B; do echo -n "$v: "; SYLOW_HIP_LIB=$PWD/tools/ab/lib_$v.so python3 tools/dbg/time_hash2.py 2>&1 | grep "msg_len  32\|msg_len 200" | tr '\n' ' '; echo; done; done
