cd $GRAFT_REPO_ROOT
for r in 1 2 3; do for L in tools/ab/lib_g1base.so tools/ab/lib_g1inl.so; do echo "== $L"; SYLOW_HIP_LIB=$PWD/$L python3 tools/dbg/time_group.py 2>&1 | grep "g1 scalar\|bls_sign"; done; done
SYLOW_HIP_LIB=$PWD/tools/ab/lib_g1inl.so python -m pytest tests/test_gpu_groups.py tests/test_gpu_hash_bls.py tests/test_gpu_small_rows.py tests/test_gpu_evm.py -m gpu -x -q 2>&1 | tail -2
