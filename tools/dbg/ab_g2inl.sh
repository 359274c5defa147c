cd $GRAFT_REPO_ROOT
for r in 1 2; do for L in tools/ab/lib_g1inl.so tools/ab/lib_g2inl.so; do echo "== $L"; SYLOW_HIP_LIB=$PWD/$L python3 tools/dbg/time_g2.py 2>&1 | tail -8; SYLOW_HIP_LIB=$PWD/$L python3 tools/dbg/time_decode.py 2>&1 | tail -5; done; done
SYLOW_HIP_LIB=$PWD/tools/ab/lib_g2inl.so python -m pytest tests/test_gpu_groups.py tests/test_gpu_small_rows.py tests/test_gpu_evm.py tests/test_gpu_bytes.py tests/test_gpu_fr_threshold.py -m gpu -x -q 2>&1 | tail -2
