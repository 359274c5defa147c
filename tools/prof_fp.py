#!/usr/bin/env python3
"""Driver for tools/prof_fp.sh: the HBM-bound Fp micro-batch kernels (k_fp_binop: mul / add / sub at 2^24 and 2^20 elements),
a few launches each, device-resident inputs."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sylow_amd
eng = sylow_amd.Engine(0)
for log2n in (24, 20):
    n = 1 << log2n
    a = eng.empty((4, n)).upload(eng.xoshiro_fp_soa(1, n)); b = eng.empty((4, n)).upload(eng.xoshiro_fp_soa(2, n)); o = eng.empty((4, n))
    for name in ("mul", "add", "sub"):
        for _ in range(5):
            eng._call(f"sylow_hip_fp_{name}_batch", a.ptr, b.ptr, o.ptr, n)
    eng.sync()
