"""one launch each of miller_loop_batch and pairing_product_batch at n = 2^20 (for rocprofv3 --pmc)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, sylow_amd
from bench import make_points
eng = sylow_amd.Engine(0)
n = 1 << 20
p, q, ka, kb = make_points(eng, n, 5)
ml, gt1, is1 = eng.empty((48, n)), eng.empty((48, 1)), eng.empty((1,), np.uint8)
eng._call("sylow_hip_miller_loop_batch", p.ptr, q.ptr, ml.ptr, n)
eng._call("sylow_hip_pairing_product_batch", p.ptr, None, q.ptr, None, n, 0, gt1.ptr, is1.ptr)
eng.sync()
