import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sylow_amd
from bench import make_points
eng = sylow_amd.Engine(0)
n = 1 << 20
p, q, ka, kb = make_points(eng, n, 5)
ml, gt = eng.empty((48, n)), eng.empty((48, n))
eng._call("sylow_hip_miller_loop_batch", p.ptr, q.ptr, ml.ptr, n)
for _ in range(2):
    eng._call("sylow_hip_final_exp_batch", ml.ptr, gt.ptr, n)
eng.sync()
