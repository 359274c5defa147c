"""Latency of the group-law entry points at small batch sizes (one call, HIP-event timed)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch, sylow_amd
from bench import make_points
eng = sylow_amd.Engine(0)
def timed(fn, reps=10):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps
for n in [int(a) for a in sys.argv[1:]] or (1, 64, 1024, 8192):
    p, q, ka, kb = make_points(eng, n, 5)
    o1, o1i, o2, o2i = eng.empty((8, n)), eng.empty((n,), np.uint8), eng.empty((16, n)), eng.empty((n,), np.uint8)
    t1 = timed(lambda: eng._call("sylow_hip_g1_scalar_mul_batch", p.ptr, None, ka.ptr, o1.ptr, o1i.ptr, n))
    t2 = timed(lambda: eng._call("sylow_hip_g2_scalar_mul_batch", q.ptr, None, kb.ptr, o2.ptr, o2i.ptr, n))
    t3 = timed(lambda: eng._call("sylow_hip_g2_scalar_mul_subgroup_batch", q.ptr, None, kb.ptr, o2.ptr, o2i.ptr, n))
    t4 = timed(lambda: eng._call("sylow_hip_g1_add_batch", p.ptr, None, p.ptr, None, o1.ptr, o1i.ptr, n))
    print("n=%5d  g1_scalar_mul %.3f ms  g2_scalar_mul %.3f ms  g2 (r-torsion) %.3f ms  g1_add %.3f ms" % (n, t1, t2, t3, t4))
