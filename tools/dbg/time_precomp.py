import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch, sylow_amd
from bench import make_points
eng = sylow_amd.Engine(0)
n = 1 << 18
p, q, ka, kb = make_points(eng, n, 5)
def timed(fn, reps=3):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps
co = eng.empty((87 * 24, n))
f1, f2 = eng.empty((48, n)), eng.empty((48, n))
print("g2_precompute      %.2f ms" % timed(lambda: eng._call("sylow_hip_g2_precompute_batch", q.ptr, co.ptr, n)))
print("miller_loop        %.2f ms" % timed(lambda: eng._call("sylow_hip_miller_loop_batch", p.ptr, q.ptr, f1.ptr, n)))
print("miller_precomputed %.2f ms" % timed(lambda: eng._call("sylow_hip_miller_loop_precomputed_batch", co.ptr, n, None, p.ptr, f2.ptr, n)))
assert np.array_equal(f1.download(), f2.download())
# one cached key, many points
idx = eng.to_device(np.zeros(n, dtype=np.uint64))
print("miller_precomputed (one table for all) %.2f ms" % timed(lambda: eng._call("sylow_hip_miller_loop_precomputed_batch", co.ptr, n, idx.ptr, p.ptr, f2.ptr, n)))
