"""A/B for multi-pair jobs: the in-register shared-squaring kernel (sylow_hip_multi_pairing_batch) against the table-driven route
built from existing entry points (g2_precompute_batch -> glued_miller_loop_precomputed_batch -> final_exp_batch)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch, sylow_amd
from bench import make_points
eng = sylow_amd.Engine(0)
def timed(fn, reps=3):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps
n = 1 << int(os.environ.get("LOG2N", "18"))
p, q, ka, kb = make_points(eng, n, 5)
gt = eng.empty((48, n)); iso = eng.empty((n,), np.uint8)
print("pairing n=%d: %.2f ms" % (n, timed(lambda: eng._call("sylow_hip_pairing_batch", p.ptr, None, q.ptr, None, gt.ptr, n))))
print("miller  n=%d: %.2f ms" % (n, timed(lambda: eng._call("sylow_hip_miller_loop_batch", p.ptr, q.ptr, gt.ptr, n))))
print("finexp  n=%d: %.2f ms" % (n, timed(lambda: eng._call("sylow_hip_final_exp_batch", gt.ptr, gt.ptr, n))))
tab = eng.empty((87 * 24, n))
print("g2_precompute n=%d: %.2f ms" % (n, timed(lambda: eng._call("sylow_hip_g2_precompute_batch", q.ptr, tab.ptr, n), 1)))
f = eng.empty((48, n))
for k in (1, 2, 4, 8):
    nj = n // k
    off = eng.to_device(np.arange(nj + 1, dtype=np.uint64) * np.uint64(k))
    t0 = timed(lambda: eng._call("sylow_hip_multi_pairing_batch", p.ptr, None, q.ptr, None, off.ptr, nj, nj * k, 1, gt.ptr, iso.ptr))
    t1 = timed(lambda: eng._call("sylow_hip_glued_miller_loop_precomputed_batch", tab.ptr, n, None, p.ptr, off.ptr, nj, nj * k, f.ptr))
    t2 = timed(lambda: eng._call("sylow_hip_final_exp_batch", f.ptr, gt.ptr, nj))
    t3 = timed(lambda: eng._call("sylow_hip_glued_miller_loop_batch", p.ptr, q.ptr, off.ptr, nj, nj * k, f.ptr))
    print("k=%d jobs=%d: in-register multi_pairing %.2f ms | raw glued in-register %.2f | table-driven glued loop %.2f + final exp %.2f = %.2f ms" % (k, nj, t0, t3, t1, t2, t1 + t2))
