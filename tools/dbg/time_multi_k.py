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
k = int(os.environ.get("K", "4"))
p, q, ka, kb = make_points(eng, n, 5)
gt = eng.empty((48, n)); iso = eng.empty((n,), np.uint8)
nj = n // k
off = eng.to_device(np.arange(nj + 1, dtype=np.uint64) * np.uint64(k))
t = timed(lambda: eng._call("sylow_hip_multi_pairing_batch", p.ptr, None, q.ptr, None, off.ptr, nj, nj * k, 1, gt.ptr, iso.ptr))
print("multi k=%d jobs=%d: %.2f ms -> %.2f M jobs/s" % (k, nj, t, nj / t / 1e3))
t = timed(lambda: eng._call("sylow_hip_glued_miller_loop_batch", p.ptr, q.ptr, off.ptr, nj, nj * k, gt.ptr))
print("raw glued k=%d jobs=%d: %.2f ms" % (k, nj, t))
