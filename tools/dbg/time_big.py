"""k_pairing beyond the metric's batch: 2^20, 2^21, 2^22 elements in one launch -- how much of the 2^20 time is ramp and drain"""
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
for lg in (20, 21, 22):
    n = 1 << lg
    p, q, ka, kb = make_points(eng, n, 5)
    gt = eng.empty((48, n))
    t = timed(lambda: eng._call("sylow_hip_pairing_batch", p.ptr, None, q.ptr, None, gt.ptr, n))
    print("n = 2^%d: pairing %.2f ms (%.3f M/s)" % (lg, t, n / t / 1e3))
    del p, q, ka, kb, gt
