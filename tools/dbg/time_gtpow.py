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
n = 1 << 18
p, q, ka, kb = make_points(eng, n, 5)
gt = eng.empty((48, n)); out = eng.empty((48, n))
eng._call("sylow_hip_pairing_batch", p.ptr, None, q.ptr, None, gt.ptr, n)
print("%s gt_pow 2^18: %.2f ms" % (os.path.basename(os.environ.get("SYLOW_HIP_LIB", "current")), timed(lambda: eng._call("sylow_hip_gt_pow_batch", gt.ptr, ka.ptr, out.ptr, n))))
pp = eng.empty((48, 1)); iso = eng.empty((1,), np.uint8)
print("   product 2^18: %.2f ms" % timed(lambda: eng._call("sylow_hip_pairing_product_batch", p.ptr, None, q.ptr, None, n, 0, pp.ptr, iso.ptr)))
