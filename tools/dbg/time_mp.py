import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch, sylow_amd
from bench import make_points
eng = sylow_amd.Engine(0)
def timed(fn, reps=5):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps
n = 1 << 20
p, q, ka, kb = make_points(eng, n, 5)
out = []
for k, nj in ((2, 1 << 16), (3, 1 << 16), (4, 1 << 16), (4, 1 << 18), (8, 1 << 17)):
    off = eng.to_device(np.arange(nj + 1, dtype=np.uint64) * np.uint64(k))
    gt = eng.empty((48, nj)); iso = eng.empty((nj,), np.uint8)
    t = timed(lambda: eng._call("sylow_hip_multi_pairing_batch", p.ptr, None, q.ptr, None, off.ptr, nj, nj * k, 1, gt.ptr, iso.ptr))
    out.append("k=%d/%d: %.2f" % (k, nj, t))
gt1, is1 = eng.empty((48, 1)), eng.empty((1,), np.uint8)
t = timed(lambda: eng._call("sylow_hip_pairing_product_batch", p.ptr, None, q.ptr, None, n, 1, gt1.ptr, is1.ptr), 3)
out.append("product 2^20: %.2f" % t)
print(os.path.basename(os.environ.get("SYLOW_HIP_LIB", "default")), "  ".join(out))
