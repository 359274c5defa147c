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
n = 64
p, q, ka, kb = make_points(eng, n, 5)
gt1, is1 = eng.empty((48, 1)), eng.empty((1,), np.uint8)
for k in (1, 2, 4, 6):
    pm = eng.empty((8, k)).upload(np.ascontiguousarray(p.download()[:, :k])); qm = eng.empty((16, k)).upload(np.ascontiguousarray(q.download()[:, :k]))
    off = eng.to_device(np.array([0, k], dtype=np.uint64))
    t1 = timed(lambda: eng._call("sylow_hip_multi_pairing_batch", pm.ptr, None, qm.ptr, None, off.ptr, 1, k, 1, gt1.ptr, is1.ptr))
    a = gt1.download().copy()
    t2 = timed(lambda: eng._call("sylow_hip_pairing_product_batch", pm.ptr, None, qm.ptr, None, k, 1, gt1.ptr, is1.ptr))
    assert np.array_equal(a, gt1.download())
    print("one job of %d pairs: multi_pairing %.2f ms   pairing_product %.2f ms" % (k, t1, t2))
