"""upper bound on what a coalesced window-table lookup could buy: the same scalar in every lane makes every lane read the same table row"""
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
n = 1 << 20
p, q, ka, kb = make_points(eng, n, 5)
kh = ka.download()
same = eng.empty((4, n)).upload(np.repeat(kh[:, :1], n, axis=1))
o1, o1i = eng.empty((8, n)), eng.empty((n,), np.uint8); o2, o2i = eng.empty((16, n)), eng.empty((n,), np.uint8)
for name, k in (("random scalars", ka), ("one scalar   ", same)):
    print(name, "g1 %.2f ms  g2 gls %.2f ms  g2 any %.2f ms" % (
        timed(lambda: eng._call("sylow_hip_g1_scalar_mul_batch", p.ptr, None, k.ptr, o1.ptr, o1i.ptr, n)),
        timed(lambda: eng._call("sylow_hip_g2_scalar_mul_subgroup_batch", q.ptr, None, k.ptr, o2.ptr, o2i.ptr, n)),
        timed(lambda: eng._call("sylow_hip_g2_scalar_mul_batch", q.ptr, None, k.ptr, o2.ptr, o2i.ptr, n))))
