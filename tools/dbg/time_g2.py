import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch, sylow_amd
from bench import make_points, limbs_row, G2
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
gt = eng.empty((48, n)); o2, o2i = eng.empty((16, n)), eng.empty((n,), np.uint8); o1, o1i = eng.empty((8, n)), eng.empty((n,), np.uint8)
ok = eng.empty((n,), np.uint8)
for rep in range(2):
    print("pairing      %.2f ms" % timed(lambda: eng._call("sylow_hip_pairing_batch", p.ptr, None, q.ptr, None, gt.ptr, n)))
    print("g2_scalarmul %.2f ms" % timed(lambda: eng._call("sylow_hip_g2_scalar_mul_batch", q.ptr, None, kb.ptr, o2.ptr, o2i.ptr, n)))
    print("g1_scalarmul %.2f ms" % timed(lambda: eng._call("sylow_hip_g1_scalar_mul_batch", p.ptr, None, ka.ptr, o1.ptr, o1i.ptr, n)))
    print("miller       %.2f ms" % timed(lambda: eng._call("sylow_hip_miller_loop_batch", p.ptr, q.ptr, gt.ptr, n)))
    print("subgroup     %.2f ms" % timed(lambda: eng._call("sylow_hip_g2_subgroup_check_batch", q.ptr, None, ok.ptr, n)))
    print("g2_mul_subgr %.2f ms" % timed(lambda: eng._call("sylow_hip_g2_scalar_mul_subgroup_batch", q.ptr, None, kb.ptr, o2.ptr, o2i.ptr, n)))
    print("g2_generator %.2f ms" % timed(lambda: eng._call("sylow_hip_g2_generator_mul_batch", kb.ptr, o2.ptr, o2i.ptr, n)))
    print("g1_generator %.2f ms" % timed(lambda: eng._call("sylow_hip_g1_generator_mul_batch", ka.ptr, o1.ptr, o1i.ptr, n)))
