"""hash_to_g1 / hash_to_field / Fp inv / sqrt / is_square timings at 2^20 (one element per lane)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch, sylow_amd
eng = sylow_amd.Engine(0)
n = 1 << 20
rng = np.random.default_rng(7)
msgs = rng.integers(0, 256, size=(n, 32), dtype=np.uint8)
dm, doff = eng.to_device(msgs.reshape(-1)), eng.to_device(np.arange(n + 1, dtype=np.uint64) * np.uint64(32))
hh, hhi = eng.empty((8, n)), eng.empty((n,), np.uint8)
uu = eng.empty((8, n))
a = eng.empty((4, n)).upload(eng.xoshiro_fp_soa(3, n))
o = eng.empty((4, n)); fl = eng.empty((n,), np.uint8)
def timed(fn, reps=5):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps
print("hash_to_g1   %.2f ms" % timed(lambda: eng._call("sylow_hip_hash_to_g1_batch", dm.ptr, doff.ptr, None, 0, hh.ptr, hhi.ptr, n)))
print("hash_to_field %.2f ms" % timed(lambda: eng._call("sylow_hip_hash_to_field_batch", dm.ptr, doff.ptr, None, 0, uu.ptr, n)))
print("fp_inv       %.2f ms" % timed(lambda: eng._call("sylow_hip_fp_inv_batch", a.ptr, o.ptr, n)))
print("fp_sqrt      %.2f ms" % timed(lambda: eng._call("sylow_hip_fp_sqrt_batch", a.ptr, o.ptr, fl.ptr, n)))
print("fp_mul       %.3f ms" % timed(lambda: eng._call("sylow_hip_fp_mul_batch", a.ptr, a.ptr, o.ptr, n)))
print("fp_is_square %.2f ms" % timed(lambda: eng._call("sylow_hip_fp_is_square_batch", a.ptr, fl.ptr, n)))
