import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch, sylow_amd
from bench import make_points, SEED
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
o, oi = eng.empty((8, n)), eng.empty((n,), np.uint8)
print("g1 scalar mul: %.2f ms" % timed(lambda: eng._call("sylow_hip_g1_scalar_mul_batch", p.ptr, None, ka.ptr, o.ptr, oi.ptr, n)))
print("g1 generator mul: %.2f ms" % timed(lambda: eng._call("sylow_hip_g1_generator_mul_batch", ka.ptr, o.ptr, oi.ptr, n)))
rng = np.random.default_rng(7)
msgs = rng.integers(0, 256, size=(n, 32), dtype=np.uint8)
dm, doff = eng.to_device(msgs.reshape(-1)), eng.to_device(np.arange(n + 1, dtype=np.uint64) * np.uint64(32))
print("bls_sign: %.2f ms" % timed(lambda: eng._call("sylow_hip_bls_sign_batch", ka.ptr, dm.ptr, doff.ptr, o.ptr, oi.ptr, n)))
print("hash_to_g1: %.2f ms" % timed(lambda: eng._call("sylow_hip_hash_to_g1_batch", dm.ptr, doff.ptr, None, 0, o.ptr, oi.ptr, n)))
