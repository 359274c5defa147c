"""bls_batch_verify_weighted at 2^18 signatures (run with SYLOW_HIP_AGG_FORK=0 for the in-line schedule)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch, sylow_amd
from bench import make_points
eng = sylow_amd.Engine(0)
n = 1 << 18
g = np.random.default_rng(3)
msgs = g.integers(0, 256, size=(n, 32), dtype=np.uint8)
dm, doff = eng.to_device(msgs.reshape(-1)), eng.to_device(np.arange(n + 1, dtype=np.uint64) * np.uint64(32))
p, q, ka, kb = make_points(eng, n, 5)
sig, sigi = eng.empty((8, n)), eng.empty((n,), np.uint8)
eng._call("sylow_hip_bls_sign_batch", ka.ptr, dm.ptr, doff.ptr, sig.ptr, sigi.ptr, n)
pk, pki = eng.empty((16, n)), eng.empty((n,), np.uint8)
g2 = eng.empty((16, n)).upload(np.repeat(np.array(__import__("bench").limbs_row(__import__("bench").G2)).T, n, axis=1))
eng._call("sylow_hip_g2_scalar_mul_subgroup_batch", g2.ptr, None, ka.ptr, pk.ptr, pki.ptr, n)
w = eng.to_device_soa((g.integers(0, 1 << 62, size=(n, 4), dtype=np.uint64) & np.array([2**64 - 1, 2**64 - 1, 0, 0], dtype=np.uint64)), 4)
gt, one = eng.empty((48,)), eng.empty((1,), np.uint8)
def timed(fn, reps=3):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps
t = timed(lambda: eng._call("sylow_hip_bls_batch_verify_weighted", pk.ptr, None, n, dm.ptr, doff.ptr, sig.ptr, None, w.ptr, n, None, gt.ptr, one.ptr))
print("weighted 2^18 (AGG_FORK=%s): %.2f ms, is_one %d" % (os.environ.get("SYLOW_HIP_AGG_FORK", "1"), t, int(one.download()[0])))
