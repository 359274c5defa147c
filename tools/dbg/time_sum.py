"""HIP-event times of the G1 batch sum and the two aggregate verifiers at several sizes (one GPU)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch, sylow_amd
from bench import make_points, limbs_row, G2, SEED
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
o, oi = eng.empty((8, 1)), eng.empty((1,), np.uint8)
for m in (1 << 20, 1 << 18, 1 << 16, 1 << 12, 600):
    pm = eng.empty((8, m)).upload(np.ascontiguousarray(p.download()[:, :m]))
    t = timed(lambda: eng._call("sylow_hip_g1_sum_batch", pm.ptr, None, m, o.ptr, oi.ptr))
    print("g1_sum %8d points: %.3f ms" % (m, t))
rng = np.random.default_rng(7)
msgs_np = rng.integers(0, 256, size=(n, 32), dtype=np.uint8)
dm, doff = eng.to_device(msgs_np.reshape(-1)), eng.to_device(np.arange(n + 1, dtype=np.uint64) * np.uint64(32))
sk = eng.empty((4, n)).upload(eng.xoshiro_fp_soa(SEED + 4, n))
g2 = eng.empty((16, n)).upload(np.repeat(limbs_row(G2).T, n, axis=1))
pk, pki, sig, sigi = eng.empty((16, n)), eng.empty((n,), np.uint8), eng.empty((8, n)), eng.empty((n,), np.uint8)
gt1, is1 = eng.empty((48, 1)), eng.empty((1,), np.uint8)
eng._call("sylow_hip_g2_scalar_mul_subgroup_batch", g2.ptr, None, sk.ptr, pk.ptr, pki.ptr, n)
eng._call("sylow_hip_bls_sign_batch", sk.ptr, dm.ptr, doff.ptr, sig.ptr, sigi.ptr, n)
t = timed(lambda: eng._call("sylow_hip_bls_aggregate_verify_batch", pk.ptr, None, n, dm.ptr, doff.ptr, sig.ptr, None, n, None, gt1.ptr, is1.ptr), 3)
print("aggregate verify (distinct keys) 2^20: %.2f ms ok=%d" % (t, int(is1.download()[0])))
k1 = eng.xoshiro_fp_soa(SEED + 9, 1)
sk1, sk1one = eng.empty((4, n)).upload(np.repeat(k1, n, axis=1)), eng.empty((4, 1)).upload(k1)
pk1, pk1i = eng.empty((16, 1)), eng.empty((1,), np.uint8)
g2one = eng.empty((16, 1)).upload(limbs_row(G2).T.copy())
eng._call("sylow_hip_bls_sign_batch", sk1.ptr, dm.ptr, doff.ptr, sig.ptr, sigi.ptr, n)
eng._call("sylow_hip_g2_scalar_mul_subgroup_batch", g2one.ptr, None, sk1one.ptr, pk1.ptr, pk1i.ptr, 1)
for m in (1 << 20, 1 << 16):
    t = timed(lambda: eng._call("sylow_hip_bls_aggregate_verify_batch", pk1.ptr, None, 1, dm.ptr, doff.ptr, sig.ptr, None, m, None, gt1.ptr, is1.ptr), 3)
    print("aggregate verify (one key) %d: %.2f ms ok=%d" % (m, t, int(is1.download()[0])))
