import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch, sylow_amd
from bench import make_points, limbs_row, G2, SEED
eng = sylow_amd.Engine(0)
def timed(fn, reps=3):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps
n = 1 << int(os.environ.get("LOG2N", "20"))
p, q, ka, kb = make_points(eng, n, 5)
gt1, is1 = eng.empty((48, 1)), eng.empty((1,), np.uint8)
for m in (n, n // 4, n // 16, 4096, 301):
    t = timed(lambda: eng._call("sylow_hip_pairing_product_batch", p.ptr, None, q.ptr, None, m, 1, gt1.ptr, is1.ptr))
    print("pairing_product over %d pairs (first %d of the arrays): %.2f ms -> %.2f M pairs/s" % (m, m, t, m / t / 1e3))
nv = n
rng = np.random.default_rng(7)
msgs_np = rng.integers(0, 256, size=(nv, 32), dtype=np.uint8)
dm, doff = eng.to_device(msgs_np.reshape(-1)), eng.to_device(np.arange(nv + 1, dtype=np.uint64) * np.uint64(32))
sk = eng.empty((4, nv)).upload(eng.xoshiro_fp_soa(SEED + 4, nv))
g2 = eng.empty((16, nv)).upload(np.repeat(limbs_row(G2).T, nv, axis=1))
pk, pki, sig, sigi = eng.empty((16, nv)), eng.empty((nv,), np.uint8), eng.empty((8, nv)), eng.empty((nv,), np.uint8)
eng._call("sylow_hip_g2_scalar_mul_subgroup_batch", g2.ptr, None, sk.ptr, pk.ptr, pki.ptr, nv)
eng._call("sylow_hip_bls_sign_batch", sk.ptr, dm.ptr, doff.ptr, sig.ptr, sigi.ptr, nv)
t = timed(lambda: eng._call("sylow_hip_bls_aggregate_verify_batch", pk.ptr, None, nv, dm.ptr, doff.ptr, sig.ptr, None, nv, None, gt1.ptr, is1.ptr))
print("aggregate verify %d sigs: %.2f ms -> %.2f M sigs/s, ok=%d" % (nv, t, nv / t / 1e3, int(is1.download()[0])))
