"""latency of the one-boolean shapes: a one-pair product, a small product, the same-signer aggregate"""
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
n = 1 << 12
p, q, ka, kb = make_points(eng, n, 5)
gt1, is1 = eng.empty((48, 1)), eng.empty((1,), np.uint8)
out = []
for m in (1, 2, 4, 64, 4096):
    pm = eng.empty((8, m)).upload(np.ascontiguousarray(p.download()[:, :m])); qm = eng.empty((16, m)).upload(np.ascontiguousarray(q.download()[:, :m]))
    out.append("product(%d) %.2f ms" % (m, timed(lambda: eng._call("sylow_hip_pairing_product_batch", pm.ptr, None, qm.ptr, None, m, 1, gt1.ptr, is1.ptr))))
nv = 1 << 20
rng = np.random.default_rng(7)
msgs_np = rng.integers(0, 256, size=(nv, 32), dtype=np.uint8)
dm, doff = eng.to_device(msgs_np.reshape(-1)), eng.to_device(np.arange(nv + 1, dtype=np.uint64) * np.uint64(32))
k1 = eng.xoshiro_fp_soa(SEED + 9, 1)
sk1, sk1one = eng.empty((4, nv)).upload(np.repeat(k1, nv, axis=1)), eng.empty((4, 1)).upload(k1)
pk1, pk1i = eng.empty((16, 1)), eng.empty((1,), np.uint8)
g2one = eng.empty((16, 1)).upload(limbs_row(G2).T.copy())
sig, sigi = eng.empty((8, nv)), eng.empty((nv,), np.uint8)
eng._call("sylow_hip_bls_sign_batch", sk1.ptr, dm.ptr, doff.ptr, sig.ptr, sigi.ptr, nv)
eng._call("sylow_hip_g2_scalar_mul_subgroup_batch", g2one.ptr, None, sk1one.ptr, pk1.ptr, pk1i.ptr, 1)
for m in (1 << 20, 1 << 16, 1 << 12):
    t = timed(lambda: eng._call("sylow_hip_bls_aggregate_verify_batch", pk1.ptr, None, 1, dm.ptr, doff.ptr, sig.ptr, None, m, None, gt1.ptr, is1.ptr))
    out.append("same-signer(2^%d) %.2f ms ok=%d" % (m.bit_length() - 1, t, int(is1.download()[0])))
print("wide=%s  " % os.environ.get("SYLOW_HIP_WIDE_TAIL", "1") + "  ".join(out))
