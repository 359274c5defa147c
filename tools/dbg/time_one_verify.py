import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch, sylow_amd
from bench import limbs_row, G2, SEED
eng = sylow_amd.Engine(0)
def timed(fn, reps=5):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps
nv = 1
msgs = np.frombuffer(b"\x00\x00\x00\x14", dtype=np.uint8).copy()
dm, doff = eng.to_device(msgs), eng.to_device(np.array([0, 4], dtype=np.uint64))
sk = eng.empty((4, 1)).upload(eng.xoshiro_fp_soa(SEED + 4, 1))
g2 = eng.empty((16, 1)).upload(limbs_row(G2).T.copy())
pk, pki, sig, sigi, ok = eng.empty((16, 1)), eng.empty((1,), np.uint8), eng.empty((8, 1)), eng.empty((1,), np.uint8), eng.empty((1,), np.uint8)
gt1, is1 = eng.empty((48, 1)), eng.empty((1,), np.uint8)
eng._call("sylow_hip_g2_scalar_mul_subgroup_batch", g2.ptr, None, sk.ptr, pk.ptr, pki.ptr, 1)
print("sign(1)    %.2f ms" % timed(lambda: eng._call("sylow_hip_bls_sign_batch", sk.ptr, dm.ptr, doff.ptr, sig.ptr, sigi.ptr, 1)))
print("verify(1)  %.2f ms ok=%d" % (timed(lambda: eng._call("sylow_hip_bls_verify_batch", pk.ptr, None, dm.ptr, doff.ptr, sig.ptr, None, ok.ptr, 1)), int(ok.download()[0])))
print("aggregate(1) %.2f ms ok=%d" % (timed(lambda: eng._call("sylow_hip_bls_aggregate_verify_batch", pk.ptr, None, 1, dm.ptr, doff.ptr, sig.ptr, None, 1, None, gt1.ptr, is1.ptr)), int(is1.download()[0])))
