import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch, sylow_amd
from bench import limbs_row, G2
eng = sylow_amd.Engine(0)
nv = 1 << 20
rng = np.random.default_rng(7)
msgs = rng.integers(0, 256, size=(nv, 32), dtype=np.uint8)
dm, doff = eng.to_device(msgs.reshape(-1)), eng.to_device(np.arange(nv + 1, dtype=np.uint64) * np.uint64(32))
sk = eng.empty((4, nv)).upload(eng.xoshiro_fp_soa(99, nv))
g2 = eng.empty((16, nv)).upload(np.repeat(limbs_row(G2).T, nv, axis=1))
pk, pki, sig, sigi, ok = eng.empty((16, nv)), eng.empty((nv,), np.uint8), eng.empty((8, nv)), eng.empty((nv,), np.uint8), eng.empty((nv,), np.uint8)
eng._call("sylow_hip_g2_scalar_mul_batch", g2.ptr, None, sk.ptr, pk.ptr, pki.ptr, nv)
eng._call("sylow_hip_bls_sign_batch", sk.ptr, dm.ptr, doff.ptr, sig.ptr, sigi.ptr, nv)
def timed(fn, reps=3):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps
tf = timed(lambda: eng._call("sylow_hip_bls_verify_batch", pk.ptr, None, dm.ptr, doff.ptr, sig.ptr, None, ok.ptr, nv))
assert ok.download().all()
ts = timed(lambda: eng._call("sylow_hip_bls_verify_same_signer_batch", pk.ptr, None, dm.ptr, doff.ptr, sig.ptr, None, ok.ptr, nv))
print("verify %.2f ms  same-signer shape %.2f ms" % (tf, ts))
