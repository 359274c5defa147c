"""pairing / verify kernel time against the batch size: what one GPU of a strong-scaled 2^20 batch (2^17 at N = 8) pays"""
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
for lg in (20, 19, 18, 17, 16, 15):
    n = 1 << lg
    p, q, ka, kb = make_points(eng, n, 5)
    gt = eng.empty((48, n))
    tp = timed(lambda: eng._call("sylow_hip_pairing_batch", p.ptr, None, q.ptr, None, gt.ptr, n))
    rng = np.random.default_rng(7)
    msgs = rng.integers(0, 256, size=(n, 32), dtype=np.uint8)
    dm, doff = eng.to_device(msgs.reshape(-1)), eng.to_device(np.arange(n + 1, dtype=np.uint64) * np.uint64(32))
    g2 = eng.empty((16, n)).upload(np.repeat(limbs_row(G2).T, n, axis=1))
    pk, pki, sig, sigi, ok = eng.empty((16, n)), eng.empty((n,), np.uint8), eng.empty((8, n)), eng.empty((n,), np.uint8), eng.empty((n,), np.uint8)
    eng._call("sylow_hip_g2_scalar_mul_subgroup_batch", g2.ptr, None, ka.ptr, pk.ptr, pki.ptr, n)
    eng._call("sylow_hip_bls_sign_batch", ka.ptr, dm.ptr, doff.ptr, sig.ptr, sigi.ptr, n)
    tv = timed(lambda: eng._call("sylow_hip_bls_verify_batch", pk.ptr, None, dm.ptr, doff.ptr, sig.ptr, None, ok.ptr, n))
    print("n = 2^%d: pairing %.2f ms (%.2f M/s)   verify %.2f ms (%.2f M/s)" % (lg, tp, n / tp / 1e3, tv, n / tv / 1e3))
    del p, q, ka, kb, gt, dm, doff, g2, pk, pki, sig, sigi, ok
