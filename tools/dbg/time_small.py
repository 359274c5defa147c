"""Latency of the one-wavefront-per-element route: pairing and verify at n = 1, 64, 1024 (or the sizes on the command line) for the
library SYLOW_HIP_LIB names."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch, sylow_amd
from bench import make_points, limbs_row, G2, SEED
eng = sylow_amd.Engine(0)
def timed(fn, reps=20):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps
name = os.path.basename(os.environ.get("SYLOW_HIP_LIB", "current"))
for n in ([int(a) for a in sys.argv[1:]] or (1, 64, 1024)):
    p, q, ka, kb = make_points(eng, n, 5)
    gt = eng.empty((48, n))
    tp = timed(lambda: eng._call("sylow_hip_pairing_batch", p.ptr, None, q.ptr, None, gt.ptr, n))
    msgs = np.random.default_rng(7).integers(0, 256, size=(n, 32), dtype=np.uint8)
    dm, doff = eng.to_device(msgs.reshape(-1)), eng.to_device(np.arange(n + 1, dtype=np.uint64) * np.uint64(32))
    g2 = eng.empty((16, n)).upload(np.repeat(limbs_row(G2).T, n, axis=1))
    pk, pki, sig, sigi, ok = eng.empty((16, n)), eng.empty((n,), np.uint8), eng.empty((8, n)), eng.empty((n,), np.uint8), eng.empty((n,), np.uint8)
    eng._call("sylow_hip_g2_scalar_mul_subgroup_batch", g2.ptr, None, ka.ptr, pk.ptr, pki.ptr, n)
    eng._call("sylow_hip_bls_sign_batch", ka.ptr, dm.ptr, doff.ptr, sig.ptr, sigi.ptr, n)
    tv = timed(lambda: eng._call("sylow_hip_bls_verify_batch", pk.ptr, None, dm.ptr, doff.ptr, sig.ptr, None, ok.ptr, n))
    print("%s n=%d: pairing %.3f ms, verify %.3f ms, all ok %d" % (name, n, tp, tv, int(ok.download().all())))
