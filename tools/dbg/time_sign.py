"""sign_batch latency over batch sizes for the route the environment selects (SYLOW_HIP_SIGN_WIDE_MAX / SYLOW_HIP_WIDE_TAIL), with an oracle
check of the first rows at every size."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch, sylow_amd
from oracle import coracle
eng = sylow_amd.Engine(0)
def timed(fn, reps=5):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps
rng = np.random.default_rng(11)
for n in [int(x) for x in (sys.argv[1:] or "1 8 64 512 2048 4096 8192 16384 32768 65536 131072".split())]:
    msgs = rng.integers(0, 256, size=(n, 32), dtype=np.uint8)
    dm, doff = eng.to_device(msgs.reshape(-1)), eng.to_device(np.arange(n + 1, dtype=np.uint64) * np.uint64(32))
    sk_aos = eng.xoshiro_fp_soa(100 + n, n).T.copy()
    sk = eng.to_device_soa(sk_aos, 4)
    sig, sigi = eng.empty((8, n)), eng.empty((n,), np.uint8)
    t = timed(lambda: eng._call("sylow_hip_bls_sign_batch", sk.ptr, dm.ptr, doff.ptr, sig.ptr, sigi.ptr, n))
    m = min(n, 16)
    exp, einf = coracle.g1_to_affine(coracle.sign(sk_aos[-m:], [msgs[i].tobytes() for i in range(n - m, n)]))
    got = eng.from_device_soa(sig)[-m:]
    print("n=%7d  sign %.3f ms  (%.2f M/s)  oracle rows ok: %d" % (n, t, n / t / 1e3, int(np.array_equal(got, exp) and not einf.any() and not sigi.download().any())))
