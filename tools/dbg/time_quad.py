"""Mid-size batches: pairing_batch / bls_verify_batch with the library defaults (lane-quad route up to 16384 elements, quad tails beside whole
rounds: plk_quad.hip) against the lane-pair kernels alone (QUAD_MAX option 0), same box, same inputs.  Column "quad" = the defaults.
usage: time_quad.py [sizes...]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import sylow_amd
from bench import SEED, make_points
eng = sylow_amd.Engine(0)   # SYLOW_HIP_LIB selects the build
sizes = [int(x) for x in sys.argv[1:]] or [6144, 7168, 8192, 12288, 16384, 24576, 32768, 33000, 40000, 49152, 65536, 70000, 81920, 98304, 100000, 114688, 131072]
nmax = max(sizes)
p, q, ka, kb = make_points(eng, nmax, SEED + 3)
ph, qh = p.download(), q.download()
for n in sizes:
    dp, dq = eng.empty((8, n)).upload(np.ascontiguousarray(ph[:, :n])), eng.empty((16, n)).upload(np.ascontiguousarray(qh[:, :n]))
    dg = eng.empty((48, n))
    res = {}
    for name, qm in (("pair", 0), ("quad", -1)):
        eng.set_option("QUAD_MAX", qm)
        eng._call("sylow_hip_pairing_batch", dp.ptr, None, dq.ptr, None, dg.ptr, n); eng.sync()
        best = 1e9
        for _ in range(5):
            t0 = time.perf_counter()
            eng._call("sylow_hip_pairing_batch", dp.ptr, None, dq.ptr, None, dg.ptr, n); eng.sync()
            best = min(best, time.perf_counter() - t0)
        res[name] = (best * 1e3, dg.download().sum(dtype=np.uint64))
    eng.set_option("QUAD_MAX", -1)
    print("n = %6d   lane pair %.3f ms (%.2f M/s)   quad %.3f ms (%.2f M/s)   same: %s" % (n, res["pair"][0], n / res["pair"][0] / 1e3, res["quad"][0], n / res["quad"][0] / 1e3, res["pair"][1] == res["quad"][1]), flush=True)

# verify at the same sizes
rng = np.random.default_rng(7)
msgs = rng.integers(0, 256, size=(nmax, 32), dtype=np.uint8)
dm, doff = eng.to_device(msgs.reshape(-1)), eng.to_device(np.arange(nmax + 1, dtype=np.uint64) * np.uint64(32))
sk = eng.empty((4, nmax)).upload(eng.xoshiro_fp_soa(SEED + 4, nmax))
pk, pki = eng.empty((16, nmax)), eng.empty((nmax,), np.uint8)
sig, sigi = eng.empty((8, nmax)), eng.empty((nmax,), np.uint8)
eng._call("sylow_hip_g2_generator_mul_batch", sk.ptr, pk.ptr, pki.ptr, nmax)
eng._call("sylow_hip_bls_sign_batch", sk.ptr, dm.ptr, doff.ptr, sig.ptr, sigi.ptr, nmax)
pkh, sigh = pk.download(), sig.download()
for n in sizes:
    dpk, dsig = eng.empty((16, n)).upload(np.ascontiguousarray(pkh[:, :n])), eng.empty((8, n)).upload(np.ascontiguousarray(sigh[:, :n]))
    ok = eng.empty((n,), np.uint8)
    res = {}
    for name, qm in (("pair", 0), ("quad", -1)):
        eng.set_option("QUAD_MAX", qm)
        eng._call("sylow_hip_bls_verify_batch", dpk.ptr, None, dm.ptr, doff.ptr, dsig.ptr, None, ok.ptr, n); eng.sync()
        best = 1e9
        for _ in range(5):
            t0 = time.perf_counter()
            eng._call("sylow_hip_bls_verify_batch", dpk.ptr, None, dm.ptr, doff.ptr, dsig.ptr, None, ok.ptr, n); eng.sync()
            best = min(best, time.perf_counter() - t0)
        res[name] = (best * 1e3, int(ok.download().sum()))
    eng.set_option("QUAD_MAX", -1)
    print("verify n = %6d   lane pair %.3f ms (%.2f M/s)   quad %.3f ms (%.2f M/s)   ok: %d %d" % (n, res["pair"][0], n / res["pair"][0] / 1e3, res["quad"][0], n / res["quad"][0] / 1e3, res["pair"][1], res["quad"][1]), flush=True)
