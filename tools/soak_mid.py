"""Determinism soak of the mid-size routes of round 6 (plk_quad.hip: lane quads up to 16384 elements; quad tails on a side stream beside whole
rounds of lane pairs): at each size the first default-route result must equal the lane-pair-only result (QUAD_MAX option 0) row for row, the
planted wrong signatures must be the only zero flags, and every repetition must be bit-identical to the first.
`python tools/soak_mid.py [seconds]`"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import sylow_amd
from bench import make_points, SEED

eng = sylow_amd.Engine(0)
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
sizes = [7001, 12289, 16384, 32768 + 777, 49152, 65536 + 4099, 98304 + 1234]
nmax = max(sizes)
p, q, ka, kb = make_points(eng, nmax, SEED + 78)
ph, qh = p.download(), q.download()
msgs = np.random.default_rng(4).integers(0, 256, size=(nmax, 32), dtype=np.uint8)
dm, doff = eng.to_device(msgs.reshape(-1)), eng.to_device(np.arange(nmax + 1, dtype=np.uint64) * np.uint64(32))
pk, pki, sig, sigi = eng.empty((16, nmax)), eng.empty((nmax,), np.uint8), eng.empty((8, nmax)), eng.empty((nmax,), np.uint8)
eng._call("sylow_hip_g2_generator_mul_batch", ka.ptr, pk.ptr, pki.ptr, nmax)
eng._call("sylow_hip_bls_sign_batch", ka.ptr, dm.ptr, doff.ptr, sig.ptr, sigi.ptr, nmax)
pkh, sigh = pk.download(), sig.download()
state = {}
for n in sizes:
    bad = np.arange(3, n, 997)
    s = np.ascontiguousarray(sigh[:, :n]); s[:, bad] = s[:, (bad + 1) % n]
    d = dict(p=eng.empty((8, n)).upload(np.ascontiguousarray(ph[:, :n])), q=eng.empty((16, n)).upload(np.ascontiguousarray(qh[:, :n])),
             pk=eng.empty((16, n)).upload(np.ascontiguousarray(pkh[:, :n])), sig=eng.empty((8, n)).upload(s), gt=eng.empty((48, n)), ok=eng.empty((n,), np.uint8))
    want = np.ones(n, np.uint8); want[bad] = 0

    def run(d=d, n=n):
        eng._call("sylow_hip_pairing_batch", d["p"].ptr, None, d["q"].ptr, None, d["gt"].ptr, n)
        eng._call("sylow_hip_bls_verify_batch", d["pk"].ptr, None, dm.ptr, doff.ptr, d["sig"].ptr, None, d["ok"].ptr, n)
        return d["gt"].download(), d["ok"].download()

    eng.set_option("QUAD_MAX", 0)
    g_pair, o_pair = run()
    eng.set_option("QUAD_MAX", -1)
    g0, o0 = run()
    assert np.array_equal(g0, g_pair) and np.array_equal(o0, o_pair), ("default route differs from the lane-pair route", n)
    assert np.array_equal(o0, want), ("verify pattern", n)
    state[n] = (run, g0, o0)
t0, rounds = time.time(), 0
while time.time() - t0 < budget:
    for n, (run, g0, o0) in state.items():
        g, o = run()
        assert np.array_equal(g, g0) and np.array_equal(o, o0), ("nondeterministic", n, rounds)
    rounds += 1
print("soak_mid ok: %d rounds over sizes %s in %.0f s (every repetition bit-identical; default route == lane-pair route at every size)" % (rounds, sizes, time.time() - t0))
