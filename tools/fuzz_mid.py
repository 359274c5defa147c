"""Seeded differential fuzz of the mid-size routes (lane quads, quad tails beside whole rounds: plk_quad.hip) against the lane-pair kernels alone
(QUAD_MAX option 0): random batch sizes between the one-wavefront caps and four rounds, random identity flags on both sides, random wrong
signatures, random single-key / per-element keys; every Gt value and every flag must agree, and the flags must equal the planted pattern.
   python3 tools/fuzz_mid.py [seconds] [seed]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import sylow_amd
from bench import make_points, SEED

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed = int(sys.argv[2], 0) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed)
eng = sylow_amd.Engine(0)
NMAX = 1 << 17
p, q, ka, kb = make_points(eng, NMAX, SEED + 79 + seed)
ph, qh = p.download(), q.download()
msgs = rng.integers(0, 256, size=(NMAX, 32), dtype=np.uint8)
dm, doff = eng.to_device(msgs.reshape(-1)), eng.to_device(np.arange(NMAX + 1, dtype=np.uint64) * np.uint64(32))
pk, pki, sig, sigi = eng.empty((16, NMAX)), eng.empty((NMAX,), np.uint8), eng.empty((8, NMAX)), eng.empty((NMAX,), np.uint8)
eng._call("sylow_hip_g2_generator_mul_batch", ka.ptr, pk.ptr, pki.ptr, NMAX)
eng._call("sylow_hip_bls_sign_batch", ka.ptr, dm.ptr, doff.ptr, sig.ptr, sigi.ptr, NMAX)
pkh, sigh = pk.download(), sig.download()
t0, rounds, quad_rounds, tail_rounds = time.time(), 0, 0, 0
while time.time() - t0 < budget:
    kind = rng.integers(0, 3)
    n = int(rng.integers(4097, 16385)) if kind == 0 else int(32768 * rng.integers(1, 4) + rng.integers(1, 16385)) if kind == 1 else int(rng.integers(16385, NMAX))
    n = min(n, NMAX)
    off = int(rng.integers(0, NMAX - n + 1))
    pinf = (rng.random(n) < 0.003).astype(np.uint8); qinf = (rng.random(n) < 0.003).astype(np.uint8)
    bad = np.flatnonzero(rng.random(n) < 0.002)
    s = np.ascontiguousarray(sigh[:, off:off + n]); s[:, bad] = s[:, (bad + 1) % n]
    sinf = (rng.random(n) < 0.001).astype(np.uint8)
    dp, dq = eng.empty((8, n)).upload(np.ascontiguousarray(ph[:, off:off + n])), eng.empty((16, n)).upload(np.ascontiguousarray(qh[:, off:off + n]))
    dpk, dsig = eng.empty((16, n)).upload(np.ascontiguousarray(pkh[:, off:off + n])), eng.empty((8, n)).upload(s)
    dpi, dqi, dsi = eng.to_device(pinf), eng.to_device(qinf), eng.to_device(sinf)
    dmm, doffm = eng.to_device(msgs[off:off + n].reshape(-1)), eng.to_device(np.arange(n + 1, dtype=np.uint64) * np.uint64(32))
    gt, ok = eng.empty((48, n)), eng.empty((n,), np.uint8)
    res = []
    for qm in (0, -1):
        eng.set_option("QUAD_MAX", qm)
        eng._call("sylow_hip_pairing_batch", dp.ptr, dpi.ptr, dq.ptr, dqi.ptr, gt.ptr, n)
        eng._call("sylow_hip_bls_verify_batch", dpk.ptr, None, dmm.ptr, doffm.ptr, dsig.ptr, dsi.ptr, ok.ptr, n)
        res.append((gt.download(), ok.download()))
    assert np.array_equal(res[0][0], res[1][0]), ("pairing differs", seed, rounds, n)
    assert np.array_equal(res[0][1], res[1][1]), ("verify differs", seed, rounds, n)
    want = np.ones(n, np.uint8); want[bad] = 0; want[sinf.astype(bool)] = 0
    assert np.array_equal(res[1][1], want), ("verify pattern", seed, rounds, n, np.flatnonzero(res[1][1] != want)[:8].tolist())
    ident = np.zeros(48, dtype=np.uint64); ident[0] = 1
    flagged = np.flatnonzero(pinf | qinf)
    assert all(np.array_equal(res[1][0][:, i], ident) for i in flagged[:64]), ("identity rule", seed, rounds, n)
    quad_rounds += n <= 16384; tail_rounds += (32768 < n < 3 * 32768 and 0 < n % 32768 <= 16384)
    rounds += 1
print("fuzz_mid ok: seed %s, %d rounds (%d on the quad route, %d with a quad tail), %.0f s" % (sys.argv[2] if len(sys.argv) > 2 else "1", rounds, quad_rounds, tail_rounds, time.time() - t0))
