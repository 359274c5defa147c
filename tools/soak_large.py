"""Determinism soak of the LARGE-batch kernels (the ones the small sizes of tools/soak.py never reach: k_pairing, k_bls_verify_fused,
the line-table route of the multi-pair jobs, the byte-level ecPairing adapter): every call is repeated and every repetition must be
bit-identical to the first; the first result of each shape is checked on a few rows against the oracle / the planted pattern.
`python tools/soak_large.py [seconds] [log2n]`"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import sylow_amd
from bench import make_points, limbs_row, G2, SEED
from oracle import coracle as C

eng = sylow_amd.Engine(0)
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
L = int(sys.argv[2]) if len(sys.argv) > 2 else 16
n = 1 << L
p, q, ka, kb = make_points(eng, n, SEED + 77)
gt = eng.empty((48, n))
# pairing reference (first call) + oracle rows
eng._call("sylow_hip_pairing_batch", p.ptr, None, q.ptr, None, gt.ptr, n)
gt0 = gt.download()
idx = np.arange(0, n, n // 8)[:8]
ph, qh = p.download()[:, idx].T.copy(), q.download()[:, idx].T.copy()
one = np.zeros((8, 4), np.uint64); one[:, 0] = 1
exp = C.pairing(np.concatenate([ph, one], axis=1), np.concatenate([qh, one, np.zeros((8, 4), np.uint64)], axis=1))
assert np.array_equal(gt0[:, idx].T, exp), "pairing parity"
# verify inputs
msgs = np.random.default_rng(3).integers(0, 256, size=(n, 32), dtype=np.uint8)
dm, doff = eng.to_device(msgs.reshape(-1)), eng.to_device(np.arange(n + 1, dtype=np.uint64) * np.uint64(32))
g2 = eng.empty((16, n)).upload(np.repeat(limbs_row(G2).T, n, axis=1))
pk, pki, sig, sigi, ok = eng.empty((16, n)), eng.empty((n,), np.uint8), eng.empty((8, n)), eng.empty((n,), np.uint8), eng.empty((n,), np.uint8)
eng._call("sylow_hip_g2_scalar_mul_subgroup_batch", g2.ptr, None, ka.ptr, pk.ptr, pki.ptr, n)
eng._call("sylow_hip_bls_sign_batch", ka.ptr, dm.ptr, doff.ptr, sig.ptr, sigi.ptr, n)
sh = sig.download(); bad = np.arange(5, n, 1013); sh[:, bad] = sh[:, (bad + 1) % n]; sig.upload(sh)          # planted wrong signatures
eng._call("sylow_hip_bls_verify_batch", pk.ptr, None, dm.ptr, doff.ptr, sig.ptr, None, ok.ptr, n)
ok0 = ok.download()
want = np.ones(n, np.uint8); want[bad] = 0
assert np.array_equal(ok0, want), "verify pattern"
# ecPairing from bytes: nj jobs of k pairs, e(P,Q) e(-P,Q) [...]; every other job spoiled
nj = n // 4
ny = eng.empty((4, n))
eng._call("sylow_hip_fp_neg_batch", p.ptr + 4 * n * 8, ny.ptr, n)
pneg = eng.empty((8, n)).upload(np.concatenate([p.download()[:4], ny.download()], axis=0))
b1, b1n, b2 = eng.empty((n * 64,), np.uint8), eng.empty((n * 64,), np.uint8), eng.empty((n * 128,), np.uint8)
eng._call("sylow_hip_g1_to_be_bytes_batch", p.ptr, None, b1.ptr, n)
eng._call("sylow_hip_g1_to_be_bytes_batch", pneg.ptr, None, b1n.ptr, n)
eng._call("sylow_hip_g2_to_be_bytes_batch", q.ptr, None, b2.ptr, n)
g1b, g1nb, g2b = (x.download().reshape(n, -1)[:2 * nj] for x in (b1, b1n, b2))
pos, neg = np.concatenate([g1b, g2b], axis=1), np.concatenate([g1nb, g2b], axis=1)
shapes = {}
for k in (2, 4):
    jobs = np.concatenate([pos[:nj], neg[:nj]], axis=1) if k == 2 else np.concatenate([pos[0:2 * nj:2], neg[0:2 * nj:2], pos[1:2 * nj:2], neg[1:2 * nj:2]], axis=1)
    jobs = jobs.copy()
    spoil = np.arange(nj) % 2 == 1
    jobs[spoil, (k - 1) * 192:(k - 1) * 192 + 64] = g1b[(np.arange(nj)[spoil] + 7) % (2 * nj)]
    shapes[k] = (eng.to_device(jobs.reshape(-1)), eng.to_device(np.arange(nj + 1, dtype=np.uint64) * np.uint64(k)), (~spoil).astype(np.uint8))
d_res, d_st = eng.empty((nj,), np.uint8), eng.empty((nj,), np.uint8)
offk = {k: eng.to_device(np.arange(nj + 1, dtype=np.uint64) * np.uint64(k)) for k in (2, 3, 4)}
gtj, iso = eng.empty((48, nj)), eng.empty((nj,), np.uint8)
mp0 = {}
# host pipelines on PAGEABLE arrays (pipeline.hip: H2D / kernels / D2H on two streams, results through pageable hipMemcpyAsync + stream
# synchronise -- the copy pattern whose set-up twin once came back stale under rocprofv3 --pmc): every repetition against what the
# device-resident launches wrote
from sylow_amd import _lib
h_p, h_q = np.ascontiguousarray(p.download().T), np.ascontiguousarray(q.download().T)
h_pk, h_sig = np.ascontiguousarray(pk.download().T), np.ascontiguousarray(sig.download().T)
h_blob, h_off = np.ascontiguousarray(msgs.reshape(-1)), np.arange(n + 1, dtype=np.uint64) * np.uint64(32)
h_gt, h_ok = np.empty((n, 48), dtype=np.uint64), np.empty((n,), dtype=np.uint8)
gt0_aos = np.ascontiguousarray(gt0.T)
host_rounds = 0
t0 = time.time()
rounds = 0
while time.time() - t0 < budget:
    eng._call("sylow_hip_pairing_batch", p.ptr, None, q.ptr, None, gt.ptr, n)
    assert np.array_equal(gt.download(), gt0), ("pairing nondeterministic", rounds)
    eng._call("sylow_hip_bls_verify_batch", pk.ptr, None, dm.ptr, doff.ptr, sig.ptr, None, ok.ptr, n)
    assert np.array_equal(ok.download(), ok0), ("verify nondeterministic", rounds, np.flatnonzero(ok.download() != ok0)[:8].tolist())
    for k, (d_in, d_off, pattern) in shapes.items():
        eng._call("sylow_hip_evm_ecpairing_batch", d_in.ptr, d_off.ptr, nj, k * nj, d_res.ptr, d_st.ptr)
        r, s = d_res.download(), d_st.download()
        assert np.array_equal(r, pattern) and not s.any(), ("ecPairing pattern", k, rounds, np.flatnonzero(r != pattern)[:8].tolist(), np.flatnonzero(s)[:8].tolist())
    for k in (2, 3, 4):
        m = (n // k) if k == 3 else nj
        eng._call("sylow_hip_multi_pairing_batch", p.ptr, None, q.ptr, None, offk[k].ptr if k != 3 else eng.to_device(np.arange(m + 1, dtype=np.uint64) * np.uint64(3)).ptr,
                  m, k * m, 1, gtj.ptr if m == nj else eng.empty((48, m)).ptr, iso.ptr if m == nj else eng.empty((m,), np.uint8).ptr) if k != 3 else None
        if k != 3:
            g = gtj.download()
            if k not in mp0:
                mp0[k] = g
            assert np.array_equal(g, mp0[k]), ("multi_pairing nondeterministic", k, rounds)
    if rounds % 3 == 0:
        h_gt.fill(0); h_ok.fill(2)
        _lib.check(eng.lib.sylow_hip_pairing_host(h_p.ctypes.data, None, h_q.ctypes.data, None, h_gt.ctypes.data, n, 0), "pairing_host")
        assert np.array_equal(h_gt, gt0_aos), ("pairing_host differs from the device-resident result", rounds, np.flatnonzero((h_gt != gt0_aos).any(axis=1))[:8].tolist())
        _lib.check(eng.lib.sylow_hip_bls_verify_host(h_pk.ctypes.data, None, h_blob.ctypes.data, h_off.ctypes.data, h_sig.ctypes.data, None, h_ok.ctypes.data, n, 0), "bls_verify_host")
        assert np.array_equal(h_ok, ok0), ("bls_verify_host differs from the device-resident result", rounds, np.flatnonzero(h_ok != ok0)[:8].tolist())
        host_rounds += 1
    rounds += 1
print("soak_large ok: %d rounds (%d of them also through the pageable host pipelines) at n = 2^%d in %.0f s" % (rounds, host_rounds, L, time.time() - t0))
