"""Mid-size batches on one lane QUAD per element (sylow_amd/csrc/plk_quad.hip: the lane-pair tower compiled with BN_QUAD 1, every pair of
independent product leaves split between an element's two lane pairs).  Same formulas and operand classes, hence the same digits: every row
of pairing_batch / miller_loop_batch / final_exp_batch on the quad route equals the lane-pair route's, and samples equal the oracle's
(pairing.rs:590-619, 245-492, 870-893).  Sizes straddle the route's bounds (one-wavefront cap 6144 < n <= 16384) and are ragged."""
import numpy as np
import pytest

from helpers import SEED, Xoshiro, limbs, pack
from test_gpu_multi_pairing import G1, G2

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def points(engine):
    rng = Xoshiro(SEED + 60)
    d = 96
    p, _ = engine.g1_scalar_mul(np.repeat(pack(G1, 8), d, 0), limbs([rng.fp() for _ in range(d)]))
    q, _ = engine.g2_scalar_mul(np.repeat(pack(G2, 16), d, 0), limbs([rng.fp() for _ in range(d)]))
    return p, q


def _both_routes(engine, fn):
    prev = engine.get_option("QUAD_MAX")
    try:
        engine.set_option("QUAD_MAX", 1 << 20)
        quad = fn()
        engine.set_option("QUAD_MAX", 0)
        pair = fn()
    finally:
        engine.set_option("QUAD_MAX", prev)
    return quad, pair


@pytest.mark.parametrize("n", [6145, 7001, 16384])
def test_pairing_quad_route_every_row(engine, coracle, points, n):
    p, q = points
    d = p.shape[0]
    idx = (np.arange(n) * 7 + 3) % d
    pinf, qinf = np.zeros(n, dtype=np.uint8), np.zeros(n, dtype=np.uint8)
    pinf[[0, 5, 6, 7, n - 1]] = 1                              # whole quads, single lanes' worth of elements, the ragged tail
    qinf[[1, 5, 4097, n - 2]] = 1
    dp, dq = engine.to_device_soa(p[idx], 8), engine.to_device_soa(q[idx], 16)
    dpi, dqi = engine.to_device(pinf), engine.to_device(qinf)
    dg = engine.empty((48, n))

    def run():
        dg.upload(np.zeros((48, n), dtype=np.uint64))
        engine._call("sylow_hip_pairing_batch", dp.ptr, dpi.ptr, dq.ptr, dqi.ptr, dg.ptr, n)
        return engine.from_device_soa(dg)

    quad, pair = _both_routes(engine, run)
    assert np.array_equal(quad, pair)
    one4 = np.zeros((d, 4), dtype=np.uint64); one4[:, 0] = 1
    gt = coracle.pairing(np.concatenate([p, one4], axis=1), np.concatenate([q, one4, np.zeros((d, 4), dtype=np.uint64)], axis=1))
    exp = gt[idx]                                               # element i pairs p[idx[i]] with q[idx[i]]
    ident = np.zeros(48, dtype=np.uint64); ident[0] = 1
    exp[(pinf | qinf).astype(bool)] = ident
    assert np.array_equal(quad, exp)


def test_raw_miller_and_final_exp_quad_route(engine, coracle, points):
    """The raw Miller value (the reference's curves, pairing.rs:590-619) and final_exponentiation on the quad route: every row against the
    lane-pair route, 64 rows against the oracle."""
    p, q = points
    d, n = p.shape[0], 6400
    idx = (np.arange(n) * 11 + 1) % d
    dp, dq = engine.to_device_soa(p[idx], 8), engine.to_device_soa(q[idx], 16)
    df, dg = engine.empty((48, n)), engine.empty((48, n))

    def run():
        engine._call("sylow_hip_miller_loop_batch", dp.ptr, dq.ptr, df.ptr, n)
        engine._call("sylow_hip_final_exp_batch", df.ptr, dg.ptr, n)
        return engine.from_device_soa(df), engine.from_device_soa(dg)

    (fq, gq), (fp_, gp) = _both_routes(engine, run)
    assert np.array_equal(fq, fp_) and np.array_equal(gq, gp)
    rows = np.arange(64) * 97 % n
    f_exp = coracle.miller_loop(p[idx[rows]], q[idx[rows]])
    assert np.array_equal(fq[rows], f_exp)
    assert np.array_equal(gq[rows], coracle.final_exponentiation(f_exp))


@pytest.mark.parametrize("same_signer", [False, True])
def test_verify_quad_route_every_row(engine, coracle, same_signer):
    """bls_verify_batch / bls_verify_same_signer_batch (lib.rs:223-236) on the quad route: every flag equals the lane-pair route's and the planted
    pattern (wrong signatures, identity signatures, identity keys), 48 rows against the oracle's verify."""
    from test_gpu_aggregate import signed_batch
    d, n = 48, 7001
    pk, msgs, sig = signed_batch(engine, d, same_signer=same_signer, seed=SEED + 61)
    idx = (np.arange(n) * 5 + 2) % d
    sigs = sig[idx].copy()
    bad = np.array([0, 1, 2, 3, 4, 5, 6, 7, 3000, 4097, n - 1])
    sigs[bad] = sig[(idx[bad] + 1) % d]                                  # a valid point, the wrong signature
    siginf = np.zeros(n, dtype=np.uint8); siginf[[9, 5000]] = 1          # identity signature: e(O, G2) = 1 != e(H, pk)
    blob = b"".join(msgs[i] for i in idx)
    off = np.zeros(n + 1, dtype=np.uint64); off[1:] = np.cumsum([len(msgs[i]) for i in idx])
    dm, doff = engine.to_device(np.frombuffer(blob, dtype=np.uint8)), engine.to_device(off)
    dsig, dsi = engine.to_device_soa(sigs, 8), engine.to_device(siginf)
    dpk = engine.to_device_soa(pk if same_signer else pk[idx], 16)
    ok = engine.empty((n,), np.uint8)
    name = "sylow_hip_bls_verify_same_signer_batch" if same_signer else "sylow_hip_bls_verify_batch"

    def run():
        ok.upload(np.full(n, 7, dtype=np.uint8))
        engine._call(name, dpk.ptr, None, dm.ptr, doff.ptr, dsig.ptr, dsi.ptr, ok.ptr, n)
        return ok.download()

    quad, pair = _both_routes(engine, run)
    assert np.array_equal(quad, pair)
    want = np.ones(n, dtype=np.uint8); want[bad] = 0; want[[9, 5000]] = 0
    if same_signer:
        want[bad] = 1                                                    # one key, one message per residue: the "wrong" signature of another row ...
        want[bad] = (np.array([msgs[idx[b]] == msgs[(idx[b] + 1) % d] for b in bad])).astype(np.uint8)   # ... is right only for an equal message
    assert np.array_equal(quad, want)
    rows = np.concatenate([bad, np.arange(100, 137)])
    one4 = np.zeros((rows.size, 4), dtype=np.uint64); one4[:, 0] = 1
    pk_rows = (np.repeat(pk, rows.size, 0) if same_signer else pk[idx[rows]])
    pk_proj = np.concatenate([pk_rows, one4, np.zeros((rows.size, 4), dtype=np.uint64)], axis=1)
    sig_proj = np.concatenate([sigs[rows], one4], axis=1)
    exp = coracle.verify(pk_proj, [msgs[idx[r]] for r in rows], sig_proj).astype(np.uint8)
    assert np.array_equal(quad[rows], exp)


@pytest.mark.parametrize("n", [32768 + 777, 65536 + 4099])
def test_tail_split_every_row(engine, coracle, points, n):
    """A batch of whole rounds (32768 elements each) plus a short tail: the tail runs on quads on a side stream beside the rounds' lane-pair grid
    (plk_pairing.hip, plk_verify.hip: tail_split).  Every row of pairing_batch and every flag of bls_verify_batch equals the single-launch
    route's (TAIL_SPLIT option 0); rows around the seam and in the tail against the oracle; identities and wrong signatures on both sides."""
    from test_gpu_aggregate import signed_batch
    p, q = points
    d = p.shape[0]
    idx = (np.arange(n) * 13 + 5) % d
    seam = (n // 32768) * 32768
    pinf, qinf = np.zeros(n, dtype=np.uint8), np.zeros(n, dtype=np.uint8)
    pinf[[3, seam - 1, seam, seam + 1, n - 1]] = 1
    qinf[[4, seam + 2, n - 2]] = 1
    dp, dq = engine.to_device_soa(p[idx], 8), engine.to_device_soa(q[idx], 16)
    dpi, dqi, dg = engine.to_device(pinf), engine.to_device(qinf), engine.empty((48, n))
    pk, msgs, sig = signed_batch(engine, 48, same_signer=False, seed=SEED + 62)
    vid = idx % 48
    sigs = sig[vid].copy()
    bad = np.array([0, seam - 2, seam - 1, seam, seam + 1, seam + 300, n - 1])
    sigs[bad] = sig[(vid[bad] + 1) % 48]
    blob = b"".join(msgs[i] for i in vid)
    off = np.zeros(n + 1, dtype=np.uint64); off[1:] = np.cumsum([len(msgs[i]) for i in vid])
    dm, doff = engine.to_device(np.frombuffer(blob, dtype=np.uint8)), engine.to_device(off)
    dsig, dpk, ok = engine.to_device_soa(sigs, 8), engine.to_device_soa(pk[vid], 16), engine.empty((n,), np.uint8)

    # one key for the whole batch (the line-table form of the fused check: the key's table is shared by the rounds and the tail)
    pk1, msgs1, sig1 = signed_batch(engine, 48, same_signer=True, seed=SEED + 63)
    sigs1 = sig1[vid].copy()
    sigs1[bad] = sig[vid[bad]]                                        # a signature under ANOTHER key
    blob1 = b"".join(msgs1[i] for i in vid)
    off1 = np.zeros(n + 1, dtype=np.uint64); off1[1:] = np.cumsum([len(msgs1[i]) for i in vid])
    dm1, doff1 = engine.to_device(np.frombuffer(blob1, dtype=np.uint8)), engine.to_device(off1)
    dsig1, dpk1, ok1 = engine.to_device_soa(sigs1, 8), engine.to_device_soa(pk1, 16), engine.empty((n,), np.uint8)

    def run():
        dg.upload(np.zeros((48, n), dtype=np.uint64)); ok.upload(np.full(n, 7, dtype=np.uint8)); ok1.upload(np.full(n, 7, dtype=np.uint8))
        engine._call("sylow_hip_pairing_batch", dp.ptr, dpi.ptr, dq.ptr, dqi.ptr, dg.ptr, n)
        engine._call("sylow_hip_bls_verify_batch", dpk.ptr, None, dm.ptr, doff.ptr, dsig.ptr, None, ok.ptr, n)
        engine._call("sylow_hip_bls_verify_same_signer_batch", dpk1.ptr, None, dm1.ptr, doff1.ptr, dsig1.ptr, None, ok1.ptr, n)
        return engine.from_device_soa(dg), np.stack([ok.download(), ok1.download()])

    prev = engine.get_option("TAIL_SPLIT")
    try:
        engine.set_option("TAIL_SPLIT", 1)
        g1, o1 = run()
        engine.set_option("TAIL_SPLIT", 0)
        g0, o0 = run()
    finally:
        engine.set_option("TAIL_SPLIT", prev)
    assert np.array_equal(g1, g0) and np.array_equal(o1, o0)
    want = np.ones(n, dtype=np.uint8); want[bad] = 0
    assert np.array_equal(o1[0], want) and np.array_equal(o1[1], want)
    rows = np.concatenate([np.arange(seam - 8, seam + 24), np.arange(n - 8, n), [3, 4]])
    one4 = np.zeros((d, 4), dtype=np.uint64); one4[:, 0] = 1
    gt = coracle.pairing(np.concatenate([p, one4], axis=1), np.concatenate([q, one4, np.zeros((d, 4), dtype=np.uint64)], axis=1))
    exp = gt[idx[rows]]
    ident = np.zeros(48, dtype=np.uint64); ident[0] = 1
    exp[(pinf | qinf)[rows].astype(bool)] = ident
    assert np.array_equal(g1[rows], exp)
