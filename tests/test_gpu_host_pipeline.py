"""The value-typed entry points (host arrays in, host results out: sylow_hip_pairing_host / sylow_hip_bls_verify_host,
pairing.rs:870-893, lib.rs:223-236): the chunked, double-buffered pipeline must return exactly what upload -> device call ->
download returns -- ragged last chunks, chunks smaller and larger than the wide-route thresholds, identity flags, pageable and
pinned host memory -- and what the oracle computes."""
import numpy as np
import pytest

from helpers import SEED, Xoshiro, limbs, pack

pytestmark = pytest.mark.gpu

G1 = [1, 2]
G2 = [0x1800DEEF121F1E76426A00665E5C4479674322D4F75EDADD46DEBD5CD992F6ED,
      0x198E9393920D483A7260BFB731FB5D25F1AA493335A9E71297E485B7AEF312C2,
      0x12C85EA5DB8C6DEB4AAB71808DCB408FE3D1E7690C43D37B4CE6CC0166FA7DAA,
      0x090689D0585FF075EC9E99AD690C3395BC4B313370B38EF355ACDADCD122975B]


def points(engine, seed, n):
    ka, kb = engine.xoshiro_fp_soa(seed, n).T.copy(), engine.xoshiro_fp_soa(seed + 77, n).T.copy()
    p_xy, _ = engine.g1_scalar_mul(np.tile(pack(G1, 8), (n, 1)), ka)
    q_xy, _ = engine.g2_scalar_mul(np.tile(pack(G2, 16), (n, 1)), kb)
    return ka, p_xy, q_xy


@pytest.mark.parametrize("n,chunk", [(1, 0), (37, 5), (1037, 100), (4500, 2100), (5000, 0)])
def test_pairing_host_equals_device_call(engine, n, chunk):
    _, p_xy, q_xy = points(engine, SEED + 40 + n, n)
    g = np.random.default_rng(n)
    p_inf, q_inf = (g.random(n) < 0.05).astype(np.uint8), (g.random(n) < 0.05).astype(np.uint8)
    ref = engine.pairing(p_xy, q_xy, p_inf, q_inf, pipelined=False)
    got = engine.pairing(p_xy, q_xy, p_inf, q_inf, chunk=chunk)
    assert np.array_equal(got, ref)
    assert np.array_equal(engine.pairing(p_xy, q_xy, chunk=chunk), engine.pairing(p_xy, q_xy, pipelined=False))     # NULL flag arrays


def test_pairing_host_vs_oracle_and_pinned(engine, coracle):
    n = 300
    _, p_xy, q_xy = points(engine, SEED + 41, n)
    one = np.zeros((n, 4), dtype=np.uint64); one[:, 0] = 1
    exp = coracle.pairing(np.concatenate([p_xy, one], axis=1), np.concatenate([q_xy, one, np.zeros((n, 4), dtype=np.uint64)], axis=1))
    hp, hq, out = engine.pinned_empty((n, 8)), engine.pinned_empty((n, 16)), engine.pinned_empty((n, 48))
    hp[:], hq[:] = p_xy, q_xy
    got = engine.pairing(hp, hq, chunk=64, out=out)
    assert got is out and np.array_equal(out, exp)
    assert np.array_equal(engine.pairing(p_xy, q_xy, chunk=64), exp)


@pytest.mark.parametrize("n,chunk", [(1, 0), (29, 4), (1200, 500), (3000, 0)])
def test_verify_host_equals_device_call(engine, n, chunk):
    rng = np.random.default_rng(1000 + n)
    sk = engine.xoshiro_fp_soa(SEED + 50 + n, n).T.copy()
    lens = rng.integers(0, 200, size=n)
    msgs = [rng.integers(0, 256, size=int(l), dtype=np.uint8).tobytes() for l in lens]
    sig_xy, sig_inf = engine.bls_sign(sk, msgs)
    pk_xy, pk_inf = engine.g2_scalar_mul(np.tile(pack(G2, 16), (n, 1)), sk)
    bad = rng.random(n) < 0.1
    sig_bad = sig_xy.copy()
    sig_bad[bad] = np.roll(sig_xy, 1, axis=0)[bad]                      # a valid point, the wrong signature (n = 1: its own: stays valid)
    flags_pk, flags_sig = (rng.random(n) < 0.03).astype(np.uint8), (rng.random(n) < 0.03).astype(np.uint8)
    for pk_f, sig_f in ((None, None), (flags_pk, flags_sig)):
        ref = engine.bls_verify(pk_xy, msgs, sig_bad, pk_f, sig_f, pipelined=False)
        got = engine.bls_verify(pk_xy, msgs, sig_bad, pk_f, sig_f, chunk=chunk)
        assert np.array_equal(got, ref)
    ok = engine.bls_verify(pk_xy, msgs, sig_bad, chunk=chunk)
    if n > 1:
        assert np.array_equal(ok.astype(bool), ~bad)
    assert engine.bls_verify(pk_xy, msgs, sig_xy, chunk=chunk).all()


def test_host_calls_empty_batch(engine):
    assert engine.pairing(np.zeros((0, 8), np.uint64), np.zeros((0, 16), np.uint64)).shape == (0, 48)
    assert engine.bls_verify(np.zeros((0, 16), np.uint64), [], np.zeros((0, 8), np.uint64)).shape == (0,)


def test_wire_format_pipelines(engine, coracle):
    """sylow_hip_pairing_host_bytes / sylow_hip_bls_verify_host_bytes: G1Affine / G2Affine::to_be_bytes blobs (g1.rs:151-180, g2.rs:319-359) in,
    decoded + validated on the device inside the pipeline.  Valid input == the word-level calls; an off-curve G1 point, a G2 point outside the
    r-torsion and a non-canonical coordinate are reported per element with the reference's GroupError codes and enter as the identity."""
    n = 700
    sk, p_xy, q_xy = points(engine, SEED + 60, n)
    pb, qb = engine.g1_to_be_bytes(p_xy), engine.g2_to_be_bytes(q_xy)
    gt, sp, sq = engine.pairing_from_bytes(pb, qb, chunk=128)
    assert not sp.any() and not sq.any()
    assert np.array_equal(gt, engine.pairing(p_xy, q_xy, pipelined=False))
    # planted failures
    pb2, qb2 = list(pb), list(qb)
    pb2[5] = pb[5][:63] + bytes([pb[5][63] ^ 1])                                   # y off by one: not on the curve
    pb2[6] = b"\x30\x64\x4e\x72\xe1\x31\xa0\x29\xb8\x50\x45\xb6\x81\x81\x58\x5d\x97\x81\x6a\x91\x68\x71\xca\x8d\x3c\x20\x8c\x16\xd8\x7c\xfd\x47" + pb[6][32:]   # x = p: not canonical
    # a twist point outside G2: x = 1 is on the twist for some y?  take a valid point and add a point of the cofactor part instead: simplest
    # reliable construction is the test vector the subgroup-check test uses -- reuse the engine's own check to find one by scanning small x
    from helpers import fp2_sqrt, P as PRIME
    from oracle import pyref as R
    off_sub = None
    for x0 in range(1, 200):
        rhs = R.fp2_add(R.fp2_mul(R.fp2_square((x0, 0)), (x0, 0)), R.TWIST_B)
        y = fp2_sqrt(rhs)
        if y is not None:
            cand = pack([x0, 0, y[0], y[1]], 16)
            if engine.g2_subgroup_check(cand)[0] == 2:
                off_sub = cand
                break
    assert off_sub is not None
    qb2[9] = engine.g2_to_be_bytes(off_sub)[0]
    gt2, sp2, sq2 = engine.pairing_from_bytes(pb2, qb2, chunk=128)
    assert sp2[5] == 1 and sp2[6] == 4 and sq2[9] == 2 and sp2.sum() == 5 and sq2.sum() == 2       # NOT_ON_CURVE, DECODE_ERROR, NOT_IN_SUBGROUP
    one = np.zeros(48, dtype=np.uint64); one[0] = 1
    for i in (5, 6, 9):
        assert np.array_equal(gt2[i], one)
    keep = np.ones(n, dtype=bool); keep[[5, 6, 9]] = False
    assert np.array_equal(gt2[keep], gt[keep])
    # verify from bytes
    msgs = [bytes([i & 255]) * (i % 40) for i in range(n)]
    sig_xy, sig_inf = engine.bls_sign(sk, msgs)
    pk_xy, pk_inf = engine.g2_scalar_mul(np.tile(pack(G2, 16), (n, 1)), sk)
    ok, s1, s2 = engine.bls_verify_from_bytes(engine.g2_to_be_bytes(pk_xy), msgs, engine.g1_to_be_bytes(sig_xy), chunk=300)
    assert ok.all() and not s1.any() and not s2.any()
    sb = engine.g1_to_be_bytes(sig_xy)
    sb[3], sb[4] = sb[4], sb[3]
    ok2, _, _ = engine.bls_verify_from_bytes(engine.g2_to_be_bytes(pk_xy), msgs, sb)
    assert not ok2[3] and not ok2[4] and ok2.sum() == n - 2
    # a REJECTED key next to an all-zero (identity) signature: e(O, g2) e(-H(m), O) = 1 satisfies the pairing equation, but the reference
    # never reaches verify for such a key (from_be_bytes / G2Projective::new return Err) -- ok must be 0 for every rejected blob
    kb = engine.g2_to_be_bytes(pk_xy)
    kb2, sb2 = list(kb), list(engine.g1_to_be_bytes(sig_xy))
    ident_sig = bytes(64)                                       # to_be_bytes_scrubbed of the identity (g1.rs:151-180, EVM convention)
    kb2[7] = bytes([0xFF]) * 128;            sb2[7] = ident_sig   # garbage key (non-canonical coordinates)
    kb2[8] = engine.g2_to_be_bytes(off_sub)[0]; sb2[8] = ident_sig   # key outside the r-torsion
    kb2[11] = kb[11][:127] + bytes([kb[11][127] ^ 1]); sb2[11] = ident_sig   # key off the curve
    sb2[12] = sb[12][:63] + bytes([sb[12][63] ^ 1])              # rejected signature under a good key
    ok3, k3, s3 = engine.bls_verify_from_bytes(kb2, msgs, sb2, chunk=128)
    assert k3[7] != 0 and k3[8] == 2 and k3[11] == 1 and s3[12] == 1
    assert not ok3[[7, 8, 11, 12]].any() and ok3.sum() == n - 4
    # offsets out of order INSIDE a chunk are refused on the host
    off = np.zeros(n + 1, dtype=np.uint64); off[1:] = np.cumsum([len(m) for m in msgs]); off[50], off[51] = off[51], off[50]
    blob = np.frombuffer(b"".join(msgs), dtype=np.uint8)
    kk, ss = np.frombuffer(b"".join(kb), dtype=np.uint8), np.frombuffer(b"".join(sb2), dtype=np.uint8)
    o, a, b = (np.empty(n, dtype=np.uint8) for _ in range(3))
    rc = engine.lib.sylow_hip_bls_verify_host_bytes(kk.ctypes.data, blob.ctypes.data, off.ctypes.data, ss.ctypes.data, o.ctypes.data, a.ctypes.data, b.ctypes.data, n, 0)
    assert rc != 0
