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
