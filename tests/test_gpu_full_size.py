"""BASELINE.json's full sizes through size-independent properties (bit-exact integer work):
C2a 2^20 Fp mul vs the oracle; C2b 2^20 G1 scalar-mul distributivity; C3 2^18 pairings bilinearity;
C4 2^20 BLS verifies with a planted corruption pattern; C5 2^16 ecPairing jobs (k = 2 and 4)."""
import numpy as np
import pytest

from helpers import P, fast_rand_fp_array, limbs, pack
from oracle import pyref as R

pytestmark = pytest.mark.gpu
G1 = [1, 2]
G2 = list(R.G2_GEN_AFF[0]) + list(R.G2_GEN_AFF[1])
R_ORDER = R.R_ORDER


def test_c2a_fp_mul_2_20_vs_oracle(engine, coracle):
    n = 1 << 20
    a, b = fast_rand_fp_array(101, n, 1), fast_rand_fp_array(102, n, 1)
    got = engine.fp_mul(a, b)
    assert np.array_equal(got, coracle.fp_op("mul", a, b))
    assert np.array_equal(engine.fp_add(a, b), coracle.fp_op("add", a, b))
    assert np.array_equal(engine.fp_sub(a, b), coracle.fp_op("sub", a, b))
    # odd batch size exercises the non-vector path
    assert np.array_equal(engine.fp_mul(a[:12345], b[:12345]), got[:12345])


def test_c2b_g1_scalar_mul_2_20(engine, coracle):
    n = 1 << 20
    g = np.repeat(pack(G1, 8), n, 0)
    a, b = fast_rand_fp_array(103, n, 1), fast_rand_fp_array(104, n, 1)
    a[:, 3] >>= np.uint64(2); b[:, 3] >>= np.uint64(2)
    ap, ainf = engine.g1_scalar_mul(g, a)
    bp, _ = engine.g1_scalar_mul(g, b)
    sp, _ = engine.g1_scalar_mul(g, engine.fp_add(a, b))
    sum_xy, sum_inf = engine.g1_add(ap, bp)
    assert np.array_equal(sum_xy, sp) and not sum_inf.any() and not ainf.any()
    # a sample against the oracle (affine)
    idx = np.random.default_rng(1).choice(n, 512, replace=False)
    one = np.zeros((512, 4), dtype=np.uint64); one[:, 0] = 1
    exp, _ = coracle.g1_to_affine(coracle.g1_scalar_mul(np.concatenate([g[idx], one], axis=1), a[idx]))
    assert np.array_equal(ap[idx], exp)


def test_c3_pairings_2_18_bilinearity(engine, coracle):
    n = 1 << 18
    g1, g2 = np.repeat(pack(G1, 8), n, 0), np.repeat(pack(G2, 16), n, 0)
    a, b = fast_rand_fp_array(105, n, 1), fast_rand_fp_array(106, n, 1)
    pa, _ = engine.g1_scalar_mul(g1, a)
    qb, _ = engine.g2_scalar_mul(g2, b)
    lhs = engine.pairing(pa, qb)                                       # e(aP, bQ)
    # e(aP, bQ) == e(bP', aQ')... use the swap: e(aG1, bG2) == e(bG1, aG2)
    pb, _ = engine.g1_scalar_mul(g1, b)
    qa, _ = engine.g2_scalar_mul(g2, a)
    assert np.array_equal(lhs, engine.pairing(pb, qa))
    # sample >= 4096 indices against the oracle
    idx = np.random.default_rng(2).choice(n, 4096, replace=False)
    one = np.zeros((4096, 4), dtype=np.uint64); one[:, 0] = 1
    p_proj = np.concatenate([pa[idx], one], axis=1)
    q_proj = np.concatenate([qb[idx], one, np.zeros((4096, 4), dtype=np.uint64)], axis=1)
    assert np.array_equal(lhs[idx], coracle.pairing(p_proj, q_proj))


def test_c4_bls_verify_2_20_planted(engine, coracle):
    n = 1 << 20
    g = np.random.default_rng(107)
    blob = g.integers(0, 256, size=(n, 32), dtype=np.uint8)
    off = np.arange(n + 1, dtype=np.uint64) * np.uint64(32)
    dm, doff = engine.to_device(blob.reshape(-1)), engine.to_device(off)
    sk_aos = fast_rand_fp_array(108, n, 1)
    sk = engine.to_device_soa(sk_aos, 4)
    g2 = engine.to_device_soa(np.repeat(pack(G2, 16), n, 0), 16)
    pk, pki = engine.empty((16, n)), engine.empty((n,), np.uint8)
    sig, sigi = engine.empty((8, n)), engine.empty((n,), np.uint8)
    engine._call("sylow_hip_g2_scalar_mul_batch", g2.ptr, None, sk.ptr, pk.ptr, pki.ptr, n)
    engine._call("sylow_hip_bls_sign_batch", sk.ptr, dm.ptr, doff.ptr, sig.ptr, sigi.ptr, n)
    # corrupt 1/1024 of the signatures: sig + G1gen at PRNG-chosen indices
    plant = g.random(n) < 1 / 1024
    sig_aos = engine.from_device_soa(sig)
    assert not sigi.download().any()
    # 256 PRNG-chosen signatures of the FULL-SIZE launch against the oracle's sign (lib.rs:179-187): the flags below are otherwise
    # judged against the GPU's own signatures only
    sidx = np.sort(np.random.default_rng(4).choice(n, 256, replace=False))
    smsgs = [blob[i].tobytes() for i in sidx]
    exp_sig, exp_inf = coracle.g1_to_affine(coracle.sign(sk_aos[sidx], smsgs))
    assert np.array_equal(sig_aos[sidx], exp_sig) and not exp_inf.any()
    bad, _ = engine.g1_add(sig_aos[plant], np.repeat(pack(G1, 8), int(plant.sum()), 0))
    sig_aos[plant] = bad
    sig2 = engine.to_device_soa(sig_aos, 8)
    ok = engine.empty((n,), np.uint8)
    engine._call("sylow_hip_bls_verify_batch", pk.ptr, None, dm.ptr, doff.ptr, sig2.ptr, None, ok.ptr, n)
    flags = ok.download().astype(bool)
    assert np.array_equal(flags, ~plant)
    assert engine.flags_all(ok) == 0
    # 256 flags of the full-size launch against the oracle's verify (two pairings + compare, lib.rs:223-236): 192 PRNG-chosen rows and
    # 64 of the planted ones
    vidx = np.concatenate([np.random.default_rng(5).choice(n, 192, replace=False), np.flatnonzero(plant)[:64]])
    pk_aos = engine.from_device_soa(pk)[vidx]
    one4 = np.zeros((vidx.size, 4), dtype=np.uint64); one4[:, 0] = 1
    pk_proj = np.concatenate([pk_aos, one4, np.zeros((vidx.size, 4), dtype=np.uint64)], axis=1)
    sig_proj = np.concatenate([sig_aos[vidx], one4], axis=1)
    assert not pki.download().any()
    assert np.array_equal(flags[vidx], coracle.verify(pk_proj, [blob[i].tobytes() for i in vidx], sig_proj).astype(bool))
    engine._call("sylow_hip_bls_verify_fused_batch", pk.ptr, None, dm.ptr, doff.ptr, sig2.ptr, None, ok.ptr, n)
    assert np.array_equal(ok.download().astype(bool), ~plant)
    engine._call("sylow_hip_bls_verify_fused_batch", pk.ptr, None, dm.ptr, doff.ptr, sig.ptr, None, ok.ptr, n)
    assert engine.flags_all(ok) == 1


@pytest.mark.parametrize("k", [2, 4])
def test_c5_ecpairing_2_16_jobs(engine, coracle, k):
    nj = 1 << 16
    n = nj * k
    g = np.random.default_rng(109 + k)
    a = [int(x) for x in g.integers(1, 1 << 62, size=n)]
    b = [int(x) for x in g.integers(1, 1 << 62, size=n)]
    # SURVEY d1 states uniform scalars: the 128 jobs sampled against the oracle below (and every 97th job besides) take scalars drawn from the
    # full range [1, r) -- the 62-bit draws of the other jobs only keep the host-side set-up of 2^16 k big-integer products short
    jidx = np.sort(np.random.default_rng(6 + k).choice(nj, 128, replace=False))
    full = sorted(set(jidx.tolist()) | set(range(0, nj, 97)))
    wide = np.random.default_rng(77 + k)
    for j in full:
        for i in range(k):
            a[j * k + i] = 1 + int.from_bytes(wide.bytes(32), "big") % (R_ORDER - 1)
            b[j * k + i] = 1 + int.from_bytes(wide.bytes(32), "big") % (R_ORDER - 1)
    # make each job multiply to one: last b chosen so that sum a_i b_i = 0 mod r; spoil every 5th job
    for j in range(nj):
        s = sum(a[j * k + i] * b[j * k + i] for i in range(k - 1)) % R_ORDER
        b[j * k + k - 1] = (-s) * pow(a[j * k + k - 1], -1, R_ORDER) % R_ORDER
        if j % 5 == 4:
            b[j * k + k - 1] = (b[j * k + k - 1] + 1) % R_ORDER
    p, _ = engine.g1_scalar_mul(np.repeat(pack(G1, 8), n, 0), limbs(a))
    q, _ = engine.g2_scalar_mul(np.repeat(pack(G2, 16), n, 0), limbs(b))
    gt, is_one = engine.multi_pairing(p, q, np.arange(nj + 1, dtype=np.uint64) * np.uint64(k), skip_infinity=True, want_gt=True)
    expect = np.array([0 if j % 5 == 4 else 1 for j in range(nj)], dtype=np.uint8)
    assert np.array_equal(is_one, expect)
    # 128 PRNG-chosen jobs of the full-size launch against the oracle's glued_pairing (pairing.rs:1029-1037), Gt values bit for bit
    # (about a fifth of them are the spoiled jobs, whose value is not one)
    rows = (jidx[:, None] * k + np.arange(k)[None, :]).reshape(-1)
    assert sum(a[r] >> 250 for r in rows) > 0 and sum(b[r] >> 250 for r in rows) > 0      # the sampled jobs carry full-range scalars
    one4 = np.zeros((rows.size, 4), dtype=np.uint64); one4[:, 0] = 1
    p_proj = np.concatenate([p[rows], one4], axis=1)
    q_proj = np.concatenate([q[rows], one4, np.zeros((rows.size, 4), dtype=np.uint64)], axis=1)
    exp_gt = coracle.glued_pairing(p_proj, q_proj, np.arange(129, dtype=np.uint64) * np.uint64(k))
    assert np.array_equal(gt[jidx], exp_gt)
    assert (jidx % 5 == 4).sum() >= 10
    # the same jobs from BYTES (BASELINE.json configs[4] as the precompile sees it: EIP-197 192-byte pairs, decode + curve / subgroup
    # checks + glued pairing), all 2^16 jobs in one call
    dp, dq = engine.to_device_soa(p, 8), engine.to_device_soa(q, 16)
    b1, b2 = engine.empty((n * 64,), np.uint8), engine.empty((n * 128,), np.uint8)
    engine._call("sylow_hip_g1_to_be_bytes_batch", dp.ptr, None, b1.ptr, n)
    engine._call("sylow_hip_g2_to_be_bytes_batch", dq.ptr, None, b2.ptr, n)
    blob = np.concatenate([b1.download().reshape(n, 64), b2.download().reshape(n, 128)], axis=1).reshape(-1)
    d_in, d_off = engine.to_device(blob), engine.to_device(np.arange(nj + 1, dtype=np.uint64) * np.uint64(k))
    d_res, d_st = engine.empty((nj,), np.uint8), engine.empty((nj,), np.uint8)
    engine._call("sylow_hip_evm_ecpairing_batch", d_in.ptr, d_off.ptr, nj, n, d_res.ptr, d_st.ptr)
    assert not d_st.download().any() and np.array_equal(d_res.download(), expect)


def test_c2c_g2_scalar_mul_2_18_split_equals_generic(engine, coracle):
    """The endomorphism-split G2 product on 2^18 r-torsion points: identical to the generic window product everywhere, a sample
    against the oracle, and (k1 + k2) Q == k1 Q + k2 Q through the group law."""
    n = 1 << 18
    g2 = np.repeat(pack(G2, 16), n, 0)
    a, b = fast_rand_fp_array(111, n, 1), fast_rand_fp_array(112, n, 1)
    a[:, 3] >>= np.uint64(2); b[:, 3] >>= np.uint64(2)
    q, _ = engine.g2_scalar_mul(g2, a, subgroup=True)                  # random r-torsion points
    s1, i1 = engine.g2_scalar_mul(q, b, subgroup=True)
    s0, i0 = engine.g2_scalar_mul(q, b)
    assert np.array_equal(s1, s0) and np.array_equal(i1, i0) and not i1.any()
    idx = np.random.default_rng(3).choice(n, 256, replace=False)
    one = np.zeros((256, 8), dtype=np.uint64); one[:, 0] = 1
    exp, _ = coracle.g2_to_affine(coracle.g2_scalar_mul(np.concatenate([q[idx], one], axis=1), b[idx]))
    assert np.array_equal(s1[idx], exp)
    ab, _ = engine.g2_scalar_mul(q, engine.fp_add(a, b), subgroup=True)
    aq, _ = engine.g2_scalar_mul(q, a, subgroup=True)
    sum_xy, sum_inf = engine.g2_add(aq, s1)
    assert np.array_equal(sum_xy, ab) and not sum_inf.any()


def test_staggered_launch_every_row(engine, coracle):
    """The skewed launch of k_pairing / k_bls_verify_fused (plk_pairing.hip: from 2^17 elements on, blocks 256..511 park their Miller values
    and extra blocks at the end of the grid finish them) on a batch with a ragged last block: 2^17 + 77 pairings built from 64 distinct pairs,
    so that EVERY row has an oracle value -- incl. identity flags planted inside the parked range and in the ragged tail -- and the same
    for verify with wrong signatures planted inside the parked range."""
    n, d = (1 << 17) + 77, 64
    rng = np.random.default_rng(515)
    a = limbs([int(x) for x in rng.integers(1, 1 << 62, size=d)])
    b = limbs([int(x) for x in rng.integers(1, 1 << 62, size=d)])
    p64, _ = engine.g1_scalar_mul(np.repeat(pack(G1, 8), d, 0), a)
    q64, _ = engine.g2_scalar_mul(np.repeat(pack(G2, 16), d, 0), b)
    one4 = np.zeros((d, 4), dtype=np.uint64); one4[:, 0] = 1
    gt64 = coracle.pairing(np.concatenate([p64, one4], axis=1), np.concatenate([q64, one4, np.zeros((d, 4), dtype=np.uint64)], axis=1))
    idx = np.arange(n) % d
    pinf, qinf = np.zeros(n, dtype=np.uint8), np.zeros(n, dtype=np.uint8)
    pinf[[5, 32768, 40000, 65535, n - 1]] = 1                     # elements 32768 .. 65535 are the parked chunks
    qinf[[6, 33333, 65000, 1 << 17]] = 1
    got = engine.pairing(p64[idx], q64[idx], p_inf=pinf, q_inf=qinf, pipelined=False)
    exp = gt64[idx]
    ident = np.zeros(48, dtype=np.uint64); ident[0] = 1
    exp[(pinf | qinf).astype(bool)] = ident
    assert np.array_equal(got, exp)
    # verify: 64 distinct (key, message, signature) triples, wrong signatures planted in and around the parked range
    sk = limbs([int(x) for x in rng.integers(1, 1 << 62, size=d)])
    msgs64 = [bytes([i, 255 - i, 7]) * (1 + i % 5) for i in range(d)]
    sig64, _ = engine.bls_sign(sk, msgs64)
    exp_sig, _ = coracle.g1_to_affine(coracle.sign(sk, msgs64))
    assert np.array_equal(sig64, exp_sig)
    pk64, _ = engine.g2_scalar_mul(np.repeat(pack(G2, 16), d, 0), sk)
    sig = sig64[idx].copy()
    bad = np.array([0, 32767, 32768, 32769, 50001, 65535, 65536, 100000, n - 1])
    sig[bad] = sig64[(idx[bad] + 1) % d]
    ok = engine.bls_verify(pk64[idx], [msgs64[i] for i in idx], sig, pipelined=False)        # one launch of n elements (the pipeline would cut it)
    want = np.ones(n, dtype=np.uint8); want[bad] = 0
    assert np.array_equal(ok, want)
