"""GPU parity: k-pair multi-Miller product with one final exponentiation (glued_pairing,
pairing.rs:970-1037; the ecPairing shape of examples/reth_bn128.rs:156-217) vs the oracle."""
import os
import subprocess
import sys

import numpy as np
import pytest

from helpers import P, SEED, Xoshiro, ints, limbs, pack
from oracle import pyref as R

pytestmark = pytest.mark.gpu
G1 = [1, 2]
G2 = list(R.G2_GEN_AFF[0]) + list(R.G2_GEN_AFF[1])
ONE4 = np.array([[1, 0, 0, 0]], dtype=np.uint64)


def proj1(xy, inf=None):
    z = np.repeat(ONE4, xy.shape[0], 0)
    if inf is not None:
        z = z * (1 - np.asarray(inf, dtype=np.uint64))[:, None]
    return np.concatenate([xy, z], axis=1)


def proj2(xy, inf=None):
    n = xy.shape[0]
    z = np.concatenate([np.repeat(ONE4, n, 0), np.zeros((n, 4), dtype=np.uint64)], axis=1)
    if inf is not None:
        z = z * (1 - np.asarray(inf, dtype=np.uint64))[:, None]
    return np.concatenate([xy, z], axis=1)


def test_eip197_vector(engine, kats):
    v = {e["line"]: bytes.fromhex(e["hex"]) for e in kats["eip_vectors_raw"]["hex_literals"]}
    inp = v[389]
    p, q = [], []
    for i in range(0, len(inp), 192):
        c = inp[i:i + 192]
        ax, ay = int.from_bytes(c[0:32], "big"), int.from_bytes(c[32:64], "big")
        bay, bax, bby, bbx = [int.from_bytes(c[64 + 32 * j:96 + 32 * j], "big") for j in range(4)]
        p += [ax, ay]
        q += [bax, bay, bbx, bby]
    p, q = pack(p, 8), pack(q, 16)
    assert engine.g2_subgroup_check(q).tolist() == [0, 0]
    gt, is_one = engine.multi_pairing(p, q, [0, 2])
    assert is_one.tolist() == [1]                                     # reth_bn128.rs:406 expects 1
    one = np.zeros((1, 48), dtype=np.uint64); one[0, 0] = 1
    assert np.array_equal(gt, one)
    # empty job -> identity (pairing.rs:1218-1219; reth_bn128.rs:445-457 "no input" -> true)
    gt, is_one = engine.multi_pairing(p, q, [0, 0, 2, 2])
    assert is_one.tolist() == [1, 1, 1]


def test_random_jobs_vs_oracle(engine, coracle):
    rng = Xoshiro(SEED + 40)
    ks = [1, 2, 3, 4, 5, 6, 9, 2, 4, 1]                                # crosses the KMAX = 4 chunk boundary
    n = sum(ks)
    off = np.concatenate([[0], np.cumsum(ks)]).astype(np.uint64)
    p, _ = engine.g1_scalar_mul(np.repeat(pack(G1, 8), n, 0), limbs([rng.fp() for _ in range(n)]))
    q, _ = engine.g2_scalar_mul(np.repeat(pack(G2, 16), n, 0), limbs([rng.fp() for _ in range(n)]))
    gt, is_one = engine.multi_pairing(p, q, off)
    exp = coracle.glued_pairing(proj1(p), proj2(q), off)
    assert np.array_equal(gt, exp) and not is_one.any()
    # product == product of single pairings (pairing.rs:1239-1241)
    single = engine.pairing(p, q)
    acc = single[0:1]
    assert np.array_equal(gt[0:1], acc)
    acc = engine.fp12_mul(single[1:2], single[2:3])
    assert np.array_equal(gt[1:2], acc)


def test_products_equal_to_one(engine):
    """jobs built so that sum a_ij * b_ij = 0 mod r multiply to one (BLS / Groth16 shapes, k = 2 and 4)."""
    rng = Xoshiro(SEED + 41)
    r = R.R_ORDER
    jobs, a_all, b_all = [], [], []
    for k in (2, 4, 2, 4, 2, 4, 3, 5):
        a = [rng.fp() % r for _ in range(k)]
        b = [rng.fp() % r for _ in range(k - 1)]
        s = sum(x * y for x, y in zip(a, b)) % r
        b.append((-s) * pow(a[-1], -1, r) % r)
        if len(jobs) % 4 == 3:
            b[-1] = (b[-1] + 1) % r                                    # spoil every 4th job
        a_all += a; b_all += b; jobs.append(k)
    n = len(a_all)
    off = np.concatenate([[0], np.cumsum(jobs)]).astype(np.uint64)
    p, _ = engine.g1_scalar_mul(np.repeat(pack(G1, 8), n, 0), limbs(a_all))
    q, _ = engine.g2_scalar_mul(np.repeat(pack(G2, 16), n, 0), limbs(b_all))
    _, is_one = engine.multi_pairing(p, q, off, want_gt=False)
    assert is_one.tolist() == [1, 1, 1, 0, 1, 1, 1, 0]


def test_infinity_semantics(engine, coracle):
    """replay mode reproduces the reference (P = inf neutral, Q = inf collapses the product
    to Gt(0)); skip mode drops identity pairs as EIP-197 requires."""
    rng = Xoshiro(SEED + 42)
    p, _ = engine.g1_scalar_mul(np.repeat(pack(G1, 8), 3, 0), limbs([rng.fp() for _ in range(3)]))
    q, _ = engine.g2_scalar_mul(np.repeat(pack(G2, 16), 3, 0), limbs([rng.fp() for _ in range(3)]))
    base = engine.pairing(p[:1], q[:1])
    # P = inf in pair 1 (canonical encoding (0,1))
    p2 = p.copy(); p2[1] = pack([0, 1], 8)[0]
    pinf = np.array([0, 1, 0], dtype=np.uint8)
    gt, _ = engine.multi_pairing(p2[:2], q[:2], [0, 2], p_inf=pinf[:2])
    assert np.array_equal(gt, base)
    assert np.array_equal(gt, coracle.glued_pairing(proj1(p2[:2], pinf[:2]), proj2(q[:2]), [0, 2]))
    # Q = inf in pair 1: reference defect -> all-zero "Gt"
    q2 = q.copy(); q2[1] = pack([0, 0, 1, 0], 16)[0]
    qinf = np.array([0, 1, 0], dtype=np.uint8)
    gt, is_one = engine.multi_pairing(p[:2], q2[:2], [0, 2], q_inf=qinf[:2])
    exp = coracle.glued_pairing(proj1(p[:2]), proj2(q2[:2], qinf[:2]), [0, 2])
    assert np.array_equal(gt, exp) and not gt.any() and is_one.tolist() == [0]
    # EIP-197 semantics: identity pairs are skipped
    gt, _ = engine.multi_pairing(p[:2], q2[:2], [0, 2], q_inf=qinf[:2], skip_infinity=True)
    assert np.array_equal(gt, base)
    gt, is_one = engine.multi_pairing(p2[:2], q2[:2], [0, 2], p_inf=np.array([1, 1], dtype=np.uint8), q_inf=qinf[:2], skip_infinity=True)
    assert is_one.tolist() == [1]


@pytest.mark.parametrize("n", [0, 1, 2, 4, 5, 37, 301])
def test_batch_wide_product_vs_oracle(engine, coracle, n):
    """sylow_hip_pairing_product_batch: the whole batch as ONE glued product (chunked Miller loops, product tree, one final
    exponentiation) equals the reference's sequential glued_pairing over the same pairs."""
    rng = Xoshiro(SEED + 60 + n)
    one = np.zeros((1, 48), dtype=np.uint64); one[0, 0] = 1
    if n == 0:
        gt, is_one = engine.pairing_product(np.zeros((0, 8), np.uint64), np.zeros((0, 16), np.uint64))
        assert is_one and np.array_equal(gt, one)
        return
    p, _ = engine.g1_scalar_mul(np.repeat(pack(G1, 8), n, 0), limbs([rng.fp() for _ in range(n)]))
    q, _ = engine.g2_scalar_mul(np.repeat(pack(G2, 16), n, 0), limbs([rng.fp() for _ in range(n)]))
    gt, is_one = engine.pairing_product(p, q)
    exp = coracle.glued_pairing(proj1(p), proj2(q), np.array([0, n], dtype=np.uint64))
    assert np.array_equal(gt, exp) and not is_one
    if n >= 4:
        # skip mode drops flagged pairs (EIP-197); replay mode with a G2 identity zeroes the product
        pinf = np.zeros(n, np.uint8); qinf = np.zeros(n, np.uint8); pinf[1] = 1; qinf[3] = 1
        keep = [i for i in range(n) if i not in (1, 3)]
        gt_s, _ = engine.pairing_product(p, q, p_inf=pinf, q_inf=qinf, skip_infinity=True)
        exp_s = coracle.glued_pairing(proj1(p[keep]), proj2(q[keep]), np.array([0, len(keep)], dtype=np.uint64))
        assert np.array_equal(gt_s, exp_s)
        q0 = q.copy(); q0[3] = pack([0, 0, 1, 0], 16)[0]
        gt_r, _ = engine.pairing_product(p, q0, q_inf=qinf)
        exp_r = coracle.glued_pairing(proj1(p), proj2(q0, qinf), np.array([0, n], dtype=np.uint64))
        assert np.array_equal(gt_r, exp_r)


def test_batch_wide_same_signer_verification(engine, coracle):
    """examples/verify_multiple_messages_same_signer.rs:41-60 at batch size 64: the 2n pairs (sig_i, G2gen), (-H_i, pk) as one
    product == identity; one forged signature flips the aggregate."""
    rng = Xoshiro(SEED + 70)
    n = 64
    sk = limbs([rng.fp() % R.R_ORDER] * n)
    msgs = [bytes([i, 7, 7]) for i in range(n)]
    h_xy, _ = engine.hash_to_g1(msgs)
    sig, _ = engine.g1_scalar_mul(h_xy, sk)
    pk, _ = engine.g2_scalar_mul(np.repeat(pack(G2, 16), n, 0), sk)
    neg_h = h_xy.copy()
    neg_h[:, 4:] = engine.fp_neg(h_xy[:, 4:])
    p = np.empty((2 * n, 8), np.uint64); p[0::2] = sig; p[1::2] = neg_h
    q = np.empty((2 * n, 16), np.uint64); q[0::2] = np.repeat(pack(G2, 16), n, 0); q[1::2] = pk
    _, ok = engine.pairing_product(p, q)
    assert ok
    p[10] = sig[11]
    _, ok = engine.pairing_product(p, q)
    assert not ok


def test_batch_wide_product_is_deterministic(engine, coracle):
    """Regression: the stream-ordered allocator on the default stream once made ~7 % of these calls return a wrong product
    (tools/dbg_prod.py); the entry point now uses a per-device grow-only workspace."""
    rng = Xoshiro(SEED + 75)
    n = 96
    p, _ = engine.g1_scalar_mul(np.repeat(pack(G1, 8), n, 0), limbs([rng.fp() for _ in range(n)]))
    q, _ = engine.g2_scalar_mul(np.repeat(pack(G2, 16), n, 0), limbs([rng.fp() for _ in range(n)]))
    exp = coracle.glued_pairing(proj1(p), proj2(q), np.array([0, n], dtype=np.uint64))
    for _ in range(25):
        gt, _ = engine.pairing_product(p, q)
        assert np.array_equal(gt, exp)


def test_glued_miller_loop_raw_value(engine, coracle):
    """glued_miller_loop (pairing.rs:970-1022): the raw value of a job is the product of its pairs' Miller values; final
    exponentiation of it is glued_pairing; an empty job yields one."""
    rng = Xoshiro(SEED + 80)
    ks = [1, 2, 3, 5, 0, 9]
    n = sum(ks)
    off = np.concatenate([[0], np.cumsum(ks)]).astype(np.uint64)
    p, _ = engine.g1_scalar_mul(np.repeat(pack(G1, 8), n, 0), limbs([rng.fp() for _ in range(n)]))
    q, _ = engine.g2_scalar_mul(np.repeat(pack(G2, 16), n, 0), limbs([rng.fp() for _ in range(n)]))
    f = engine.glued_miller_loop(p, q, off)
    per_pair = coracle.miller_loop(p, q)
    one = np.zeros((1, 48), dtype=np.uint64); one[0, 0] = 1
    for j, k in enumerate(ks):
        e = one
        for i in range(int(off[j]), int(off[j + 1])):
            e = coracle.fp12_op("mul", e, per_pair[i:i + 1])
        assert np.array_equal(f[j:j + 1], e), j
    assert np.array_equal(engine.final_exp(f), coracle.glued_pairing(proj1(p), proj2(q), off))


def test_ragged_jobs_against_per_pair_miller_values(engine):
    """A differential check that needs no CPU: 4096 jobs of 0..9 pairs (empty jobs, one-pair jobs and jobs longer than the table's
    slots in the same wavefronts).  The raw glued value of a job must equal the Fp12 product of its pairs' single-pair Miller values
    (miller_loop_batch: another kernel, no shared squarings, no tables), and the glued pairing its final exponentiation."""
    g = np.random.default_rng(SEED % (1 << 31))
    nj = 4096
    sizes = g.integers(0, 10, size=nj)
    sizes[:64] = g.permutation(np.repeat(np.arange(8), 8))            # one wavefront with every size
    off = np.concatenate([[0], np.cumsum(sizes)]).astype(np.uint64)
    n = int(off[-1])
    rng = Xoshiro(SEED + 77)
    ka, kb = limbs([rng.fp() for _ in range(n)]), limbs([rng.fp() for _ in range(n)])
    p, _ = engine.g1_scalar_mul(np.repeat(pack(G1, 8), n, 0), ka)
    q, _ = engine.g2_scalar_mul(np.repeat(pack(G2, 16), n, 0), kb, subgroup=True)
    f = engine.miller_loop(p, q)                                       # [n, 48]
    one = np.zeros((nj, 48), dtype=np.uint64); one[:, 0] = 1
    acc = one.copy()
    for s_ in range(int(sizes.max())):
        live = np.nonzero(sizes > s_)[0]
        acc[live] = engine.fp12_mul(acc[live], f[off[live].astype(np.int64) + s_])
    raw = engine.glued_miller_loop(p, q, off)
    assert np.array_equal(raw, acc)
    gt, is_one = engine.multi_pairing(p, q, off, skip_infinity=True)
    assert np.array_equal(gt, engine.final_exp(acc))
    assert np.array_equal(is_one.astype(bool), (sizes == 0))           # only the empty product is the identity here


def test_single_job_route_equals_the_generic_one(engine, coracle):
    """A batch of ONE job with skip_infinity (a single ecPairing call) takes a route of its own: every pair on its own lane pair, a
    product tree, the final exponentiation on a whole wavefront.  Its job may be any sub-range of the pair arrays, empty included, with
    identities inside: same Gt and flag as the same job inside a two-job batch (the generic kernels) and as the oracle."""
    rng = Xoshiro(SEED + 77)
    n = 14
    p, _ = engine.g1_scalar_mul(np.repeat(pack(G1, 8), n, 0), limbs([rng.fp() for _ in range(n)]))
    q, _ = engine.g2_scalar_mul(np.repeat(pack(G2, 16), n, 0), limbs([rng.fp() for _ in range(n)]))
    pinf = np.zeros(n, np.uint8); qinf = np.zeros(n, np.uint8)
    pinf[4] = 1; qinf[7] = 1
    one = np.zeros((1, 48), np.uint64); one[0, 0] = 1
    for lo, hi in ((0, 1), (2, 3), (2, 4), (3, 6), (1, 9), (0, 14), (5, 5), (4, 5), (7, 8)):
        g1, o1 = engine.multi_pairing(p, q, [lo, hi], p_inf=pinf, q_inf=qinf, skip_infinity=True)
        g2, o2 = engine.multi_pairing(p, q, [lo, hi, hi], p_inf=pinf, q_inf=qinf, skip_infinity=True)
        assert np.array_equal(g1, g2[:1]) and o1[0] == o2[0], (lo, hi)
        live = [i for i in range(lo, hi) if not (pinf[i] or qinf[i])]
        if live:
            exp = coracle.glued_pairing(proj1(p[live]), proj2(q[live]), np.array([0, len(live)], dtype=np.uint64))
            assert np.array_equal(g1, exp), (lo, hi)
        else:
            assert np.array_equal(g1, one) and o1[0] == 1, (lo, hi)
    # e(aP, Q) e(-aP, Q) = 1 through the single-job route
    pp = np.concatenate([p[:1], p[:1]])
    pp[1, 4:] = engine.fp_neg(p[:1, 4:])[0]
    _, o = engine.multi_pairing(pp, np.concatenate([q[:1], q[:1]]), [0, 2], skip_infinity=True)
    assert o[0] == 1
