"""GPU parity for the small tower / group items SURVEY.md rows a9, a10, a14, a16, a17, a19 name, each through an entry point of
its own against the C oracle: Fp2 / Fp6 residue_mul and frobenius (fp2.rs:99-133, fp6.rs:189-211), Fp6::square (fp6.rs:213-236),
projective Sub (group.rs:614-624), G1Projective::new / G2Projective::new on [x, y, z] (g1.rs:383-402, g2.rs:460-525) and the
projective ct_eq (group.rs:426-447)."""
import numpy as np
import pytest

from helpers import P, SEED, Xoshiro, fp2_sqrt, limbs, pack, rand_fp_array
from oracle import pyref as R

pytestmark = pytest.mark.gpu

G1 = [1, 2]
G2 = list(R.G2_GEN_AFF[0]) + list(R.G2_GEN_AFF[1])
ONE4 = np.array([[1, 0, 0, 0]], dtype=np.uint64)


def edge_rows(width_fp):
    """all-zero, one, p - 1 everywhere, a lone top coefficient"""
    rows = [[0] * width_fp, [1] + [0] * (width_fp - 1), [P - 1] * width_fp, [0] * (width_fp - 1) + [P - 1]]
    return np.concatenate([limbs(r).reshape(1, 4 * width_fp) for r in rows], axis=0)


def test_fp2_fp6_small_items_vs_oracle(engine, coracle):
    rng = Xoshiro(SEED + 300)
    a2 = np.concatenate([edge_rows(2), rand_fp_array(rng, 60, 2)], axis=0)
    a6 = np.concatenate([edge_rows(6), rand_fp_array(rng, 60, 6)], axis=0)
    assert np.array_equal(engine.fp2_residue_mul(a2), coracle.fp2_op("mul_xi", a2))
    assert np.array_equal(engine.fp6_residue_mul(a6), coracle.fp6_residue_mul(a6))
    assert np.array_equal(engine.fp6_sqr(a6), coracle.fp6_op("sqr", a6))
    assert np.array_equal(engine.fp6_sqr(a6), coracle.fp6_op("mul", a6, a6))           # square() == a * a (fp6.rs tests)
    for e in (0, 1, 2, 3, 7):
        assert np.array_equal(engine.fp2_frobenius(a2, e), coracle.fp2_frobenius(a2, e)), e
    for e in (0, 1, 2, 3, 4, 5, 6, 11):
        assert np.array_equal(engine.fp6_frobenius(a6, e), coracle.fp6_frobenius(a6, e)), e
    # the automorphism has order 6 on Fp6 and 2 on Fp2
    x = a6
    for _ in range(6):
        x = engine.fp6_frobenius(x, 1)
    assert np.array_equal(x, a6)
    assert np.array_equal(engine.fp2_frobenius(engine.fp2_frobenius(a2, 1), 1), a2)


def _points(engine, rng, n):
    k = limbs([rng.fp() for _ in range(2 * n)])
    p_xy, _ = engine.g1_scalar_mul(np.repeat(pack(G1, 8), n, 0), k[:n])
    q_xy, _ = engine.g2_scalar_mul(np.repeat(pack(G2, 16), n, 0), k[n:], subgroup=True)
    return p_xy, q_xy


def _g1_proj(xy, inf, scale):
    """(x s, y s, s) for a non-zero s; identity -> (0, 1, 0)"""
    n = xy.shape[0]
    out = np.zeros((n, 12), dtype=np.uint64)
    for i in range(n):
        if inf[i]:
            out[i] = limbs([0, 1, 0]).reshape(-1)
        else:
            x, y = [sum(int(xy[i, 4 * c + k]) << (64 * k) for k in range(4)) for c in range(2)]
            out[i] = limbs([x * scale[i] % P, y * scale[i] % P, scale[i] % P]).reshape(-1)
    return out


def _g2_proj(coracle, xy, inf, scale):
    """(x s, y s, s) with s in Fp2"""
    n = xy.shape[0]
    s = limbs([v for pair in scale for v in pair]).reshape(n, 8)
    x = coracle.fp2_op("mul", xy[:, :8], s)
    y = coracle.fp2_op("mul", xy[:, 8:], s)
    out = np.concatenate([x, y, s], axis=1)
    for i in range(n):
        if inf[i]:
            out[i] = limbs([0, 0, 1, 0, 0, 0]).reshape(-1)
    return out


def test_group_sub_vs_oracle(engine, coracle):
    rng = Xoshiro(SEED + 301)
    n = 24
    p_xy, q_xy = _points(engine, rng, 2 * n)
    a1, b1, a2, b2 = p_xy[:n].copy(), p_xy[n:].copy(), q_xy[:n].copy(), q_xy[n:].copy()
    b1[0], b2[0] = a1[0], a2[0]                                  # P - P = identity
    ainf = np.zeros(n, dtype=np.uint8); binf = np.zeros(n, dtype=np.uint8)
    ainf[1] = 1; binf[2] = 1; ainf[3] = binf[3] = 1              # identity on either / both sides, spelled (0, 1, inf) as the
    for arr, flags, ident in ((a1, ainf, [0, 1]), (b1, binf, [0, 1]), (a2, ainf, [0, 0, 1, 0]), (b2, binf, [0, 0, 1, 0])):   # reference's affine form does
        arr[flags == 1] = limbs(ident).reshape(-1)
    z4 = np.zeros((n, 4), dtype=np.uint64)
    one1 = np.repeat(ONE4, n, 0)
    pa = np.concatenate([a1, one1 * (1 - ainf.astype(np.uint64))[:, None]], axis=1)
    pb = np.concatenate([b1, one1 * (1 - binf.astype(np.uint64))[:, None]], axis=1)
    got_xy, got_inf = engine.g1_sub(a1, b1, ainf, binf)
    exp_xy, exp_inf = coracle.g1_to_affine(coracle.g1_sub(pa, pb))
    assert np.array_equal(got_inf, exp_inf) and np.array_equal(got_xy, exp_xy) and got_inf[0] == 1 and got_inf[3] == 1
    qa = np.concatenate([a2, one1 * (1 - ainf.astype(np.uint64))[:, None], z4], axis=1)
    qb = np.concatenate([b2, one1 * (1 - binf.astype(np.uint64))[:, None], z4], axis=1)
    got_xy, got_inf = engine.g2_sub(a2, b2, ainf, binf)
    exp_xy, exp_inf = coracle.g2_to_affine(coracle.g2_sub(qa, qb))
    assert np.array_equal(got_inf, exp_inf) and np.array_equal(got_xy, exp_xy) and got_inf[0] == 1 and got_inf[3] == 1
    # a - b == a + (-b) through the existing add entry point
    nb = b1.copy(); nb[:, 4:] = coracle.fp_op("neg", b1[:, 4:])
    add_xy, add_inf = engine.g1_add(a1, nb, ainf, binf)
    sub_xy, sub_inf = engine.g1_sub(a1, b1, ainf, binf)
    assert np.array_equal(add_xy, sub_xy) and np.array_equal(add_inf, sub_inf)


def test_projective_new_vs_oracle(engine, coracle):
    rng = Xoshiro(SEED + 302)
    n = 16
    p_xy, q_xy = _points(engine, rng, n)
    inf = np.zeros(n, dtype=np.uint8); inf[5] = 1
    a = _g1_proj(p_xy, inf, [rng.fp() or 1 for _ in range(n)])
    a[6] = limbs([3, 7, 1]).reshape(-1)                          # off the curve
    a[7] = limbs([3, 7, 0]).reshape(-1)                          # Z = 0 passes whatever X, Y are (g1.rs:391)
    got, exp = engine.g1_projective_new(a), coracle.g1_projective_new(a)
    assert np.array_equal(got, exp) and got[6] == 1 and got[7] == 0 and not got[:6].any()
    b = _g2_proj(coracle, q_xy, inf, [(rng.fp() or 1, rng.fp()) for _ in range(n)])
    # a twist point outside the r-torsion: x from a counter, y by square root in Fp2 (pyref), in projective form
    y = None
    while y is None:
        x = (rng.fp(), rng.fp())
        y = fp2_sqrt(R.fp2_add(R.fp2_mul(R.fp2_square(x), x), R.TWIST_B))
    b[8] = _g2_proj(coracle, pack(list(x) + list(y), 16), [0], [(rng.fp() or 1, rng.fp())])[0]
    b[9] = limbs([1, 2, 3, 4, 1, 0]).reshape(-1)                 # off the curve
    b[10] = limbs([1, 2, 3, 4, 0, 0]).reshape(-1)                # Z = 0 with X, Y != 0: passes the curve test, fails the torsion test
    b[11] = limbs([0, 0, 5, 6, 0, 0]).reshape(-1)                # Z = 0, X = 0: every intermediate keeps Z = 0 -> Ok
    b[12] = limbs([1, 2, 0, 0, 0, 0]).reshape(-1)                # Z = 0, Y = 0 -> Ok
    b[13] = limbs([0, 0, 0, 0, 0, 0]).reshape(-1)
    got, exp = engine.g2_projective_new(b), coracle.g2_projective_new(b)
    exp = np.where(exp == 3, 1, exp).astype(np.uint8)            # the reference panics there: NOT_ON_CURVE here (documented)
    assert np.array_equal(got, exp)
    assert got[9] == 1 and got[10] == 2 and not got[:8].any() and not got[11:14].any()
    assert got[8] == 2                                           # on the twist, outside the r-torsion


def test_projective_ct_eq_vs_oracle(engine, coracle):
    rng = Xoshiro(SEED + 303)
    n = 12
    p_xy, q_xy = _points(engine, rng, n)
    inf = np.zeros(n, dtype=np.uint8); inf[4] = 1
    a1 = _g1_proj(p_xy, inf, [rng.fp() or 1 for _ in range(n)])
    b1 = _g1_proj(p_xy, inf, [rng.fp() or 1 for _ in range(n)])          # the same points under other representatives
    b1[1] = b1[2]                                                          # a different point
    b1[3] = limbs([0, 1, 0]).reshape(-1)                                   # identity vs finite
    b1[4] = limbs([5, 9, 0]).reshape(-1)                                   # two spellings of the identity
    got, exp = engine.g1_ct_eq(a1, b1), coracle.g1_ct_eq(a1, b1)
    assert np.array_equal(got, exp) and list(got[:5]) == [1, 0, 1, 0, 1]
    a2 = _g2_proj(coracle, q_xy, inf, [(rng.fp() or 1, rng.fp()) for _ in range(n)])
    b2 = _g2_proj(coracle, q_xy, inf, [(rng.fp() or 1, rng.fp()) for _ in range(n)])
    b2[1] = b2[2]
    b2[3] = limbs([0, 0, 1, 0, 0, 0]).reshape(-1)
    b2[4] = limbs([7, 1, 2, 3, 0, 0]).reshape(-1)
    got, exp = engine.g2_ct_eq(a2, b2), coracle.g2_ct_eq(a2, b2)
    assert np.array_equal(got, exp) and list(got[:5]) == [1, 0, 1, 0, 1]


@pytest.mark.parametrize("n", [0, 1, 2, 17, 513, 600, 5000, 70000])
def test_g1_sum_vs_oracle(engine, coracle, n):
    """sylow_hip_g1_sum_batch: the fold of Add for G1Projective (group.rs:528-599) over a batch -- staged serial accumulation on the device
    against a left-to-right fold by the oracle's own addition; identity flags inside, and P + (-P) in the batch."""
    from helpers import P as PRIME
    k = engine.xoshiro_fp_soa(SEED + 600 + n, max(n, 1)).T.copy()[:n]
    pts, _ = engine.g1_scalar_mul(np.tile(pack([1, 2], 8), (n, 1)), k) if n else (np.zeros((0, 8), np.uint64), None)
    flags = (np.random.default_rng(n).random(n) < 0.1).astype(np.uint8)
    if n >= 4:                                        # a point and its negative: the complete formulas must pass through the identity
        neg = pts[2].copy()
        y = sum(int(neg[4 + j]) << (64 * j) for j in range(4))
        neg[4:8] = limbs([(PRIME - y) % PRIME])[0]
        pts[3] = neg
        flags[2] = flags[3] = 0
    got_xy, got_inf = engine.g1_sum(pts, flags)
    one = np.zeros((1, 4), dtype=np.uint64); one[0, 0] = 1
    acc = np.concatenate([np.zeros((1, 4), np.uint64), one, np.zeros((1, 4), np.uint64)], axis=1)      # (0 : 1 : 0)
    live = pts[flags == 0]
    # fold in chunks: pairwise tree on the host side of the ORACLE's addition (associativity is the group law's; the affine result is unique)
    cur = np.concatenate([live, np.tile(one, (live.shape[0], 1))], axis=1) if live.shape[0] else np.zeros((0, 12), np.uint64)
    while cur.shape[0] > 1:
        if cur.shape[0] % 2:
            cur = np.concatenate([cur, acc], axis=0)
        cur = coracle.g1_add(cur[0::2], cur[1::2])
    total = cur if cur.shape[0] else acc
    exp_xy, exp_inf = coracle.g1_to_affine(total)
    assert np.array_equal(got_inf, exp_inf) and (exp_inf[0] or np.array_equal(got_xy, exp_xy))
