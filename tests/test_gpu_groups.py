"""GPU parity: G1/G2 scalar multiplication, addition, normalisation, G2 subgroup check vs the oracle.
Points are compared after affine normalisation, as the reference's own fixtures are."""
import numpy as np
import pytest

from helpers import P, SEED, Xoshiro, fast_rand_fp_array, fp2_sqrt, ints, limbs, pack
from oracle import pyref as R

pytestmark = pytest.mark.gpu

G1 = [1, 2]
G2 = list(R.G2_GEN_AFF[0]) + list(R.G2_GEN_AFF[1])
ONE4 = np.array([[1, 0, 0, 0]], dtype=np.uint64)


def g1_proj(xy, inf=None):
    n = xy.shape[0]
    z = np.repeat(ONE4, n, 0)
    if inf is not None:
        z = z * (1 - np.asarray(inf, dtype=np.uint64))[:, None]
    return np.concatenate([xy, z], axis=1)


def g2_proj(xy, inf=None):
    n = xy.shape[0]
    z = np.concatenate([np.repeat(ONE4, n, 0), np.zeros((n, 4), dtype=np.uint64)], axis=1)
    if inf is not None:
        z = z * (1 - np.asarray(inf, dtype=np.uint64))[:, None]
    return np.concatenate([xy, z], axis=1)


def scalars(rng, n):
    # edge scalars: 0, 1, 2, p-1, r, r-1, r+1, 2^253 pattern, then random (Fp values, NOT reduced mod r: N4)
    edge = [0, 1, 2, 3, P - 1, R.R_ORDER, R.R_ORDER - 1, R.R_ORDER + 1, (1 << 253) - 1, 0x5555555555555555555555555555555555555555555555555555555555555555 % P, P, P + 5]
    return edge + [rng.fp() for _ in range(n - len(edge))]


def test_g1_scalar_mul_vs_oracle(engine, coracle):
    rng = Xoshiro(SEED + 20)
    n = 512
    k = limbs(scalars(rng, n))
    base, _ = engine.g1_scalar_mul(np.repeat(pack(G1, 8), n, 0), limbs([rng.fp() for _ in range(n)]))
    got_xy, got_inf = engine.g1_scalar_mul(base, k)
    exp_xy, exp_inf = coracle.g1_to_affine(coracle.g1_scalar_mul(g1_proj(base), coracle.fp_op("add", k, np.zeros_like(k))))
    assert np.array_equal(got_inf, exp_inf) and np.array_equal(got_xy, exp_xy)
    assert got_inf[0] == 1 and ints(got_xy[0:1]) == [0, 1]              # 0 * P = identity = (0, 1, inf)
    assert np.array_equal(got_xy[1], base[1])                           # 1 * P = P
    assert got_inf[5] == 1                                              # r * P = identity
    # identity input
    z_xy, z_inf = engine.g1_scalar_mul(pack([0, 1], 8), limbs([12345]), p_inf=[1])
    assert z_inf[0] == 1


def test_g1_scalar_mul_glv_edge_scalars(engine, coracle):
    """Scalars that stress the GLV split k = k1 + k2 lambda (bn254_pairing.hpp): multiples and neighbours of lambda, of the
    lattice vectors, powers of two around 2^127 / 2^128, values >= r, on random points and on the identity."""
    rng = Xoshiro(SEED + 25)
    r = R.R_ORDER
    lam = 0x30644e72e131a029048b6e193fd84104cc37a73fec2bc5e9b8ca0b2d36636f23
    a1, b1 = 9931322734385697763, 147946756881789319010696353538189108491
    a2 = 147946756881789319000765030803803410728
    base = [0, 1, lam, r - lam, lam * lam % r, a1, b1, a2, (a1 + b1 * lam) % r, 1 << 127, 1 << 128, (1 << 128) - 1, (1 << 126) + 1,
            r - 1, r, r + 1, P - 1, (r + lam) % P, (3 * lam) % r, (lam << 3) % r]
    ks = []
    for v in base:
        ks += [v % P, (v + 1) % P, (v - 1) % P]
    ks += [(x * lam + y) % r for x in (1, 2, 7, (1 << 127) - 1) for y in (0, 1, (1 << 127) - 1)]
    n = len(ks)
    pts_k = limbs([rng.fp() for _ in range(n)])
    pts, _ = engine.g1_scalar_mul(np.tile(pack(G1, 8), (n, 1)), pts_k)
    inf = np.zeros(n, np.uint8); inf[5] = 1; inf[17] = 1
    k = limbs(ks)
    got_xy, got_inf = engine.g1_scalar_mul(pts, k, p_inf=inf)
    exp_xy, exp_inf = coracle.g1_to_affine(coracle.g1_scalar_mul(g1_proj(pts, inf), k))
    assert np.array_equal(got_inf, exp_inf) and np.array_equal(got_xy, exp_xy)


def test_g2_scalar_mul_vs_oracle(engine, coracle):
    rng = Xoshiro(SEED + 21)
    n = 192
    k = limbs(scalars(rng, n))
    base, _ = engine.g2_scalar_mul(np.repeat(pack(G2, 16), n, 0), limbs([rng.fp() for _ in range(n)]))
    got_xy, got_inf = engine.g2_scalar_mul(base, k)
    exp_xy, exp_inf = coracle.g2_to_affine(coracle.g2_scalar_mul(g2_proj(base), coracle.fp_op("add", k, np.zeros_like(k))))
    assert np.array_equal(got_inf, exp_inf) and np.array_equal(got_xy, exp_xy)


def test_g1_add_and_normalize(engine, coracle):
    rng = Xoshiro(SEED + 22)
    n = 256
    g = np.repeat(pack(G1, 8), n, 0)
    a, _ = engine.g1_scalar_mul(g, limbs([rng.fp() for _ in range(n)]))
    b, _ = engine.g1_scalar_mul(g, limbs([rng.fp() for _ in range(n)]))
    b[0] = a[0]                                           # doubling through the complete addition
    b[1] = a[1]; b[1, 4:8] = limbs([(-ints(a[1:2])[1]) % P])[0]   # P + (-P) = identity
    ainf = np.zeros(n, dtype=np.uint8); binf = np.zeros(n, dtype=np.uint8)
    ainf[2] = 1; a[2] = pack([0, 1], 8)[0]               # identity + Q = Q
    binf[3] = 1; b[3] = pack([0, 1], 8)[0]
    ainf[4] = binf[4] = 1; a[4] = b[4] = pack([0, 1], 8)[0]
    got_xy, got_inf = engine.g1_add(a, b, ainf, binf)
    exp_xy, exp_inf = coracle.g1_to_affine(coracle.g1_add(g1_proj(a, ainf), g1_proj(b, binf)))
    assert np.array_equal(got_inf, exp_inf) and np.array_equal(got_xy, exp_xy)
    assert got_inf[1] == 1 and got_inf[4] == 1 and np.array_equal(got_xy[2], b[2]) and np.array_equal(got_xy[3], a[3])
    # normalisation of non-trivial projective representatives: (X*z, Y*z, z) -> (X, Y)
    z = limbs([rng.fp() or 1 for _ in range(n)])
    proj = np.concatenate([engine.fp_mul(a[:, :4], z), engine.fp_mul(a[:, 4:], z), z], axis=1)
    nx, ninf = engine.g1_normalize(proj)
    assert np.array_equal(nx, a) and not ninf.any()
    proj[7, 8:12] = 0
    nx, ninf = engine.g1_normalize(proj)
    assert ninf[7] == 1 and ints(nx[7:8]) == [0, 1]                     # Z = 0 -> (0, 1, inf), group.rs:480-492
    q, _ = engine.g2_scalar_mul(np.repeat(pack(G2, 16), 8, 0), limbs([rng.fp() for _ in range(8)]))
    zq = fast_rand_fp_array(5, 8, 2)
    projq = np.concatenate([engine.fp2_mul(q[:, :8], zq), engine.fp2_mul(q[:, 8:], zq), zq], axis=1)
    nq, nqinf = engine.g2_normalize(projq)
    assert np.array_equal(nq, q) and not nqinf.any()
    eq, einf = coracle.g2_to_affine(projq)
    assert np.array_equal(nq, eq)


def test_g2_subgroup_check(engine, coracle):
    """g2.rs:460-525: generator multiples pass; twist points outside the r-torsion and off-curve points fail."""
    rng = Xoshiro(SEED + 23)
    good, _ = engine.g2_scalar_mul(np.repeat(pack(G2, 16), 16, 0), limbs([rng.fp() for _ in range(16)]))
    bad = []
    while len(bad) < 8:
        x = (rng.fp(), rng.fp())
        y = fp2_sqrt(R.fp2_add(R.fp2_mul(R.fp2_square(x), x), R.TWIST_B))
        if y is not None:
            assert R.g2_is_on_curve_affine(x, y)
            bad.append(list(x) + list(y))
    bad = pack([v for q in bad for v in q], 16)
    off = good[:4].copy()
    off[:, 8] ^= np.uint64(1)                                            # perturb y: off the curve
    pts = np.concatenate([good, bad, off, pack(G2, 16)])
    inf = np.zeros(pts.shape[0], dtype=np.uint8)
    st = engine.g2_subgroup_check(pts, inf)
    assert st[:16].tolist() == [0] * 16 and st[-1] == 0
    assert st[16:24].tolist() == [2] * 8                                 # NotInSubgroup (cofactor is huge: random twist points are outside G2)
    assert st[24:28].tolist() == [1] * 4                                 # NotOnCurve
    exp = coracle.g2_projective_new(g2_proj(np.concatenate([good, bad])))
    assert np.array_equal(st[:24], exp)
    inf[0] = 1
    assert engine.g2_subgroup_check(pts, inf)[0] == 0                     # identity passes (Z == 0 branch)


def test_g2_small_order_twist_points(engine, coracle):
    """The twist E'(Fp2) has order r * (2p - r) and 2p - r = 10069 * 5864401 * 1875725156269 * (a 177-bit prime): points of SMALL order exist, and on
    them a double-and-add chain keeps meeting P + P, P + (-P) and the identity.  The reference's complete formulas (group.rs:528-599) take all of
    that in their stride; so must the lazy-linear-layer forms and the constant-digit chain of the subgroup check.  S = [r * (2p - r) / 10069] T."""
    h2 = 2 * P - R.R_ORDER
    assert h2 % 10069 == 0
    rng = Xoshiro(SEED + 29)
    tw = []
    while len(tw) < 6:
        x = (rng.fp(), rng.fp())
        y = fp2_sqrt(R.fp2_add(R.fp2_mul(R.fp2_square(x), x), R.TWIST_B))
        if y is not None:
            tw.append(list(x) + list(y))
    t = pack([v for q in tw for v in q], 16)
    n = t.shape[0]
    s1, i1 = engine.g2_scalar_mul(t, limbs([h2 // 10069] * n))
    s, si = engine.g2_scalar_mul(s1, limbs([R.R_ORDER] * n), i1)
    e1 = coracle.g2_scalar_mul(g2_proj(t), limbs([h2 // 10069] * n))
    e_xy, e_inf = coracle.g2_to_affine(coracle.g2_scalar_mul(e1, limbs([R.R_ORDER] * n)))
    assert np.array_equal(si, e_inf) and np.array_equal(s, e_xy)
    assert not si.all()                                                   # 6 random twist points: some have a component of order 10069
    live = np.flatnonzero(si == 0)
    sl = s[live]
    st = engine.g2_subgroup_check(sl)
    assert st.tolist() == [2] * len(live)                                 # on the twist, order 10069: not in G2
    assert np.array_equal(st, coracle.g2_projective_new(g2_proj(sl)))
    # [10069] S = identity, [10068] S = -S, [k] S for k around multiples of the order: the window walk runs inside a group of 10069 elements
    ks = [10069, 10068, 10070, 2 * 10069, 16 * 10069 + 3, 10069 * 10069, (1 << 200) + 12345, 8, 15, 16, 17]
    for k in ks:
        got_xy, got_inf = engine.g2_scalar_mul(sl, limbs([k] * len(live)))
        exp_xy, exp_inf = coracle.g2_to_affine(coracle.g2_scalar_mul(g2_proj(sl), limbs([k] * len(live))))
        assert np.array_equal(got_inf, exp_inf) and np.array_equal(got_xy, exp_xy), k
        if k % 10069 == 0:
            assert got_inf.all()
    # S + S, S + (-S), S + 2S ... through the batch addition and doubling
    d_xy, d_inf = engine.g2_double(sl)
    a_xy, a_inf = engine.g2_add(sl, sl)
    assert np.array_equal(d_xy, a_xy) and np.array_equal(d_inf, a_inf)
    z_xy, z_inf = engine.g2_sub(sl, sl)
    assert z_inf.all()


def test_scalar_mul_group_properties_large(engine):
    """size-independent properties at 2^14: (a+b)P = aP + bP and a(bP) = (ab)P (groups/mod.rs:699-768)."""
    n = 1 << 14
    g = np.repeat(pack(G1, 8), n, 0)
    a, b = fast_rand_fp_array(31, n, 1), fast_rand_fp_array(32, n, 1)
    a[:, 3] >>= np.uint64(2); b[:, 3] >>= np.uint64(2)                    # keep a+b < p
    ap, _ = engine.g1_scalar_mul(g, a)
    bp, _ = engine.g1_scalar_mul(g, b)
    s = engine.fp_add(a, b)
    sp, _ = engine.g1_scalar_mul(g, s)
    sum_xy, sum_inf = engine.g1_add(ap, bp)
    assert np.array_equal(sum_xy, sp) and not sum_inf.any()


def test_g2_endomorphism_vs_oracle(engine, coracle):
    """G2Affine::endomorphism (g2.rs:140-152): psi against the oracle, psi^2 ... relations, identity and off-curve inputs."""
    rng = Xoshiro(SEED + 28)
    n = 70
    q, _ = engine.g2_scalar_mul(np.repeat(pack(G2, 16), n, 0), limbs([rng.fp() for _ in range(n)]))
    inf = np.zeros(n, np.uint8); inf[3] = 1
    out, oinf, st = engine.g2_psi(q, inf)
    exp = coracle.g2_psi(q)
    live = inf == 0
    assert np.array_equal(out[live], exp[live]) and np.array_equal(oinf, inf) and not st.any()
    assert np.array_equal(out[3], q[3])                                   # identity -> identity, coordinates untouched
    # psi acts on G2 as multiplication by p mod r (the Frobenius eigenvalue)
    mu, _ = engine.g2_scalar_mul(q[8:16], limbs([P % R.R_ORDER] * 8))
    assert np.array_equal(mu, out[8:16])
    bad = q.copy(); bad[0, 0] ^= np.uint64(1)                             # off the twist: the image is off the twist too
    _, _, st2 = engine.g2_psi(bad)
    assert st2[0] == 1 and not st2[1:].any()


def test_g1_affine_new_check(engine):
    """G1Affine::new (g1.rs:111-132): curve equation check on random points, perturbed points, edge coordinates and the identity."""
    rng = Xoshiro(SEED + 29)
    n = 64
    pts, _ = engine.g1_scalar_mul(np.repeat(pack(G1, 8), n, 0), limbs([rng.fp() for _ in range(n)]))
    bad = pts.copy()
    bad[::2, 4] ^= np.uint64(1)                                   # y perturbed on even rows
    st = engine.g1_on_curve(bad)
    assert st.tolist() == [1 if i % 2 == 0 else 0 for i in range(n)]
    extra = np.concatenate([pack([0, 0], 8), pack([1, 2], 8), pack([1, P - 2], 8), pack([0, 1], 8), pack([P - 1, 5], 8)])
    inf = np.array([0, 0, 0, 1, 0], np.uint8)
    exp = [0 if (y * y - x * x * x - 3) % P == 0 else 1 for x, y in [(0, 0), (1, 2), (1, P - 2), (0, 1), (P - 1, 5)]]
    exp[3] = 0
    assert engine.g1_on_curve(extra, inf).tolist() == exp


def test_g2_scalar_mul_subgroup_split_vs_generic_and_oracle(engine, coracle):
    """sylow_hip_g2_scalar_mul_subgroup_batch (4-way split along psi, tools/gls4_model.py) == the generic window product == the
    oracle's double-and-add on r-torsion points, for edge scalars (0, 1, r-1, r, r+1, p-1, lambda and its powers, 2^k) and random ones."""
    from helpers import P
    r = R.R_ORDER
    lam = P % r
    rng = Xoshiro(SEED + 27)
    edge = [0, 1, 2, r - 1, r, r + 1, P - 1, lam, lam + 1, r - lam, lam * lam % r, pow(lam, 3, r), 1 << 253, (1 << 254) - 1, R.BLS_X, 6 * R.BLS_X ** 2,
            (1 << 64) - 1, 1 << 64, (1 << 128) + 5, 7 << 190]
    ks = [k % P for k in edge] + [rng.fp() for _ in range(90 - len(edge))]            # scalars are Fp values (N4)
    n = len(ks)
    base, _ = engine.g2_scalar_mul(np.repeat(pack(G2, 16), n, 0), limbs([rng.fp() for _ in range(n)]))
    base[0] = pack(G2, 16)[0]
    got, got_inf = engine.g2_scalar_mul(base, limbs(ks), subgroup=True)
    ref, ref_inf = engine.g2_scalar_mul(base, limbs(ks))
    assert np.array_equal(got, ref) and np.array_equal(got_inf, ref_inf)
    assert got_inf.tolist() == [1 if k % r == 0 else 0 for k in ks]
    exp, exp_inf = coracle.g2_to_affine(coracle.g2_scalar_mul(g2_proj(base[:24]), limbs(ks[:24])))
    assert np.array_equal(got[:24], exp) and np.array_equal(got_inf[:24], exp_inf)
    # identity in, identity out; and the eigenvalue itself: lambda * Q == psi(Q)
    inf = np.zeros(n, dtype=np.uint8); inf[::3] = 1
    a, ai = engine.g2_scalar_mul(base, limbs(ks), inf, subgroup=True)
    b, bi = engine.g2_scalar_mul(base, limbs(ks), inf)
    assert np.array_equal(a, b) and np.array_equal(ai, bi) and ai[::3].all()
    psi_xy, _, _ = engine.g2_psi(base[:8])
    lam_q, _ = engine.g2_scalar_mul(base[:8], limbs([lam] * 8), subgroup=True)
    assert np.array_equal(lam_q, psi_xy)


def test_g2_generator_mul_fixed_base_table(engine, coracle):
    """sylow_hip_g2_generator_mul_batch (32 signed 8-bit digits against the device's table j 256^w G) == the window product of the
    generator == the oracle, for scalars that hit digit 0, +-128, carries through every window, k >= r, and random ones."""
    r = R.R_ORDER
    rng = Xoshiro(SEED + 28)
    edge = [0, 1, 2, 127, 128, 129, 255, 256, 257, (1 << 8) - 1, (1 << 248), (1 << 253) + 128, r - 1, r, r + 1, P - 1,
            int("80" * 31, 16), int("7f" * 31, 16), int("ff" * 31, 16), int("0180" * 15, 16), R.BLS_X]
    ks = [k % P for k in edge] + [rng.fp() for _ in range(100 - len(edge))]
    n = len(ks)
    got, got_inf = engine.g2_generator_mul(limbs(ks))
    ref, ref_inf = engine.g2_scalar_mul(np.repeat(pack(G2, 16), n, 0), limbs(ks))
    assert np.array_equal(got, ref) and np.array_equal(got_inf, ref_inf)
    assert got_inf.tolist() == [1 if k % r == 0 else 0 for k in ks]
    exp, exp_inf = coracle.g2_to_affine(coracle.g2_scalar_mul(g2_proj(np.repeat(pack(G2, 16), 24, 0)), limbs(ks[:24])))
    assert np.array_equal(got[:24], exp) and np.array_equal(got_inf[:24], exp_inf)
    # a second call reuses the table
    again, _ = engine.g2_generator_mul(limbs(ks))
    assert np.array_equal(again, got)


def test_g1_generator_mul_fixed_base_table(engine, coracle):
    """sylow_hip_g1_generator_mul_batch == the GLV window product of the generator == the oracle (same scalar set as the G2 table)."""
    r = R.R_ORDER
    rng = Xoshiro(SEED + 29)
    edge = [0, 1, 2, 127, 128, 129, 255, 256, 257, (1 << 248), (1 << 253) + 128, r - 1, r, r + 1, P - 1, P - 2,
            int("80" * 31, 16), int("7f" * 31, 16), int("ff" * 31, 16), int("0180" * 15, 16), int("30" + "ff" * 31, 16)]
    ks = [k % P for k in edge] + [rng.fp() for _ in range(130 - len(edge))]
    n = len(ks)
    got, got_inf = engine.g1_generator_mul(limbs(ks))
    ref, ref_inf = engine.g1_scalar_mul(np.repeat(pack([1, 2], 8), n, 0), limbs(ks))
    assert np.array_equal(got, ref) and np.array_equal(got_inf, ref_inf)
    assert got_inf.tolist() == [1 if k % r == 0 else 0 for k in ks]
    one = np.zeros((32, 4), dtype=np.uint64); one[:, 0] = 1
    exp, _ = coracle.g1_to_affine(coracle.g1_scalar_mul(np.concatenate([np.repeat(pack([1, 2], 8), 32, 0), one], axis=1), limbs(ks[:32])))
    assert np.array_equal(got[:32][~got_inf[:32].astype(bool)], exp[~got_inf[:32].astype(bool)])


def test_g1_double_direct_vs_oracle(engine, coracle):
    """sylow_hip_g1_double_batch against the oracle's `double` (group.rs:339-386 through g1.rs) row by row -- random points, the generator,
    its negative, the identity (flagged, and as the canonical (0, 1)), a ragged size -- and against add(P, P): E(Fp) has no 2-torsion
    (r is odd), so no finite input doubles to the identity."""
    rng = Xoshiro(SEED + 31)
    n = 131
    pts, _ = engine.g1_scalar_mul(np.repeat(pack(G1, 8), n, 0), limbs([rng.fp() for _ in range(n)]))
    pts[0] = pack(G1, 8)[0]
    pts[1] = pack([1, P - 2], 8)[0]
    inf = np.zeros(n, dtype=np.uint8)
    inf[[2, 64, n - 1]] = 1
    pts[2] = pack([0, 1], 8)[0]                                         # the canonical encoding of the identity, flagged
    got_xy, got_inf = engine.g1_double(pts, inf)
    one = np.zeros((n, 4), dtype=np.uint64); one[:, 0] = 1
    proj = np.concatenate([pts, one], axis=1)
    proj[inf.astype(bool)] = np.concatenate([pack([0, 1], 8)[0], np.zeros(4, dtype=np.uint64)])       # (0, 1, 0)
    exp_xy, exp_inf = coracle.g1_to_affine(coracle.g1_double(proj))
    assert np.array_equal(got_inf, exp_inf) and np.array_equal(got_inf, inf)
    assert np.array_equal(got_xy, exp_xy)
    add_xy, add_inf = engine.g1_add(pts, pts, inf, inf)
    fin = ~inf.astype(bool)
    assert np.array_equal(add_xy[fin], got_xy[fin]) and np.array_equal(add_inf, got_inf)
    assert not got_inf[fin].any()                                      # 2-torsion-free
