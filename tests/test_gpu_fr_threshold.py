"""Fr batch arithmetic and the threshold-signature flow of examples/threshold_signing.rs, J independent schemes at once."""
import numpy as np
import pytest

from helpers import SEED, Xoshiro
from oracle import pyref as R

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def api(engine):
    from sylow_amd import api
    api.set_engine(engine)
    return api


def test_fr_ops_match_oracle(engine, coracle):
    rng = Xoshiro(SEED + 90)
    r = R.R_ORDER
    edge = [0, 1, 2, r - 1, r, r + 1, 2 * r, 4 * r, 5 * r + 7, (1 << 256) - 1, R.P, R.P - 1]
    for n in (len(edge) + 53, 64):                 # odd (scalar tail path) and even (vector path)
        vals = edge + [rng.u256() for _ in range(n - len(edge))]
        a, b = coracle.to_limbs(vals), coracle.to_limbs(vals[::-1])
        for op in ("add", "sub", "mul"):
            assert np.array_equal(getattr(engine, "fr_" + op)(a, b), coracle.fr_op(op, a, b)), op
        for op in ("sqr", "neg", "inv"):
            assert np.array_equal(getattr(engine, "fr_" + op)(a), coracle.fr_op(op, a)), op
    # inv(0) = 0 and x * inv(x) = 1
    x = coracle.to_limbs([0] + [rng.u256() % r or 1 for _ in range(15)])
    prod = coracle.from_limbs(engine.fr_mul(x, engine.fr_inv(x)))
    assert prod == [0] + [1] * 15


def test_threshold_signing_flow(api, coracle):
    """t-of-n BLS (threshold_signing.rs:30-172) for J independent committees: polynomial evaluation and Lagrange
    coefficients in Fr on the GPU, partial signatures, glued batch verification, weighted aggregation, final verify."""
    J, t, n = 12, 3, 5
    rng = Xoshiro(SEED + 91)
    r = R.R_ORDER
    coeffs = [[rng.u256() % r for _ in range(t)] for _ in range(J)]                    # secret polynomials
    # evaluate_polynomial (threshold_signing.rs:64-70): Horner over Fr, all (job, participant) pairs in one batch
    xs = api.Fr.from_ints([i for i in range(1, n + 1) for _ in range(J)])             # term-major: row (i-1)*J + j
    acc = api.Fr.from_ints([0] * (n * J))
    for c in reversed(range(t)):
        acc = acc * xs + api.Fr.from_ints([coeffs[j][c] for _ in range(n) for j in range(J)])
    sk = acc
    exp_sk = [sum(coeffs[j][c] * i ** c for c in range(t)) % r for i in range(1, n + 1) for j in range(J)]
    assert coracle.from_limbs(sk.v) == exp_sk
    pk = api.G2Projective.generator(n * J) * sk.v
    group_pk = api.G2Projective.generator(J) * api.fp([coeffs[j][0] for j in range(J)])
    msgs = [b"Hello, Sylow! #%d" % j for j in range(J)]
    # partial_sign (:73-90)
    partial = api.sign(sk.v, [msgs[j] for _ in range(n) for j in range(J)])
    # batch_verify_partial (:93-121): per job the 2n pairs (sig_i, G2gen), (-H, pk_i) glued, one final exp, == identity
    h = api.G1Projective.hash_to_curve(msgs)
    neg_h = -h
    g1_rows, g2_rows = [], []
    g2gen = api.G2Affine.generator(1).xy[0]
    for j in range(J):
        for i in range(n):
            g1_rows += [partial.xy[i * J + j], neg_h.xy[j]]
            g2_rows += [g2gen, pk.xy[i * J + j]]
    offsets = [2 * n * j for j in range(J + 1)]
    prod = api.glued_pairing(api.G1Affine(np.array(g1_rows)), api.G2Affine(np.array(g2_rows)), offsets)
    assert (prod == api.Gt.identity(J)).all()
    # a forged partial signature makes that committee's product != identity, and only that one
    bad = np.array(g1_rows)
    bad[2 * n * 3] = partial.xy[0 * J + 4]
    prod_bad = api.glued_pairing(api.G1Affine(bad), api.G2Affine(np.array(g2_rows)), offsets)
    assert list(prod_bad == api.Gt.identity(J)) == [j != 3 for j in range(J)]
    # lagrange_coefficient (:146-155) for the subset {1, 3, 5} and aggregate (:124-143)
    subset = [1, 3, 5]
    lam = api.Fr.from_ints([1] * (len(subset) * J))
    xi = api.Fr.from_ints([i for i in subset for _ in range(J)])
    for shift in range(1, len(subset)):                         # fold over the other participants j != i
        xj = api.Fr.from_ints([subset[(s + shift) % len(subset)] for s in range(len(subset)) for _ in range(J)])
        lam = lam * (xj * (xj - xi).inv())
    exp_lam = []
    for i in subset:
        v = 1
        for jx in subset:
            if jx != i:
                v = v * jx * pow(jx - i, r - 2, r) % r
        exp_lam += [v] * J
    assert coracle.from_limbs(lam.v) == exp_lam
    pts = api.G1Affine(np.concatenate([partial.xy[(i - 1) * J:(i) * J] for i in subset]))
    agg = api.aggregate(pts, lam, J, len(subset))
    # the aggregate equals H(m) * f(0) and verifies under the group key (:158-172)
    assert (agg == api.sign(api.fp([coeffs[j][0] for j in range(J)]), msgs)).all()
    assert api.verify(group_pk, msgs, agg).all()
    # lincomb against the oracle's scalar-mul + add, incl. an identity term and a zero weight; empty sum = identity
    k = lam.v.copy(); k[1] = 0
    inf = np.zeros(len(pts), np.uint8); inf[2] = 1
    got_xy, got_inf = api.engine().g1_lincomb(pts.xy, k, J, len(subset), inf)
    proj = np.concatenate([pts.xy, np.tile(coracle.to_limbs([1]), (len(pts), 1))], axis=1)
    proj[2, 8:] = 0
    terms = coracle.g1_scalar_mul(proj, k)
    acc_o = terms[:J]
    for s in range(1, len(subset)):
        acc_o = coracle.g1_add(acc_o, terms[s * J:(s + 1) * J])
    exp_xy, exp_inf = coracle.g1_to_affine(acc_o)
    assert np.array_equal(got_xy, exp_xy) and np.array_equal(got_inf, exp_inf)
    e_xy, e_inf = api.engine().g1_lincomb(np.zeros((0, 8), np.uint64), np.zeros((0, 4), np.uint64), 4, 0)
    assert e_inf.all()
