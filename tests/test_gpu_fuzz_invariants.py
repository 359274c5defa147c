"""The invariants of the reference's libFuzzer target (fuzz/fuzz_targets/fuzz_sylow_api.rs:10-73), replayed on
batches of random points through the API mirror, plus Gt * Fr against the oracle."""
import numpy as np
import pytest

from helpers import SEED, Xoshiro, limbs
from oracle import pyref as R

pytestmark = pytest.mark.gpu
N = 96


@pytest.fixture(scope="module")
def api(engine):
    from sylow_amd import api
    api.set_engine(engine)
    return api


def test_fuzz_target_invariants(api, coracle):
    rng = Xoshiro(SEED + 80)
    fr = lambda: api.fp([rng.fp() % R.R_ORDER for _ in range(N)])
    a = api.G1Projective.generator(N) * fr()
    b = api.G1Projective.generator(N) * fr()
    assert ((a + b) == (b + a)).all()                                      # G1 addition is commutative
    c = api.G2Projective.generator(N) * fr()
    d = api.G2Projective.generator(N) * fr()
    assert ((c + d) == (d + c)).all()                                      # G2 addition is commutative
    assert (a.double() == (a + a)).all() and (c.double() == (c + c)).all() # doubling
    three = api.fp([3] * N)
    assert ((a + (a + a)) == (a * three)).all() and ((c + (c + c)) == (c * three)).all()
    # bilinearity with Gt * Fr (gt::tests::test_bilinearity)
    p, q, s = api.G1Projective.generator(N) * fr(), api.G2Projective.generator(N) * fr(), fr()
    e = api.pairing(p, q)
    lhs = e * s
    assert (lhs == api.pairing(p * s, q)).all() and (lhs == api.pairing(p, q * s)).all()
    assert not (lhs == api.Gt.identity(N)).any()
    minus_one = api.fp([R.R_ORDER - 1] * N)
    assert (((lhs * minus_one) + lhs) == api.Gt.identity(N)).all()         # inverse property
    assert ((-lhs) + lhs == api.Gt.identity(N)).all()
    # Gt * Fr vs the oracle (reference algorithm, gt.rs:161-187), incl. edge scalars
    k = api.fp([0, 1, 2, R.R_ORDER - 1, R.R_ORDER - 2] + [rng.fp() % R.R_ORDER for _ in range(11)])
    assert np.array_equal((api.Gt(e.v[:16]) * k).v, coracle.gt_pow(e.v[:16], k))
    # hash_to_curve with the fuzz target's DST never yields the identity
    data = [bytes(rng.next() & 0xFF for _ in range(32)) for _ in range(N)]
    h = api.G1Projective.hash_to_curve(data, b"QUUX-V01-CS02-with-expander-SHA256-128")
    assert not h.is_zero().any()
    # BLS: sign / verify on fresh keys
    sk = fr()
    pk = api.G2Projective.generator(N) * sk
    assert api.verify(pk, data, api.sign(sk, data)).all()


def test_gt_pow_exact_for_any_fp12_and_every_window_pattern(engine, coracle):
    """Mul<&Fr> for &Gt on the window-table algorithm (plk_group.hip): the reference's digit walk yields g^(K+) conj(g)^(K-) for ANY
    Fp12 input (gt.rs:161-187 uses the conjugate as the inverse without checking that g is unitary), so random NON-unitary
    inputs must match the oracle's 256-step walk bit for bit; scalars chosen to hit all 21 four-digit NAF patterns, the digit
    carry out of bit 255 (dropped by the reference's 256-bit arithmetic as well) and a ragged batch size."""
    from helpers import rand_fp_array
    rng = Xoshiro(SEED + 81)
    special = [0, 1, 2, 3, 5, 7, 9, 11, 13, 15, (1 << 256) - 1, 1 << 255, (1 << 255) - 1, R.R_ORDER, R.R_ORDER - 1,
               int("a" * 64, 16), int("5" * 64, 16), int("3" * 64, 16), int("b" * 64, 16), int("d" * 64, 16), int("9" * 64, 16), int("69" * 32, 16)]
    scalars = special + [rng.u256() for _ in range(37 + 64 - len(special))]
    n = len(scalars)
    g = rand_fp_array(rng, n, 12)                       # arbitrary Fp12 values: not unitary, not even in the cyclotomic subgroup
    k = limbs(scalars)
    got = engine.gt_pow(g, k)
    assert np.array_equal(got, coracle.gt_pow(g, k))
    # every pattern of the table was exercised by this scalar set
    seen = set()
    for x in scalars:
        xh = x >> 1
        x3 = (x + xh) & ((1 << 256) - 1)
        c = xh ^ x3
        np_, nm_ = x3 & c, xh & c
        for w in range(64):
            seen.add(((np_ >> (4 * w)) & 15, (nm_ >> (4 * w)) & 15))
    assert len(seen) == 21
    # g^0 = 1 and g^1 = g for any g
    one = np.zeros(48, dtype=np.uint64); one[0] = 1
    assert np.array_equal(got[0], one) and np.array_equal(got[1], g[1])
    # pairing values (the cyclotomic subgroup: the kernel squares them with the Granger-Scott formulas) with the same scalars, a
    # full wavefront and a ragged tail; and a batch that MIXES them with arbitrary Fp12 values inside one wavefront (generic squarings)
    from test_gpu_multi_pairing import G1, G2
    from helpers import pack
    p, _ = engine.g1_scalar_mul(np.repeat(pack(G1, 8), n, 0), limbs([rng.fp() for _ in range(n)]))
    q, _ = engine.g2_scalar_mul(np.repeat(pack(G2, 16), n, 0), limbs([rng.fp() for _ in range(n)]), subgroup=True)
    e = engine.pairing(p, q)
    assert np.array_equal(engine.gt_pow(e, k), coracle.gt_pow(e, k))
    mixed = e.copy()
    mixed[3::7] = g[3::7]
    assert np.array_equal(engine.gt_pow(mixed, k), coracle.gt_pow(mixed, k))
    zero = np.zeros_like(e[:5])                                              # 0 is not in the subgroup (0 * 0 == 0 passes the naive check)
    assert np.array_equal(engine.gt_pow(zero, k[5:10]), coracle.gt_pow(zero, k[5:10]))
