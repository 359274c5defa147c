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
