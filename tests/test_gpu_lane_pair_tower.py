"""The lane-pair Fp12 layer (bn254_pair.hpp saturated, bn254_pair29.hpp carry-free) one operation at a time through the
fp12 hook (ops 16..29), on random inputs and on inputs crafted so that the INTERNAL 29-bit digits are extreme."""
import numpy as np
import pytest

from helpers import P, SEED, Xoshiro
from oracle import pyref as R

pytestmark = pytest.mark.gpu
OP = {"mul": 16, "sqr": 17, "sparse": 18, "cycsqr": 19, "frob1": 20, "frob2": 21, "frob3": 22, "expz": 23,
      "s_mul": 24, "s_sqr": 25, "s_inv": 26, "s_cycsqr": 27, "conj": 28, "sparse_unit": 29}
RP_INV = pow(pow(2, 261, P), P - 2, P)           # the carry-free core holds x * 2^261 mod p


def unit_lines(coracle, b):
    """the line operand of op 29: first coefficient (row & 1) in Fp2, the other two as given (normalised line tables)"""
    u = b[:, :24].copy()
    u[:, :8] = 0
    u[1::2, 0] = 1
    return u


def crafted_values():
    """canonical x whose internal representation v = x * 2^261 mod p (centred, |v| < 0.45 p) has extreme digits"""
    m29 = (1 << 29) - 1
    lows = [sum(m29 << (29 * i) for i in range(8)),                       # all ones
            0,                                                             # all zero
            sum((m29 if i % 2 else 0) << (29 * i) for i in range(8)),      # alternating
            sum((m29 if i % 2 == 0 else 1) << (29 * i) for i in range(8)),
            sum((1 << 28) << (29 * i) for i in range(8))]
    tops = [0, 1, -1, 1_400_000, -1_400_000, 700_001, -700_001]
    out = []
    for lo in lows:
        for t in tops:
            v = lo + (t << 232)
            assert abs(v) < 0.46 * P
            out.append(v * RP_INV % P)
    return out


def test_lane_pair_fp12_ops_random(engine, coracle):
    rng = Xoshiro(SEED + 300)
    n = 70
    a = coracle.to_limbs([rng.fp() for _ in range(12 * n)]).reshape(n, 48)
    b = coracle.to_limbs([rng.fp() for _ in range(12 * n)]).reshape(n, 48)
    a[0] = 0; a[1] = 0; a[1, 0] = 1
    for pre in ("", "s_"):
        assert np.array_equal(engine.fp12_hook(OP[pre + "mul"], a, b), coracle.fp12_op("mul", a, b)), pre
        assert np.array_equal(engine.fp12_hook(OP[pre + "sqr"], a), coracle.fp12_op("sqr", a)), pre
        assert np.array_equal(engine.fp12_hook(OP[pre + "cycsqr"], a), coracle.fp12_op("cyclotomic_squared", a)), pre
    assert np.array_equal(engine.fp12_hook(OP["s_inv"], a), coracle.fp12_op("inv", a))
    assert np.array_equal(engine.fp12_hook(OP["conj"], a), coracle.fp12_op("conj", a))
    assert np.array_equal(engine.fp12_hook(OP["sparse"], a, b), coracle.fp12_sparse_mul(a, b[:, :24]))
    assert np.array_equal(engine.fp12_hook(OP["sparse_unit"], a, b), coracle.fp12_sparse_mul(a, unit_lines(coracle, b)))
    for e in (1, 2, 3):
        assert np.array_equal(engine.fp12_hook(OP["frob%d" % e], a), coracle.fp12_op("frobenius", a, arg=e))


def test_lane_pair_fp12_ops_extreme_digits(engine, coracle):
    vals = crafted_values()
    rng = Xoshiro(SEED + 301)
    n = 64
    pick = lambda: vals[rng.next() % len(vals)]
    a = coracle.to_limbs([pick() for _ in range(12 * n)]).reshape(n, 48)
    b = coracle.to_limbs([pick() for _ in range(12 * n)]).reshape(n, 48)
    # rows where every coefficient is the same extreme value
    for k, v in enumerate(vals[:32]):
        a[k] = coracle.to_limbs([v] * 12).reshape(48)
        b[k] = coracle.to_limbs([vals[(k * 7 + 3) % len(vals)]] * 12).reshape(48)
    assert np.array_equal(engine.fp12_hook(OP["mul"], a, b), coracle.fp12_op("mul", a, b))
    assert np.array_equal(engine.fp12_hook(OP["mul"], a, a), coracle.fp12_op("mul", a, a))
    assert np.array_equal(engine.fp12_hook(OP["sqr"], a), coracle.fp12_op("sqr", a))
    assert np.array_equal(engine.fp12_hook(OP["cycsqr"], a), coracle.fp12_op("cyclotomic_squared", a))
    assert np.array_equal(engine.fp12_hook(OP["sparse"], a, b), coracle.fp12_sparse_mul(a, b[:, :24]))
    assert np.array_equal(engine.fp12_hook(OP["sparse_unit"], a, b), coracle.fp12_sparse_mul(a, unit_lines(coracle, b)))
    for e in (1, 2, 3):
        assert np.array_equal(engine.fp12_hook(OP["frob%d" % e], a), coracle.fp12_op("frobenius", a, arg=e))
    # chains: outputs fed back as inputs keep the invariants (40 dependent squarings / products)
    x, ex = a.copy(), a.copy()
    for _ in range(12):
        x, ex = engine.fp12_hook(OP["sqr"], x), coracle.fp12_op("sqr", ex)
        x, ex = engine.fp12_hook(OP["mul"], x, b), coracle.fp12_op("mul", ex, b)
        x, ex = engine.fp12_hook(OP["cycsqr"], x), coracle.fp12_op("cyclotomic_squared", ex)
    assert np.array_equal(x, ex)


def test_lane_pair_expz_matches_single_lane(engine, coracle):
    rng = Xoshiro(SEED + 302)
    n = 40
    a = coracle.to_limbs([rng.fp() for _ in range(12 * n)]).reshape(n, 48)
    easy = coracle.fp12_op("mul", coracle.fp12_op("conj", a), coracle.fp12_op("inv", a))
    cyc = coracle.fp12_op("mul", coracle.fp12_op("frobenius", easy, arg=2), easy)       # in the cyclotomic subgroup
    got = engine.fp12_hook(OP["expz"], cyc)
    assert np.array_equal(got, engine.fp12_hook(11, cyc))                                # saturated single-lane twin
    # f^x then conjugate against the oracle's signed-digit power (exact in the cyclotomic subgroup)
    for row in range(3):
        f = R.fp12_unflatten(coracle.from_limbs(cyc[row]))
        e = R.fp12_unitary_inverse(R.gt_pow(f, R.BLS_X))
        assert coracle.from_limbs(got[row]) == R.fp12_flatten(e)
