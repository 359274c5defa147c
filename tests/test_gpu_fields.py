"""GPU parity: HIP field tower (through the C ABI) vs the C oracle, bit-exact."""
import numpy as np
import pytest

from helpers import P, SEED, Xoshiro, fast_rand_fp_array, ints, limbs, pack

pytestmark = pytest.mark.gpu

EDGE = [0, 1, 2, P - 1, P - 2, (1 << 256) % P, P, P + 1, (1 << 256) - 1, (1 << 255), 0xFFFFFFFFFFFFFFFF, 1 << 64, (P - 1) // 2]


def test_fp_kats(engine, kats):
    I = lambda s: int(s, 16)
    for name, fn in (("fp_add", engine.fp_add), ("fp_sub", engine.fp_sub), ("fp_mul", engine.fp_mul)):
        a = limbs([I(c[0]) for c in kats[name]["cases"]])
        b = limbs([I(c[1]) for c in kats[name]["cases"]])
        assert ints(fn(a, b)) == [I(c[2]) for c in kats[name]["cases"]], name


def test_fp_edge_and_random(engine, coracle):
    rng = Xoshiro(SEED + 2)
    vals_a = [a for a in EDGE for _ in EDGE] + [rng.fp() for _ in range(3000)]
    vals_b = [b for _ in EDGE for b in EDGE] + [rng.fp() for _ in range(3000)]
    a, b = limbs(vals_a), limbs(vals_b)
    for op, fn in (("add", engine.fp_add), ("sub", engine.fp_sub), ("mul", engine.fp_mul)):
        got = fn(a, b)
        exp = [{"add": (x + y) % P, "sub": (x - y) % P, "mul": (x * y) % P}[op] for x, y in zip(vals_a, vals_b)]
        assert ints(got) == exp, op
        assert np.array_equal(got, coracle.fp_op(op, a, b)), op
    assert ints(engine.fp_sqr(a)) == [x * x % P for x in vals_a]
    assert ints(engine.fp_neg(a)) == [(-x) % P for x in vals_a]
    inv = ints(engine.fp_inv(a))
    assert inv == [pow(x % P, P - 2, P) for x in vals_a]
    assert inv[0] == 0  # inv(0) = 0, fp.rs:1126-1132


def test_fp_mul_large_batch_vs_oracle(engine, coracle):
    n = 1 << 18
    a, b = fast_rand_fp_array(1, n, 1), fast_rand_fp_array(2, n, 1)
    assert np.array_equal(engine.fp_mul(a, b), coracle.fp_op("mul", a, b))


def test_fp2_fp6_kats(engine, kats):
    I = lambda s: int(s, 16)
    for a, b, c in kats["fp2_mul"]["cases"]:
        assert ints(engine.fp2_mul(pack([I(x) for x in a], 8), pack([I(x) for x in b], 8))) == [I(x) for x in c]
    for a, b, c in kats["fp6_mul"]["cases"]:
        assert ints(engine.fp6_mul(pack([I(x) for x in a], 24), pack([I(x) for x in b], 24))) == [I(x) for x in c]
    # division KATs: a / b = a * inv(b)
    for a, b, c in kats["fp2_div"]["cases"]:
        binv = engine.fp2_inv(pack([I(x) for x in b], 8))
        assert ints(engine.fp2_mul(pack([I(x) for x in a], 8), binv)) == [I(x) for x in c]
    for a, b, c in kats["fp6_div"]["cases"]:
        binv = engine.fp6_inv(pack([I(x) for x in b], 24))
        assert ints(engine.fp6_mul(pack([I(x) for x in a], 24), binv)) == [I(x) for x in c]


def test_tower_random_vs_oracle(engine, coracle):
    n = 2048
    a2, b2 = fast_rand_fp_array(3, n, 2), fast_rand_fp_array(4, n, 2)
    assert np.array_equal(engine.fp2_mul(a2, b2), coracle.fp2_op("mul", a2, b2))
    assert np.array_equal(engine.fp2_sqr(a2), coracle.fp2_op("sqr", a2))
    assert np.array_equal(engine.fp2_inv(a2), coracle.fp2_op("inv", a2))
    a6, b6 = fast_rand_fp_array(5, n, 6), fast_rand_fp_array(6, n, 6)
    assert np.array_equal(engine.fp6_mul(a6, b6), coracle.fp6_op("mul", a6, b6))
    assert np.array_equal(engine.fp6_inv(a6), coracle.fp6_op("inv", a6))
    a12, b12 = fast_rand_fp_array(7, n, 12), fast_rand_fp_array(8, n, 12)
    assert np.array_equal(engine.fp12_mul(a12, b12), coracle.fp12_op("mul", a12, b12))
    assert np.array_equal(engine.fp12_sqr(a12), coracle.fp12_op("sqr", a12))
    assert np.array_equal(engine.fp12_inv(a12), coracle.fp12_op("inv", a12))
    for e in (1, 2, 3):
        assert np.array_equal(engine.fp12_frobenius(a12, e), coracle.fp12_op("frobenius", a12, arg=e)), e
    ell = fast_rand_fp_array(9, n, 6)
    assert np.array_equal(engine.fp12_sparse_mul(a12, ell), coracle.fp12_sparse_mul(a12, ell))
    # zero / one / zero-divisor-free edge elements
    z = np.zeros((4, 48), dtype=np.uint64)
    z[1, 0] = 1
    z[2] = a12[0]
    z[3] = limbs([P - 1] * 12).reshape(48)
    assert np.array_equal(engine.fp12_mul(z, z[::-1].copy()), coracle.fp12_op("mul", z, z[::-1].copy()))
    assert np.array_equal(engine.fp12_inv(z), coracle.fp12_op("inv", z))  # inv(0) = 0 propagates


def test_fp12_algebraic_properties_full_size(engine):
    """size-independent properties at 2^16: square == a*a, Frobenius order (fp12.rs:654-708)."""
    n = 1 << 16
    a = fast_rand_fp_array(10, n, 12)
    assert np.array_equal(engine.fp12_sqr(a), engine.fp12_mul(a, a))
    f = a
    for _ in range(6):
        f = engine.fp12_frobenius(f, 2)
    assert np.array_equal(f, a)  # frobenius(2)^6 = identity
    ainv = engine.fp12_inv(a)
    one = np.zeros((n, 48), dtype=np.uint64)
    one[:, 0] = 1
    assert np.array_equal(engine.fp12_mul(a, ainv), one)
