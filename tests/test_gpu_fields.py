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


def test_f29_core_vs_saturated_core(engine):
    """The carry-free 9 x 29-bit core (csrc/bn254_f29.hpp) against exact integer arithmetic, edge values included."""
    rng = Xoshiro(SEED + 5)
    va = [a for a in EDGE for _ in EDGE] + [rng.fp() for _ in range(4000)]
    vb = [b for _ in EDGE for b in EDGE] + [rng.fp() for _ in range(4000)]
    a, b = limbs(va), limbs(vb)
    va, vb = [x % P for x in va], [x % P for x in vb]
    assert ints(engine.f29_hook(0, a, b)) == va
    assert ints(engine.f29_hook(1, a, b)) == [x * y % P for x, y in zip(va, vb)]
    assert ints(engine.f29_hook(2, a, b)) == [2 * x * y % P for x, y in zip(va, vb)]
    assert ints(engine.f29_hook(3, a, b)) == [2 * x * (y - x) % P for x, y in zip(va, vb)]
    assert ints(engine.f29_hook(6, a, b)) == [x * x % P for x in va]                 # dedicated squaring
    # inversion: Bernstein-Yang safegcd (op 4) and the Fermat chain (op 5) agree with each other and with pow(x, p-2, p); inv(0) = 0
    inv = [pow(x, P - 2, P) for x in va]
    assert ints(engine.f29_hook(4, a, b)) == inv and ints(engine.f29_hook(5, a, b)) == inv


def test_f29_tower_vs_saturated_tower(engine, coracle):
    """Fp12 product, cyclotomic square and f^x on the carry-free core vs the saturated core and the oracle."""
    n = 1024
    a12, b12 = fast_rand_fp_array(21, n, 12), fast_rand_fp_array(22, n, 12)
    assert np.array_equal(engine.fp12_hook(8, a12, b12), coracle.fp12_op("mul", a12, b12))
    # cyclotomic formulas are defined on any Fp12 input (they just are not a square outside the subgroup): compare as maps
    assert np.array_equal(engine.fp12_hook(9, a12), coracle.fp12_op("cyclotomic_squared", a12))
    # elements of the cyclotomic subgroup: pairing values
    rng = Xoshiro(SEED + 6)
    G1 = [1, 2]
    G2 = [0x1800DEEF121F1E76426A00665E5C4479674322D4F75EDADD46DEBD5CD992F6ED, 0x198E9393920D483A7260BFB731FB5D25F1AA493335A9E71297E485B7AEF312C2,
          0x12C85EA5DB8C6DEB4AAB71808DCB408FE3D1E7690C43D37B4CE6CC0166FA7DAA, 0x090689D0585FF075EC9E99AD690C3395BC4B313370B38EF355ACDADCD122975B]
    m = 64
    p_xy, _ = engine.g1_scalar_mul(np.repeat(pack(G1, 8), m, 0), limbs([rng.fp() for _ in range(m)]))
    gt = engine.pairing(p_xy, np.repeat(pack(G2, 16), m, 0))
    assert np.array_equal(engine.fp12_hook(9, gt), engine.fp12_sqr(gt))          # in the subgroup the GS formula IS the square
    ez, ez_sat = engine.fp12_hook(10, gt), engine.fp12_hook(11, gt)
    assert np.array_equal(ez, ez_sat)
    # f^x conjugated: check against plain square-and-multiply through the oracle's product
    x = 4965661367192848881
    acc = gt.copy()
    for bit in bin(x)[3:]:
        acc = coracle.fp12_op("sqr", acc)
        if bit == "1":
            acc = coracle.fp12_op("mul", acc, gt)
    assert np.array_equal(ez, coracle.fp12_op("conj", acc))
    # edge: the identity and zero
    z = np.zeros((2, 48), dtype=np.uint64); z[0, 0] = 1
    assert np.array_equal(engine.fp12_hook(10, z), z) and np.array_equal(engine.fp12_hook(8, z, z), z)


def test_fp_pow_sqrt_is_square_vs_oracle(engine, coracle):
    """a5 of SURVEY.md 8: Fp::pow / sqrt / is_square (the latter as a fixed-trip Jacobi symbol on the GPU) against the oracle's
    plain powers, edge values included."""
    rng = Xoshiro(SEED + 9)
    vals = [v % P for v in EDGE] + [rng.fp() for _ in range(1500)]
    vals += [v * v % P for v in vals[:300]]                      # guaranteed squares
    a = limbs(vals)
    sq = engine.fp_is_square(a)
    assert np.array_equal(sq, coracle.fp_is_square(a))
    assert sq.tolist() == [1 if pow(v, (P - 1) // 2, P) in (0, 1) else 0 for v in vals]
    r, ok = engine.fp_sqrt(a)
    er, eok = coracle.fp_sqrt(a)
    assert np.array_equal(r, er) and np.array_equal(ok, eok) and np.array_equal(ok, sq)
    e = limbs([0, 1, 2, P - 1, P - 2, (1 << 256) - 1] + [rng.u256() for _ in range(58)])
    assert np.array_equal(engine.fp_pow(a[:64], e), coracle.fp_pow(a[:64], e))


def test_is_square_and_sqrt_structured_inputs(engine):
    """The Jacobi symbol steps on as many limbs as the wavefront still needs and the square-root chain fetches its window-table entries
    ahead of use: inputs that exercise every limb count from the first step on (values below 2^32 .. 2^256), powers of two and their
    neighbours, p - 2^b, guaranteed squares -- against Python's big-integer power (fp.rs:611-631)."""
    import random
    rnd = random.Random(4242)
    vals = [0, 1, 2, 3, 4, 5, 7, 8, P - 1, P - 2, P - 3, (P - 1) // 2, (P + 1) // 2, (P + 1) // 4]
    for b in range(1, 254):
        vals += [1 << b, (1 << b) - 1, (1 << b) + 1, P - (1 << b)]
    for nl in range(1, 9):
        vals += [rnd.getrandbits(32 * nl) % P for _ in range(600)]
    vals += [rnd.randrange(P) for _ in range(4000)]
    vals += [v * v % P for v in vals[:1500]]
    a = limbs(vals)
    exp = np.array([1 if pow(v, (P - 1) // 2, P) in (0, 1) else 0 for v in vals], dtype=np.uint8)
    assert np.array_equal(engine.fp_is_square(a), exp)
    r, ok = engine.fp_sqrt(a)
    assert np.array_equal(ok, exp)
    roots = ints(r)
    assert all(y * y % P == v for v, y, e in zip(vals, roots, exp) if e)


def test_field_extension_componentwise_operators(engine, coracle):
    """FieldExtension Add / Sub / Neg / scale (extensions.rs:67-238) for Fp2, Fp6, Fp12 batches (scope row a8) against the oracle's
    Fp arithmetic coefficient by coefficient and pyref's tower operators, odd batch size included."""
    from helpers import rand_fp_array
    rng = Xoshiro(SEED + 40)
    for deg, n in ((2, 33), (6, 17), (12, 64)):
        a, b = rand_fp_array(rng, n, deg), rand_fp_array(rng, n, deg)
        k = rand_fp_array(rng, n, 1)
        a[0] = 0; b[1] = 0
        flat = lambda x: x.reshape(n * deg, 4)
        assert np.array_equal(engine.fext_op("add", a, b), coracle.fp_op("add", flat(a), flat(b)).reshape(n, 4 * deg))
        assert np.array_equal(engine.fext_op("sub", a, b), coracle.fp_op("sub", flat(a), flat(b)).reshape(n, 4 * deg))
        assert np.array_equal(engine.fext_op("neg", a), coracle.fp_op("sub", np.zeros_like(flat(a)), flat(a)).reshape(n, 4 * deg))
        assert np.array_equal(engine.fext_op("scale", a, k), coracle.fp_op("mul", flat(a), np.repeat(k, deg, 0)).reshape(n, 4 * deg))
    from oracle import pyref as R
    x, y = rand_fp_array(rng, 3, 2), rand_fp_array(rng, 3, 2)
    got = ints(engine.fext_op("add", x, y))
    xs, ys = ints(x), ints(y)
    for i in range(3):
        assert tuple(got[2 * i: 2 * i + 2]) == R.fp2_add((xs[2 * i], xs[2 * i + 1]), (ys[2 * i], ys[2 * i + 1]))
