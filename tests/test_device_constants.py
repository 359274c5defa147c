"""CPU: the constants the device code carries as literals are the ones the models in tools/ derive from the curve
(tools/gls4_model.py for the 4-way G2 scalar split; the psi / twist digits of plk_group.hip and bn254_pair29.hpp)."""
import importlib.util
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "sylow_amd", "csrc")


def _load(path, name):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _limbs(text, name):
    m = re.search(r"const u32 %s\[\d+\] = \{([^}]*)\};" % re.escape(name), text)
    assert m, name
    words = [int(w.strip().rstrip("u"), 16) for w in m.group(1).split(",")]
    return sum(w << (32 * i) for i, w in enumerate(words))


def test_gls4_rounding_constants_and_decomposition():
    model = _load(os.path.join(ROOT, "tools", "gls4_model.py"), "gls4_model")
    text = open(os.path.join(CSRC, "bn254_pairing.hpp")).read()
    body = text[text.index("BN_DEV void gls4_decompose"):]
    for j in range(4):
        assert _limbs(body, "g%d" % j) == model.G[j], j
        assert model.GSIGN[j] == 1
    x = model.x
    assert model.BASIS == [[2 * x + 1, 0, 2 * x, 1], [2 * x, x + 1, -x, x], [x + 1, x, x, -2 * x], [2 * x + 1, -x, -(x + 1), -x]]
    # the coefficient tables of the device routine are this basis: k_i = [i == 0] k - sum_j c_j B[j][i] with a_j = c_j x
    ca = [[int(v) for v in row.split(",")] for row in re.search(r"const int ca\[4\]\[4\] = \{\{(.*?)\}\};", body).group(1).split("}, {")]
    cc = [[int(v) for v in row.split(",")] for row in re.search(r"const int cc\[4\]\[4\] = \{\{(.*?)\}\};", body).group(1).split("}, {")]
    for i in range(4):
        for j in range(4):
            assert ca[i][j] * x + cc[i][j] == -model.BASIS[j][i], (i, j)
    # and the limb-exact replay agrees with the exact decomposition on edge scalars
    r, lam = model.r, model.lam
    for k in [0, 1, r - 1, r, r + 1, model.P - 1, lam, pow(lam, 3, r), (1 << 253) + 12345]:
        d = model.decompose(k)
        assert [(-m if n else m) for m, n in model.decompose_device(k)] == d
        assert sum(c * pow(lam, i, r) for i, c in enumerate(d)) % r == k % r and max(abs(c) for c in d) < 1 << 64


def test_lane_pair_constant_digits():
    """R-class digits (value * 2^261 mod p, balanced, nine 29-bit digits) of the Fp2 constants written out in the kernels"""
    import sys
    sys.path.insert(0, ROOT)
    from oracle import pyref as R
    P = R.P

    def digits(c):
        v = c * pow(2, 261, P) % P
        if v > P // 2:
            v -= P
        return [(v >> (29 * i)) & ((1 << 29) - 1) for i in range(8)] + [v >> 232]

    def lit(text, name):
        m = re.search(r"const F29 %s\{\{([^}]*)\}\};" % re.escape(name), text)
        assert m, name
        return [int(w.strip(), 0) for w in m.group(1).split(",")]

    conj = lambda a: (a[0], (-a[1]) % P)
    e0, e1 = R.EPS_EXP0, R.EPS_EXP1
    beta = R.fp2_mul(e0, conj(e0))
    e3 = R.fp2_mul(e0, conj(beta))
    assert beta[1] == 0 and R.fp2_mul(e1, conj(e1)) == (P - 1, 0)
    grp = open(os.path.join(CSRC, "plk_group.hip")).read()
    for name, val in (("e0a", e0[0]), ("e0b", e0[1]), ("e1a", e1[0]), ("e1b", e1[1]), ("e3a", e3[0]), ("e3b", e3[1]), ("beta", beta[0])):
        assert lit(grp, name) == digits(val), name
    b3 = (3 * R.TWIST_B[0] % P, 3 * R.TWIST_B[1] % P)
    mulb3 = grp[grp.index("static BN_DEV F mul_b3"):]
    assert lit(mulb3, "k0") == digits(b3[0]) and lit(mulb3, "k1") == digits(b3[1])
    # the isomorphic twist of the multi-step G2 routines: s in Fp with s^6 = 82 / 3, so that b' s^6 = 9 - u and 3 b' s^6 = 27 - 3 u
    p29 = open(os.path.join(CSRC, "bn254_pair29.hpp")).read()

    def fn_lit(text, name):
        m = re.search(r"BN_DEV F29 %s\(\) \{ return F29\{\{([^}]*)\}\}; \}" % re.escape(name), text)
        assert m, name
        return [int(w.strip(), 0) for w in m.group(1).split(",")]

    val = lambda d: sum(w << (29 * i) for i, w in enumerate(d)) * pow(2, -261, P) % P
    s2, s3, s2i, s3i = val(fn_lit(p29, "f29_iso_s2")), val(fn_lit(p29, "f29_iso_s3")), val(lit(grp, "s2i")), val(lit(grp, "s3i"))
    assert s2 * s2i % P == 1 and s3 * s3i % P == 1 and pow(s2, 3, P) == pow(s3, 2, P)        # s2 = s^2, s3 = s^3 for one s
    s6 = pow(s2, 3, P)
    assert s6 * 3 % P == 82
    assert R.fp2_mul(R.TWIST_B, (s6, 0)) == (9, P - 1)
    assert "bn_keep(27), bn_keep_v(lane_odd() ? -3 : 3)" in p29
    tw = p29[p29.index("BN_DEV W2 w2_twist_b()"):]
    assert lit(tw, "k0") == digits(R.TWIST_B[0]) and lit(tw, "k1") == digits(R.TWIST_B[1])
