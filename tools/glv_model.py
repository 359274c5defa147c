"""Model of the GLV decomposition used by g1_scalar_mul (bn254_pairing.hpp): constants + exhaustive-ish checks.
phi(x, y) = (beta x, y) acts on G1 as multiplication by lambda; k = k1 + k2 lambda (mod r) with |k1|, |k2| < 2^128."""
import random, sys
sys.path.insert(0, __file__.rsplit("/tools/", 1)[0])
from oracle import pyref as R
P, r = R.P, R.R_ORDER

def cube_roots(m):
    # non-trivial cube roots of unity mod prime m (m = 1 mod 3)
    for g in range(2, 50):
        w = pow(g, (m - 1) // 3, m)
        if w != 1: return w, w * w % m
lam_c = cube_roots(r); beta_c = cube_roots(P)
G = (1, 2)
def g1_mul_aff(k, pt):
    out = R.g1_mul(R.proj_from_affine(R.F1, pt), k) if hasattr(R, "g1_mul") else None
    return out
# find matching (beta, lambda) using the oracle's group law
from oracle import coracle as C
import numpy as np
def smul(k, pt):
    proj = C.to_limbs([pt[0], pt[1], 1]).reshape(1, 12)
    xy, inf = C.g1_to_affine(C.g1_scalar_mul(proj, C.to_limbs([k])))
    v = C.from_limbs(xy[0]); return (v[0], v[1])
pair = None
for lam in lam_c:
    q = smul(lam, G)
    for beta in beta_c:
        if q == (beta * G[0] % P, G[1]): pair = (beta, lam)
beta, lam = pair
assert (lam * lam + lam + 1) % r == 0 and (beta * beta + beta + 1) % P == 0
# short lattice basis of {(a, b): a + b lam = 0 mod r} by the extended Euclid on (r, lam)
def basis():
    s0, t0, r0 = 1, 0, r
    s1, t1, r1 = 0, 1, lam
    rows = []
    while r1:
        q = r0 // r1
        r0, r1 = r1, r0 - q * r1
        s0, s1 = s1, s0 - q * s1
        t0, t1 = t1, t0 - q * t1
        rows.append((r0, -t0))          # r0 = s0 r + t0 lam  =>  r0 - t0 lam = 0 mod r  => (a, b) = (r0, -t0)
    sq = int(r ** 0.5)
    i = next(i for i, (a, b) in enumerate(rows) if a < sq)
    cand = [rows[i - 1], rows[i], rows[i + 1]]
    v1 = rows[i]
    v2 = min([rows[i - 1], rows[i + 1]], key=lambda v: v[0] * v[0] + v[1] * v[1])
    return v1, v2
(a1, b1), (a2, b2) = basis()
assert (a1 + b1 * lam) % r == 0 and (a2 + b2 * lam) % r == 0
det = a1 * b2 - a2 * b1
assert abs(det) == r
# k = k1 + k2 lam:  (k, 0) - c1 v1 - c2 v2 with c1 = round(b2 k / det), c2 = round(-b1 k / det)
SH = 256
g1c = (b2 << SH) // det if det > 0 else ((-b2) << SH) // (-det)
g2c = ((-b1) << SH) // det if det > 0 else (b1 << SH) // (-det)
def decompose(k):
    k %= r
    # device arithmetic: c_i = (k * |g_i|) >> 256 with the sign applied afterwards (floor instead of round: error < 2)
    c1 = (k * abs(g1c)) >> SH; c1 = c1 if g1c >= 0 else -c1
    c2 = (k * abs(g2c)) >> SH; c2 = c2 if g2c >= 0 else -c2
    k1 = k - c1 * a1 - c2 * a2
    k2 = -c1 * b1 - c2 * b2
    return k1, k2
if __name__ == "__main__":
    random.seed(7)
    mx = 0
    for k in [0, 1, 2, r - 1, r, r + 1, P - 1, lam, lam + 1, r - lam, (1 << 253), (1 << 254) - 1] + [random.randrange(P) for _ in range(20000)]:
        k1, k2 = decompose(k)
        assert (k1 + k2 * lam - k) % r == 0
        mx = max(mx, abs(k1).bit_length(), abs(k2).bit_length())
    print("max bits", mx)
    def L(v, n): return ", ".join("0x%08xu" % ((v >> (32 * i)) & 0xffffffff) for i in range(n))
    print("beta =", hex(beta)); print("lambda =", hex(lam))
    print("a1,b1 =", a1, b1); print("a2,b2 =", a2, b2, "det sign", 1 if det > 0 else -1)
    print("g1c =", g1c, g1c.bit_length(), "g2c =", g2c, g2c.bit_length())
    for name, v in (("|a1|", abs(a1)), ("|b1|", abs(b1)), ("|a2|", abs(a2)), ("|b2|", abs(b2)), ("|g1|", abs(g1c)), ("|g2|", abs(g2c))):
        print(name, v.bit_length(), L(v, 5))
    print("beta mont (R=2^256) =", L(beta * (1 << 256) % P, 8))
    print("beta f29 (x 2^261 mod p) digits =", ", ".join("0x%08x" % (((beta << 261) % P >> (29 * i)) & 0x1fffffff) for i in range(9)))

def decompose_device(k):
    """the arithmetic exactly as the device does it: everything modulo 2^160 after the two high products"""
    M = (1 << 160) - 1
    if k >= r: k -= r                     # k < p < 2r
    c1 = (k * g1c) >> 256
    c2 = (k * g2c) >> 256
    k1 = (k - c1 * a1 - c2 * a2) & M
    k2 = (c2 * (-b2) - c1 * b1) & M
    out = []
    for v in (k1, k2):
        neg = v >> 159
        mag = ((-v) & M) if neg else v
        assert mag < (1 << 128)
        out.append((mag, bool(neg)))
    return out

if __name__ == "__main__":
    random.seed(11)
    for k in [0, 1, 2, r - 1, r, r + 1, P - 1, lam, lam + 1, r - lam, (1 << 253), (1 << 254) - 1] + [random.randrange(P) for _ in range(20000)]:
        (m1, n1), (m2, n2) = decompose_device(k)
        k1 = -m1 if n1 else m1; k2 = -m2 if n2 else m2
        assert (k1 + k2 * lam - k) % r == 0
    print("device-style decomposition ok; a2 limbs:", ", ".join("0x%08xu" % ((a2 >> (32 * i)) & 0xffffffff) for i in range(4)))
