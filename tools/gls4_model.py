"""Model of the 4-dimensional GLS decomposition used by the r-torsion G2 scalar multiplication (plk_group.hip):
psi (untwist-Frobenius-twist, g2.rs:140-152) acts on G2 as multiplication by lam = p mod r = 6 x^2, and lam^4 - lam^2 + 1 = 0 mod r,
so k = k0 + k1 lam + k2 lam^2 + k3 lam^3 (mod r) with |k_i| of about 64 bits.  Derives the reduced lattice basis (LLL), the
rounding constants and the bound on |k_i|, and replays the device arithmetic limb-exactly.  Prints the constants as C."""
import random, sys
from fractions import Fraction
sys.path.insert(0, __file__.rsplit("/tools/", 1)[0])
from oracle import pyref as R
P, r, x = R.P, R.R_ORDER, R.BLS_X
lam = P % r
assert lam == 6 * x * x % r and (pow(lam, 4, r) - pow(lam, 2, r) + 1) % r == 0

def lll(B, delta=Fraction(3, 4)):
    B = [list(v) for v in B]
    n = len(B)
    def dot(a, b): return sum(Fraction(x) * y for x, y in zip(a, b))
    def gs():
        Bs, mu = [], [[Fraction(0)] * n for _ in range(n)]
        for i in range(n):
            v = [Fraction(t) for t in B[i]]
            for j in range(i):
                mu[i][j] = dot(B[i], Bs[j]) / dot(Bs[j], Bs[j])
                v = [a - mu[i][j] * b for a, b in zip(v, Bs[j])]
            Bs.append(v)
        return Bs, mu
    k = 1
    Bs, mu = gs()
    while k < n:
        for j in range(k - 1, -1, -1):
            q = round(mu[k][j])
            if q:
                B[k] = [a - q * b for a, b in zip(B[k], B[j])]
                Bs, mu = gs()
        if dot(Bs[k], Bs[k]) >= (delta - mu[k][k - 1] ** 2) * dot(Bs[k - 1], Bs[k - 1]):
            k += 1
        else:
            B[k], B[k - 1] = B[k - 1], B[k]
            Bs, mu = gs()
            k = max(k - 1, 1)
    return B

BASIS = lll([[r, 0, 0, 0], [-lam % r, 1, 0, 0], [-pow(lam, 2, r), 0, 1, 0], [-pow(lam, 3, r), 0, 0, 1]])
for v in BASIS:
    assert sum(c * pow(lam, i, r) for i, c in enumerate(v)) % r == 0

def det(M):
    if len(M) == 1: return M[0][0]
    return sum((-1) ** j * M[0][j] * det([row[:j] + row[j + 1:] for row in M[1:]]) for j in range(len(M)))
DET = det(BASIS)
assert abs(DET) == r
# (k, 0, 0, 0) = sum_j c_j BASIS[j] over the rationals:  c_j = k * cof_j / DET with cof_j the cofactor of entry (j, 0)
COF = [(-1) ** j * det([row[1:] for i, row in enumerate(BASIS) if i != j]) for j in range(4)]
for col in range(4):
    assert sum(COF[j] * BASIS[j][col] for j in range(4)) == (DET if col == 0 else 0)
SH = 320                                                   # c_j ~ round(k |g_j| / 2^320): error < 1 for k < 2^256
G = [((abs(COF[j]) << SH) + abs(DET) // 2) // abs(DET) for j in range(4)]
GSIGN = [(1 if COF[j] >= 0 else -1) * (1 if DET > 0 else -1) for j in range(4)]

def decompose(k):
    k %= r
    c = [GSIGN[j] * ((k * G[j] + (1 << (SH - 1))) >> SH) for j in range(4)]
    return [(k if i == 0 else 0) - sum(c[j] * BASIS[j][i] for j in range(4)) for i in range(4)]

W = 3                                                      # 32-bit limbs kept for each k_i (two's complement)
def decompose_device(k):
    """as the device does it: c_j from the high limbs of k * g_j (+ rounding bit), everything else modulo 2^(32 W)"""
    M = (1 << (32 * W)) - 1
    if k >= r: k -= r                                       # k < p < 2r
    c = [((k * G[j] + (1 << (SH - 1))) >> SH) for j in range(4)]
    out = []
    for i in range(4):
        v = (k if i == 0 else 0)
        for j in range(4):
            t = c[j] * abs(BASIS[j][i])
            v = v - t if GSIGN[j] * (1 if BASIS[j][i] >= 0 else -1) > 0 else v + t
        v &= M
        neg = v >> (32 * W - 1)
        mag = ((-v) & M) if neg else v
        out.append((mag, bool(neg)))
    return out

if __name__ == "__main__":
    random.seed(5)
    ks = [0, 1, 2, r - 1, r, r + 1, P - 1, lam, lam + 1, r - lam, pow(lam, 2, r), pow(lam, 3, r), (1 << 253), (1 << 254) - 1, x, 6 * x * x] + [random.randrange(P) for _ in range(30000)]
    mx = 0
    for k in ks:
        d = decompose(k)
        assert sum(c * pow(lam, i, r) for i, c in enumerate(d)) % r == k % r
        mx = max(mx, max(abs(c).bit_length() for c in d))
        dd = decompose_device(k)
        assert [(-m if n else m) for m, n in dd] == d, (k, d, dd)
    print("max |k_i| bits:", mx)
    print("lambda =", hex(lam))
    def L(v, n): return "{" + ", ".join("0x%08xu" % ((v >> (32 * i)) & 0xffffffff) for i in range(n)) + "}"
    for j in range(4):
        print("basis row", j, BASIS[j], "bits", [abs(c).bit_length() for c in BASIS[j]])
    for j in range(4):
        print("g[%d] sign %+d bits %d limbs %d:" % (j, GSIGN[j], G[j].bit_length(), (G[j].bit_length() + 31) // 32), L(G[j], (G[j].bit_length() + 31) // 32))
    for j in range(4):
        print("|basis[%d]| limbs:" % j, [L(abs(c), 3) for c in BASIS[j]], "signs", [1 if c >= 0 else -1 for c in BASIS[j]])
