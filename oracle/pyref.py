"""Pure-Python big-int restatement of sylow's BN254 hot path.  TEST INFRASTRUCTURE ONLY.

This file is the *second*, independent oracle (the first is the C restatement in
``oracle/sylow_oracle.c``).  It exists to (1) cross-check the C oracle, (2) pin both
against the reference's in-tree known-answer vectors (tests/golden/*.json) and
(3) generate small golden cases.  Pure-Python loops: use it on small inputs only.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s cpu_baseline leg may
import anything under ``oracle/``.  The product path (sylow_amd/) never does.

Every function cites the reference lines it restates (paths under /root/reference).
Values are plain canonical integers in [0, p); Fp2 = (c0, c1); Fp6 = (Fp2, Fp2, Fp2);
Fp12 = (Fp6, Fp6).  Nothing here is Montgomery form: the reference's results are exact
residues, so the representation is irrelevant to parity.

Parity status: pinned by the reference's own KATs (Fp/Fp2/Fp6 products, SvdW constants,
Gt generator, pairing test_cases, EIP-196/197 vectors).  The Keccak-256 XMD -> SvdW ->
sign chain is **parity unpinned** by the reference (no literal anywhere in its tests);
it is pinned here by public Keccak-256 KATs, RFC 9380 SHA-256 vectors through the same
XMD code, and sign/verify round trips.
"""
from __future__ import annotations

import hashlib

# --- constants: src/fields/fp.rs:51-76,538-542 ; src/groups/g2.rs:112 -------------------
P = 0x30644E72E131A029B85045B68181585D97816A916871CA8D3C208C16D87CFD47
R_ORDER = 0x30644E72E131A029B85045B68181585D2833E84879B9709143E1F593F0000001
BLS_X = 4965661367192848881
assert P == 36 * BLS_X**4 + 36 * BLS_X**3 + 24 * BLS_X**2 + 6 * BLS_X + 1
assert R_ORDER == 36 * BLS_X**4 + 36 * BLS_X**3 + 18 * BLS_X**2 + 6 * BLS_X + 1
assert P % 4 == 3

# src/pairing.rs:26-30 -- 64 NAF digits of 6x+2 after the implicit leading one, MSB first
ATE_LOOP_COUNT_NAF = [
    1, 0, 1, 0, 0, 0, -1, 0, -1, 0, 0, 0, -1, 0, 1, 0, -1, 0, 0, -1, 0, 0, 0, 0, 0, 1, 0, 0, -1, 0,
    1, 0, 0, -1, 0, 0, 0, 0, -1, 0, 1, 0, 0, 0, -1, 0, -1, 0, 0, 1, 0, 0, 0, -1, 0, 0, -1, 0, 1, 0,
    1, 0, 0, 0,
]
_acc = 1
for _d in ATE_LOOP_COUNT_NAF:
    _acc = 2 * _acc + _d
assert _acc == 6 * BLS_X + 2

DST = b"WARLOCK-CHAOS-V01-CS01-SHA-256"  # src/lib.rs:90
SECURITY_BITS = 128  # src/lib.rs:94


# --- Fp: src/fields/fp.rs:304-457,585-737 -----------------------------------------------
def fp_new(v: int) -> int:
    """Fp::new reduces ANY 256-bit value mod p (fp.rs:199-201, asserted fp.rs:878-884)."""
    return v % P


def fp_inv(a: int) -> int:
    """inv(0) = 0, no panic (fp.rs:418-433, fp.rs:1126-1132)."""
    return pow(a, P - 2, P)


def fp_sqrt(a: int):
    """a^((p+1)/4) with square check (fp.rs:611-616). Returns None when not a square."""
    s = pow(a, (P + 1) // 4, P)
    return s if s * s % P == a else None


def fp_is_square(a: int) -> bool:
    """a^((p-1)/2) in {0, 1} (fp.rs:625-631)."""
    return pow(a, (P - 1) // 2, P) in (0, 1)


def fp_sgn0(a: int) -> int:
    """fp.rs:636-644"""
    return a & 1


def compute_naf(x: int):
    """fp.rs:653-662 -- returns (np, nm) bitmasks; digit_i = np_i - nm_i."""
    xh = x >> 1
    x3 = (x + xh) & ((1 << 256) - 1)
    c = xh ^ x3
    return x3 & c, xh & c


def fp_from_be_bytes(b: bytes):
    """fp.rs:686-719 -- rejects values >= p (returns None)."""
    assert len(b) == 32
    v = int.from_bytes(b, "big")
    return v if v < P else None


def fp_to_be_bytes(a: int) -> bytes:
    """fp.rs:727-737"""
    return a.to_bytes(32, "big")


# --- Fp2 = Fp[u]/(u^2+1): src/fields/fp2.rs ------------------------------------------------
FP2_ZERO = (0, 0)
FP2_ONE = (1, 0)
TWO_INV = fp_inv(2)  # fp2.rs:18-23


def fp2_add(a, b):
    return ((a[0] + b[0]) % P, (a[1] + b[1]) % P)


def fp2_sub(a, b):
    return ((a[0] - b[0]) % P, (a[1] - b[1]) % P)


def fp2_neg(a):
    return ((-a[0]) % P, (-a[1]) % P)


def fp2_mul(a, b):
    """fp2.rs:285-306"""
    return ((a[0] * b[0] - a[1] * b[1]) % P, (a[0] * b[1] + a[1] * b[0]) % P)


def fp2_square(a):
    """fp2.rs:164-171"""
    return ((a[0] + a[1]) * (a[0] - a[1]) % P, 2 * a[0] * a[1] % P)


def fp2_scale(a, k: int):
    """extensions.rs:86-94"""
    return (a[0] * k % P, a[1] * k % P)


def fp2_inv(a):
    """fp2.rs:355-360 (QNR = p-1, so c0^2 - QNR*c1^2 = c0^2 + c1^2)."""
    t = fp_inv((a[0] * a[0] + a[1] * a[1]) % P)
    return (a[0] * t % P, (-(a[1] * t)) % P)


def fp2_residue_mul(a):
    """x (9+u): fp2.rs:99-107"""
    return ((9 * a[0] - a[1]) % P, (a[0] + 9 * a[1]) % P)


def fp2_frobenius(a, e: int):
    """fp2.rs:119-133 -- conjugate for odd exponents."""
    return a if e % 2 == 0 else (a[0], (-a[1]) % P)


def fp2_pow(a, e: int):
    r = FP2_ONE
    for bit in bin(e)[2:]:
        r = fp2_square(r)
        if bit == "1":
            r = fp2_mul(r, a)
    return r


XI = (9, 1)
# twist constant b' = 3/(9+u): fp2.rs:42-55
TWIST_B = fp2_mul((3, 0), fp2_inv(XI))

# Frobenius coefficient tables (fp6.rs:40-179, fp12.rs:29-172), recomputed from their
# definitions xi^((p^i-1)/3), xi^((2p^i-2)/3), xi^((p^i-1)/6); checked against the reference's
# literals in tests/test_oracle_kats.py.
FROB_FP6_C1 = [fp2_pow(XI, (P**i - 1) // 3) for i in range(6)]
FROB_FP6_C2 = [fp2_pow(XI, (2 * P**i - 2) // 3) for i in range(6)]
FROB_FP12_C1 = [fp2_pow(XI, (P**i - 1) // 6) for i in range(12)]
# psi constants: g2.rs:80-109
EPS_EXP0 = fp2_pow(XI, (P - 1) // 3)
EPS_EXP1 = fp2_pow(XI, (P - 1) // 2)


# --- Fp6 = Fp2[v]/(v^3 - xi): src/fields/fp6.rs -------------------------------------------
FP6_ZERO = (FP2_ZERO, FP2_ZERO, FP2_ZERO)
FP6_ONE = (FP2_ONE, FP2_ZERO, FP2_ZERO)


def fp6_add(a, b):
    return tuple(fp2_add(x, y) for x, y in zip(a, b))


def fp6_sub(a, b):
    return tuple(fp2_sub(x, y) for x, y in zip(a, b))


def fp6_neg(a):
    return tuple(fp2_neg(x) for x in a)


def fp6_mul(a, b):
    """fp6.rs:283-367 (schoolbook over v with v^3 = xi; same value as the 36-product form)."""
    a0, a1, a2 = a
    b0, b1, b2 = b
    c0 = fp2_add(fp2_mul(a0, b0), fp2_residue_mul(fp2_add(fp2_mul(a1, b2), fp2_mul(a2, b1))))
    c1 = fp2_add(fp2_add(fp2_mul(a0, b1), fp2_mul(a1, b0)), fp2_residue_mul(fp2_mul(a2, b2)))
    c2 = fp2_add(fp2_add(fp2_mul(a0, b2), fp2_mul(a1, b1)), fp2_mul(a2, b0))
    return (c0, c1, c2)


def fp6_square(a):
    """fp6.rs:219-236 (CH-SQR2); value equals fp6_mul(a, a)."""
    return fp6_mul(a, a)


def fp6_residue_mul(a):
    """x v: fp6.rs:189-191 -> (xi*c2, c0, c1)"""
    return (fp2_residue_mul(a[2]), a[0], a[1])


def fp6_scale(a, k):
    """extensions.rs:86-94 with F = Fp2."""
    return tuple(fp2_mul(x, k) for x in a)


def fp6_inv(a):
    """fp6.rs:415-423"""
    c0, c1, c2 = a
    t0 = fp2_sub(fp2_square(c0), fp2_mul(c1, fp2_residue_mul(c2)))
    t1 = fp2_sub(fp2_residue_mul(fp2_square(c2)), fp2_mul(c0, c1))
    t2 = fp2_sub(fp2_square(c1), fp2_mul(c0, c2))
    inv = fp2_inv(
        fp2_add(fp2_residue_mul(fp2_add(fp2_mul(c2, t1), fp2_mul(c1, t2))), fp2_mul(c0, t0))
    )
    return (fp2_mul(inv, t0), fp2_mul(inv, t1), fp2_mul(inv, t2))


def fp6_frobenius(a, e: int):
    """fp6.rs:203-209"""
    return (
        fp2_frobenius(a[0], e),
        fp2_mul(fp2_frobenius(a[1], e), FROB_FP6_C1[e % 6]),
        fp2_mul(fp2_frobenius(a[2], e), FROB_FP6_C2[e % 6]),
    )


# --- Fp12 = Fp6[w]/(w^2 - v): src/fields/fp12.rs ------------------------------------------
FP12_ONE = (FP6_ONE, FP6_ZERO)
FP12_ZERO = (FP6_ZERO, FP6_ZERO)


def fp12_mul(a, b):
    """fp12.rs:229-238"""
    t0 = fp6_mul(a[0], b[0])
    t1 = fp6_mul(a[1], b[1])
    c0 = fp6_add(fp6_residue_mul(t1), t0)
    c1 = fp6_sub(fp6_sub(fp6_mul(fp6_add(a[0], a[1]), fp6_add(b[0], b[1])), t0), t1)
    return (c0, c1)


def fp12_square(a):
    """fp12.rs:536-550; value equals a*a."""
    return fp12_mul(a, a)


def fp12_inv(a):
    """fp12.rs:281-286"""
    t = fp6_inv(fp6_sub(fp6_square(a[0]), fp6_residue_mul(fp6_square(a[1]))))
    return (fp6_mul(a[0], t), fp6_neg(fp6_mul(a[1], t)))


def fp12_unitary_inverse(a):
    """fp12.rs:381-383"""
    return (a[0], fp6_neg(a[1]))


def fp12_frobenius(a, e: int):
    """fp12.rs:515-522"""
    return (fp6_frobenius(a[0], e), fp6_scale(fp6_frobenius(a[1], e), FROB_FP12_C1[e % 12]))


def fp12_sparse_mul(f, ell_0, ell_vw, ell_vv):
    """fp12.rs:426-503: f * (ell_0 + ell_vv*v^2... ) -- restated as a dense product with the
    sparse operand placed in slots 0, 2, 4 of the [z0..z5] = [c0.0,c0.1,c0.2,c1.0,c1.1,c1.2]
    view (fp12.rs:427-437): x0 = ell_0 -> c0.0, x2 = ell_vv -> c0.2, x4 = ell_vw -> c1.1."""
    sparse = ((ell_0, FP2_ZERO, ell_vv), (FP2_ZERO, ell_vw, FP2_ZERO))
    return fp12_mul(f, sparse)


def fp12_sparse_mul_as_written(f, ell_0, ell_vw, ell_vv):
    """The literal operation sequence of fp12.rs:426-503 (used to check the dense restatement)."""
    z0, z1, z2 = f[0]
    z3, z4, z5 = f[1]
    x0, x2, x4 = ell_0, ell_vv, ell_vw
    d0 = fp2_mul(z0, x0)
    d2 = fp2_mul(z2, x2)
    d4 = fp2_mul(z4, x4)
    t2 = fp2_add(z0, z4)
    t1 = fp2_add(z0, z2)
    s0 = fp2_add(fp2_add(z1, z3), z5)
    s1 = fp2_mul(z1, x2)
    t3 = fp2_add(s1, d4)
    t4 = fp2_add(fp2_residue_mul(t3), d0)
    nz0 = t4
    t3 = fp2_mul(z5, x4)
    s1 = fp2_add(s1, t3)
    t3 = fp2_add(t3, d2)
    t4 = fp2_residue_mul(t3)
    t3 = fp2_mul(z1, x0)
    s1 = fp2_add(s1, t3)
    t4 = fp2_add(t4, t3)
    nz1 = t4
    t0 = fp2_add(x0, x2)
    t3 = fp2_sub(fp2_sub(fp2_mul(t1, t0), d0), d2)
    t4 = fp2_mul(z3, x4)
    s1 = fp2_add(s1, t4)
    t3 = fp2_add(t3, t4)
    t0 = fp2_add(z2, z4)
    nz2 = t3
    t1 = fp2_add(x2, x4)
    t3 = fp2_sub(fp2_sub(fp2_mul(t0, t1), d2), d4)
    t4 = fp2_residue_mul(t3)
    t3 = fp2_mul(z3, x0)
    s1 = fp2_add(s1, t3)
    t4 = fp2_add(t4, t3)
    nz3 = t4
    t3 = fp2_mul(z5, x2)
    s1 = fp2_add(s1, t3)
    t4 = fp2_residue_mul(t3)
    t0 = fp2_add(x0, x4)
    t3 = fp2_sub(fp2_sub(fp2_mul(t2, t0), d0), d4)
    t4 = fp2_add(t4, t3)
    nz4 = t4
    t0 = fp2_add(fp2_add(x0, x2), x4)
    t3 = fp2_sub(fp2_mul(s0, t0), s1)
    nz5 = t3
    return ((nz0, nz1, nz2), (nz3, nz4, nz5))


def fp12_flatten(a):
    """12 canonical integers in the reference's nesting order c0.0.0, c0.0.1, c0.1.0, ..."""
    return [x for c in a for f2 in c for x in f2]


def fp12_unflatten(v):
    v = [int(x) for x in v]
    return (
        ((v[0], v[1]), (v[2], v[3]), (v[4], v[5])),
        ((v[6], v[7]), (v[8], v[9]), (v[10], v[11])),
    )


# --- groups: src/groups/group.rs (generic over F = Fp or Fp2) ------------------------------
class _Field:
    def __init__(self, add, sub, neg, mul, inv, zero, one, b):
        self.add, self.sub, self.neg, self.mul, self.inv = add, sub, neg, mul, inv
        self.zero, self.one, self.b = zero, one, b
        # "F::from(3) * F::curve_constant()" (group.rs:353,566)
        self.b3 = mul(self._from_int(3), b)

    def _from_int(self, k):
        return k if isinstance(self.zero, int) else (k, 0)


F1 = _Field(
    lambda a, b: (a + b) % P, lambda a, b: (a - b) % P, lambda a: (-a) % P,
    lambda a, b: a * b % P, fp_inv, 0, 1, 3,
)
F2 = _Field(fp2_add, fp2_sub, fp2_neg, fp2_mul, fp2_inv, FP2_ZERO, FP2_ONE, TWIST_B)

G1_GEN_AFF = (1, 2, False)  # g1.rs:54-60
G2_GEN_AFF = (  # g2.rs:47-77 (== EIP-197 generator)
    (
        0x1800DEEF121F1E76426A00665E5C4479674322D4F75EDADD46DEBD5CD992F6ED,
        0x198E9393920D483A7260BFB731FB5D25F1AA493335A9E71297E485B7AEF312C2,
    ),
    (
        0x12C85EA5DB8C6DEB4AAB71808DCB408FE3D1E7690C43D37B4CE6CC0166FA7DAA,
        0x090689D0585FF075EC9E99AD690C3395BC4B313370B38EF355ACDADCD122975B,
    ),
    False,
)


def proj_zero(F):
    """group.rs:310-316: (0, 1, 0)"""
    return (F.zero, F.one, F.zero)


def proj_is_zero(F, p):
    return p[2] == F.zero


def proj_from_affine(F, a):
    """group.rs:506-517"""
    x, y, inf = a
    return (x, y, F.zero if inf else F.one)


def affine_zero(F):
    """group.rs:271-277: (0, 1, inf)"""
    return (F.zero, F.one, True)


def affine_from_proj(F, p):
    """group.rs:475-495: infinity iff Z^-1 == 0."""
    inv = F.inv(p[2])
    if inv == F.zero:
        return affine_zero(F)
    return (F.mul(p[0], inv), F.mul(p[1], inv), False)


def proj_double(F, p):
    """group.rs:339-386 (RCB'15 alg. 9, a = 0)."""
    X, Y, Z = p
    t0 = F.mul(Y, Y)
    z3 = F.add(t0, t0)
    z3 = F.add(z3, z3)
    z3 = F.add(z3, z3)
    t1 = F.mul(Y, Z)
    t2 = F.mul(Z, Z)
    t2 = F.mul(F.b3, t2)
    x3 = F.mul(t2, z3)
    y3 = F.add(t0, t2)
    z3 = F.mul(t1, z3)
    t1 = F.add(t2, t2)
    t2 = F.add(t1, t2)
    t0 = F.sub(t0, t2)
    y3 = F.mul(t0, y3)
    y3 = F.add(x3, y3)
    t1 = F.mul(X, Y)
    x3 = F.mul(t0, t1)
    x3 = F.add(x3, x3)
    return proj_zero(F) if proj_is_zero(F, p) else (x3, y3, z3)


def proj_add(F, p, q):
    """group.rs:528-599 (RCB'15 alg. 7, a = 0)."""
    X1, Y1, Z1 = p
    X2, Y2, Z2 = q
    t0 = F.mul(X1, X2)
    t1 = F.mul(Y1, Y2)
    t2 = F.mul(Z1, Z2)
    t3 = F.mul(F.add(X1, Y1), F.add(X2, Y2))
    t3 = F.sub(t3, F.add(t0, t1))
    t4 = F.mul(F.add(Y1, Z1), F.add(Y2, Z2))
    t4 = F.sub(t4, F.add(t1, t2))
    y3 = F.sub(F.mul(F.add(X1, Z1), F.add(X2, Z2)), F.add(t0, t2))
    x3 = F.add(t0, t0)
    t0 = F.add(x3, t0)
    t2 = F.mul(F.b3, t2)
    z3 = F.add(t1, t2)
    t1 = F.sub(t1, t2)
    y3 = F.mul(F.b3, y3)
    x3 = F.mul(t4, y3)
    t2 = F.mul(t3, t1)
    x3 = F.sub(t2, x3)
    y3 = F.mul(y3, t0)
    t1 = F.mul(t1, z3)
    y3 = F.add(t1, y3)
    t0 = F.mul(t0, t3)
    z3 = F.mul(z3, t4)
    z3 = F.add(z3, t0)
    return (x3, y3, z3)


def proj_neg(F, p):
    return (p[0], F.neg(p[1]), p[2])


def proj_eq(F, p, q):
    """group.rs:426-447 (cross-multiplication)."""
    pz, qz = proj_is_zero(F, p), proj_is_zero(F, q)
    if pz or qz:
        return pz and qz
    return F.mul(p[0], q[2]) == F.mul(q[0], p[2]) and F.mul(p[1], q[2]) == F.mul(q[1], p[2])


def proj_scalar_mul(F, p, k: int):
    """group.rs:639-667: 256-step MSB-first NAF double-and-add; scalar is an Fp VALUE (mod p,
    NOT mod r) -- SURVEY.md N4."""
    np_, nm_ = compute_naf(k % P)
    res = proj_zero(F)
    neg = proj_neg(F, p)
    for i in range(255, -1, -1):
        res = proj_double(F, res)
        if (np_ >> i) & 1:
            res = proj_add(F, res, p)
        elif (nm_ >> i) & 1:
            res = proj_add(F, res, neg)
    return res


def g1_is_on_curve_affine(x, y):
    """g1.rs:111-132: y^2 - x^3 == 3"""
    return (y * y - x * x * x) % P == 3


def g2_is_on_curve_affine(x, y):
    """g2.rs:279-297"""
    return fp2_sub(fp2_square(y), fp2_mul(fp2_square(x), x)) == TWIST_B


def g2_endomorphism_affine(a):
    """g2.rs:140-152: psi(x,y) = (eps0 * conj(x), eps1 * conj(y)); psi(inf) = inf; the result is
    re-checked on-curve and the reference PANICS when it is not (N6)."""
    x, y, inf = a
    if inf:
        return a
    xe = fp2_mul(EPS_EXP0, fp2_frobenius(x, 1))
    ye = fp2_mul(EPS_EXP1, fp2_frobenius(y, 1))
    if not g2_is_on_curve_affine(xe, ye):
        raise RuntimeError("Endomorphism failed: NotOnCurve")
    return (xe, ye, False)


def g2_endomorphism_proj(p):
    """g2.rs:208-210"""
    return proj_from_affine(F2, g2_endomorphism_affine(affine_from_proj(F2, p)))


def g2_projective_new(v):
    """g2.rs:460-525: returns 'ok' | 'NotOnCurve' | 'NotInSubgroup'; raises like the reference's
    panic when psi lands off-curve (only possible for off-curve input)."""
    X, Y, Z = v
    lhs = fp2_mul(fp2_square(Y), Z)
    rhs = fp2_add(fp2_mul(fp2_square(X), X), fp2_mul(fp2_mul(fp2_square(Z), Z), TWIST_B))
    on_curve = lhs == rhs or Z == FP2_ZERO
    tmp = (X, Y, Z)
    a = proj_scalar_mul(F2, tmp, BLS_X)
    b = g2_endomorphism_proj(a)
    a = proj_add(F2, a, tmp)
    rhs_p = g2_endomorphism_proj(b)
    lhs_p = proj_add(F2, proj_add(F2, rhs_p, b), a)
    rhs_p = proj_add(F2, proj_double(F2, g2_endomorphism_proj(rhs_p)), proj_neg(F2, lhs_p))
    torsion_free = proj_is_zero(F2, rhs_p) and on_curve
    if not on_curve:
        return "NotOnCurve"
    return "ok" if torsion_free else "NotInSubgroup"


# --- byte formats: g1.rs:151-280, g2.rs:319-433 -------------------------------------------
def g1_to_be_bytes(a) -> bytes:
    x, y, inf = a
    if inf:
        x, y = 0, 1
    b = bytearray(fp_to_be_bytes(x) + fp_to_be_bytes(y))
    if inf:
        b[0] |= 0x80
    return bytes(b)


def g1_to_be_bytes_scrubbed(a) -> bytes:
    """g1.rs:182-192: all-zero for infinity (EVM convention)."""
    return bytes(64) if a[2] else g1_to_be_bytes(a)


def g1_from_be_bytes(b: bytes):
    """g1.rs:224-280: returns projective point or None."""
    assert len(b) == 64
    inf = (b[0] >> 7) & 1
    x = fp_from_be_bytes(bytes([b[0] & 0x7F]) + b[1:32])
    y = fp_from_be_bytes(b[32:64])
    if x is None or y is None:
        return None
    if inf:
        return proj_zero(F1) if (x == 0 and y == 1) else None
    # G1Projective::new with Z = 1 (g1.rs:383-402)
    return (x, y, 1) if g1_is_on_curve_affine(x, y) else None


def g2_to_be_bytes(a) -> bytes:
    """g2.rs:319-341: x.c1 | x.c0 | y.c1 | y.c0"""
    x, y, inf = a
    if inf:
        x, y = FP2_ZERO, FP2_ONE
    b = bytearray(
        fp_to_be_bytes(x[1]) + fp_to_be_bytes(x[0]) + fp_to_be_bytes(y[1]) + fp_to_be_bytes(y[0])
    )
    if inf:
        b[0] |= 0x80
    return bytes(b)


def g2_from_be_bytes(b: bytes):
    """g2.rs:361-433 (on-curve + subgroup via G2Projective::new)."""
    assert len(b) == 128
    inf = (b[0] >> 7) & 1
    xc1 = fp_from_be_bytes(bytes([b[0] & 0x7F]) + b[1:32])
    xc0 = fp_from_be_bytes(b[32:64])
    yc1 = fp_from_be_bytes(b[64:96])
    yc0 = fp_from_be_bytes(b[96:128])
    if None in (xc1, xc0, yc1, yc0):
        return None
    x, y = (xc0, xc1), (yc0, yc1)
    if inf:
        return proj_zero(F2) if (x == FP2_ZERO and y == FP2_ONE) else None
    try:
        ok = g2_projective_new((x, y, FP2_ONE))
    except RuntimeError:
        raise
    return (x, y, FP2_ONE) if ok == "ok" else None


# --- pairing: src/pairing.rs ------------------------------------------------------------
def g2_doubling_step(r):
    """pairing.rs:798-818. Returns (new_r, ell)."""
    X, Y, Z = r
    a = fp2_scale(fp2_mul(X, Y), TWO_INV)
    b = fp2_square(Y)
    c = fp2_square(Z)
    d = fp2_add(fp2_add(c, c), c)
    e = fp2_mul(TWIST_B, d)
    f = fp2_add(fp2_add(e, e), e)
    g = fp2_scale(fp2_add(b, f), TWO_INV)
    h = fp2_sub(fp2_square(fp2_add(Y, Z)), fp2_add(b, c))
    i = fp2_sub(e, b)
    j = fp2_square(X)
    e_sq = fp2_square(e)
    nx = fp2_mul(a, fp2_sub(b, f))
    ny = fp2_sub(fp2_square(g), fp2_add(fp2_add(e_sq, e_sq), e_sq))
    nz = fp2_mul(b, h)
    ell = (fp2_residue_mul(i), fp2_neg(h), fp2_add(fp2_add(j, j), j))
    return (nx, ny, nz), ell


def g2_addition_step(r, base):
    """pairing.rs:756-772 (base affine; its infinity flag is never consulted)."""
    X, Y, Z = r
    bx, by = base[0], base[1]
    d = fp2_sub(X, fp2_mul(Z, bx))
    e = fp2_sub(Y, fp2_mul(Z, by))
    f = fp2_square(d)
    g = fp2_square(e)
    h = fp2_mul(d, f)
    i = fp2_mul(X, f)
    j = fp2_sub(fp2_add(fp2_mul(Z, g), h), fp2_add(i, i))
    nx = fp2_mul(d, j)
    ny = fp2_sub(fp2_mul(e, fp2_sub(i, j)), fp2_mul(h, Y))
    nz = fp2_mul(Z, h)
    ell = (fp2_residue_mul(fp2_sub(fp2_mul(e, bx), fp2_mul(d, by))), d, fp2_neg(e))
    return (nx, ny, nz), ell


def affine_neg(F, a):
    """group.rs Neg for GroupAffine: (x, -y, inf)"""
    return (a[0], F.neg(a[1]), a[2])


def g2_precompute(q):
    """pairing.rs:676-708: 87 line-coefficient triples for affine Q."""
    r = proj_from_affine(F2, q)
    coeffs = []
    q_neg = affine_neg(F2, q)
    for d in ATE_LOOP_COUNT_NAF:
        r, ell = g2_doubling_step(r)
        coeffs.append(ell)
        if d == 1:
            r, ell = g2_addition_step(r, q)
            coeffs.append(ell)
        elif d == -1:
            r, ell = g2_addition_step(r, q_neg)
            coeffs.append(ell)
    q1 = g2_endomorphism_affine(q)
    q2 = affine_neg(F2, g2_endomorphism_affine(q1))
    r, ell = g2_addition_step(r, q1)
    coeffs.append(ell)
    r, ell = g2_addition_step(r, q2)
    coeffs.append(ell)
    assert len(coeffs) == 87
    return coeffs


def _line(f, c, g1):
    return fp12_sparse_mul(f, c[0], fp2_scale(c[1], g1[1]), fp2_scale(c[2], g1[0]))


def miller_loop(coeffs, g1):
    """pairing.rs:590-619"""
    f = FP12_ONE
    idx = 0
    for d in ATE_LOOP_COUNT_NAF:
        f = _line(fp12_square(f), coeffs[idx], g1)
        idx += 1
        if d != 0:
            f = _line(f, coeffs[idx], g1)
            idx += 1
    f = _line(f, coeffs[idx], g1)
    idx += 1
    f = _line(f, coeffs[idx], g1)
    return f


def glued_miller_loop(precomps, g1s):
    """pairing.rs:970-1022: zip truncates; no infinity handling (N5)."""
    pairs = list(zip(precomps, g1s))
    f = FP12_ONE
    idx = 0
    for d in ATE_LOOP_COUNT_NAF:
        f = fp12_square(f)
        for c, g1 in pairs:
            f = _line(f, c[idx], g1)
        idx += 1
        if d != 0:
            for c, g1 in pairs:
                f = _line(f, c[idx], g1)
            idx += 1
    for c, g1 in pairs:
        f = _line(f, c[idx], g1)
    idx += 1
    for c, g1 in pairs:
        f = _line(f, c[idx], g1)
    return f


def _fp4_square(a, b):
    """pairing.rs:274-284"""
    t0 = fp2_square(a)
    t1 = fp2_square(b)
    c0 = fp2_add(fp2_residue_mul(t1), t0)
    c1 = fp2_sub(fp2_sub(fp2_square(fp2_add(a, b)), t0), t1)
    return c0, c1


def cyclotomic_squared(f):
    """pairing.rs:309-350 (Granger-Scott)."""
    z0, z4, z3 = f[0]
    z2, z1, z5 = f[1]
    t0, t1 = _fp4_square(z0, z1)
    z0 = fp2_sub(t0, z0)
    z0 = fp2_add(fp2_add(z0, z0), t0)
    z1 = fp2_add(t1, z1)
    z1 = fp2_add(fp2_add(z1, z1), t1)
    t0, t1 = _fp4_square(z2, z3)
    t2, t3 = _fp4_square(z4, z5)
    z4 = fp2_sub(t0, z4)
    z4 = fp2_add(fp2_add(z4, z4), t0)
    z5 = fp2_add(t1, z5)
    z5 = fp2_add(fp2_add(z5, z5), t1)
    t0 = fp2_residue_mul(t3)
    z2 = fp2_add(t0, z2)
    z2 = fp2_add(fp2_add(z2, z2), t0)
    z3 = fp2_sub(t2, z3)
    z3 = fp2_add(fp2_add(z3, z3), t2)
    return ((z0, z4, z3), (z2, z1, z5))


def cyclotomic_exp(f, e: int):
    """pairing.rs:366-378: 256 iterations as written (leading ones square the identity)."""
    res = FP12_ONE
    for i in range(255, -1, -1):
        res = cyclotomic_squared(res)
        if (e >> i) & 1:
            res = fp12_mul(res, f)
    return res


def exp_by_neg_z(f):
    """pairing.rs:390-392"""
    return fp12_unitary_inverse(cyclotomic_exp(f, BLS_X))


def final_exponentiation(f):
    """pairing.rs:245-492 (easy part :410, hard part :437 Fuentes-Castaneda)."""
    f1 = fp12_unitary_inverse(f)
    f2 = fp12_inv(f)
    f = fp12_mul(f1, f2)
    inp = fp12_mul(fp12_frobenius(f, 2), f)
    a = exp_by_neg_z(inp)
    b = cyclotomic_squared(a)
    c = cyclotomic_squared(b)
    d = fp12_mul(c, b)
    e = exp_by_neg_z(d)
    ff = cyclotomic_squared(e)
    g = exp_by_neg_z(ff)
    h = fp12_unitary_inverse(d)
    i = fp12_unitary_inverse(g)
    j = fp12_mul(i, e)
    k = fp12_mul(j, h)
    l = fp12_mul(k, b)
    m = fp12_mul(k, e)
    n = fp12_mul(inp, m)
    o = fp12_frobenius(l, 1)
    pp = fp12_mul(o, n)
    q = fp12_frobenius(k, 2)
    r = fp12_mul(q, pp)
    s = fp12_unitary_inverse(inp)
    t = fp12_mul(s, l)
    u = fp12_frobenius(t, 3)
    return fp12_mul(u, r)


def pairing(p_proj, q_proj):
    """pairing.rs:870-893: either input infinity -> Gt identity."""
    p = affine_from_proj(F1, p_proj)
    q = affine_from_proj(F2, q_proj)
    either_zero = p[2] or q[2]
    if either_zero:
        p, q = G1_GEN_AFF, G2_GEN_AFF
    tmp = miller_loop(g2_precompute(q), p)
    if either_zero:
        tmp = FP12_ONE
    return final_exponentiation(tmp)


def glued_pairing(g1s, g2s):
    """pairing.rs:1029-1037"""
    a1 = [affine_from_proj(F1, p) for p in g1s]
    a2 = [affine_from_proj(F2, q) for q in g2s]
    return final_exponentiation(glued_miller_loop([g2_precompute(q) for q in a2], a1))


def gt_pow(g, k: int):
    """Mul<&Fr> for &Gt (gt.rs:161-187): NAF square-and-multiply, negative digits use the unitary inverse."""
    np_, nm_ = compute_naf(k)
    res = FP12_ONE
    ng = fp12_unitary_inverse(g)
    for i in range(255, -1, -1):
        res = fp12_square(res)
        if (np_ >> i) & 1:
            res = fp12_mul(res, g)
        elif (nm_ >> i) & 1:
            res = fp12_mul(res, ng)
    return res


# --- SvdW: src/svdw.rs ---------------------------------------------------------------------
def svdw_constants():
    """svdw.rs:123-153 with a = 0, b = 3, Z = 1 (find_z_svdw, svdw.rs:81-111)."""
    a, b, z = 0, 3, 1
    g = lambda x: (x * x * x + a * x + b) % P
    c1 = g(z)
    c2 = (-z * fp_inv(2)) % P
    c3 = fp_sqrt((-g(z) * (3 * z * z + 4 * a)) % P)
    assert c3 is not None
    if fp_sgn0(c3) == 1:
        c3 = (-c3) % P
    c4 = (4 * (-g(z)) * fp_inv((3 * z * z + 4 * a) % P)) % P
    return dict(a=a, b=b, z=z, c1=c1, c2=c2, c3=c3, c4=c4)


SVDW = svdw_constants()


def svdw_map_to_point(u: int):
    """svdw.rs:180-262 (RFC 9380 6.6.1 straight-line)."""
    s = SVDW
    tv1 = u * u % P * s["c1"] % P
    tv2 = (1 + tv1) % P
    tv1 = (1 - tv1) % P
    tv3 = fp_inv(tv1 * tv2 % P)
    tv4 = u * tv1 % P * tv3 % P * s["c3"] % P
    x1 = (s["c2"] - tv4) % P
    gx1 = ((x1 * x1 + s["a"]) * x1 + s["b"]) % P
    e1 = fp_is_square(gx1)
    x2 = (s["c2"] + tv4) % P
    gx2 = ((x2 * x2 + s["a"]) * x2 + s["b"]) % P
    e2 = fp_is_square(gx2) and not e1
    x3 = tv2 * tv2 % P * tv3 % P
    x3 = x3 * x3 % P * s["c4"] % P
    x3 = (x3 + s["z"]) % P
    x = x1 if e1 else x3
    x = x2 if e2 else x
    gx = ((x * x + s["a"]) * x + s["b"]) % P
    y = fp_sqrt(gx)
    if y is None:
        raise RuntimeError("SvdWError")
    if fp_sgn0(u) != fp_sgn0(y):
        y = (-y) % P
    return x, y


# --- Keccak-256 + RFC 9380 expand_message_xmd: src/hasher.rs (sha3 0.11.0-pre.4 is un-vendored;
#     Keccak-f[1600] restated from FIPS 202 with the original Keccak padding 0x01) -------------
_RC = [
    0x0000000000000001, 0x0000000000008082, 0x800000000000808A, 0x8000000080008000,
    0x000000000000808B, 0x0000000080000001, 0x8000000080008081, 0x8000000000008009,
    0x000000000000008A, 0x0000000000000088, 0x0000000080008009, 0x000000008000000A,
    0x000000008000808B, 0x800000000000008B, 0x8000000000008089, 0x8000000000008003,
    0x8000000000008002, 0x8000000000000080, 0x000000000000800A, 0x800000008000000A,
    0x8000000080008081, 0x8000000000008080, 0x0000000080000001, 0x8000000080008008,
]
_ROT = [
    [0, 36, 3, 41, 18], [1, 44, 10, 45, 2], [62, 6, 43, 15, 61], [28, 55, 25, 21, 56],
    [27, 20, 39, 8, 14],
]
_M64 = (1 << 64) - 1


def _rol(x, n):
    n %= 64
    return ((x << n) | (x >> (64 - n))) & _M64 if n else x


def keccak_f1600(A):
    for rnd in range(24):
        C = [A[x][0] ^ A[x][1] ^ A[x][2] ^ A[x][3] ^ A[x][4] for x in range(5)]
        D = [C[(x - 1) % 5] ^ _rol(C[(x + 1) % 5], 1) for x in range(5)]
        A = [[A[x][y] ^ D[x] for y in range(5)] for x in range(5)]
        B = [[0] * 5 for _ in range(5)]
        for x in range(5):
            for y in range(5):
                B[y][(2 * x + 3 * y) % 5] = _rol(A[x][y], _ROT[x][y])
        A = [[B[x][y] ^ ((~B[(x + 1) % 5][y]) & B[(x + 2) % 5][y]) for y in range(5)] for x in range(5)]
        A[0][0] ^= _RC[rnd]
    return A


def keccak256(data: bytes) -> bytes:
    rate = 136
    msg = bytearray(data)
    msg.append(0x01)
    while len(msg) % rate:
        msg.append(0)
    msg[-1] |= 0x80
    A = [[0] * 5 for _ in range(5)]
    for off in range(0, len(msg), rate):
        blk = msg[off:off + rate]
        for i in range(rate // 8):
            A[i % 5][i // 5] ^= int.from_bytes(blk[8 * i:8 * i + 8], "little")
        A = keccak_f1600(A)
    out = b"".join(A[i % 5][i // 5].to_bytes(8, "little") for i in range(4))
    return out


def _sha256(d: bytes) -> bytes:
    return hashlib.sha256(d).digest()


HASHES = {"keccak256": (keccak256, 32, 136), "sha256": (_sha256, 32, 64)}


def expand_message_xmd(msg: bytes, dst: bytes, len_in_bytes: int, hash_name="keccak256",
                       security_param=SECURITY_BITS) -> bytes:
    """hasher.rs:157-173 (new: oversize DST) + :201-250 (expand_message)."""
    H, b_in_bytes, r_in_bytes = HASHES[hash_name]
    if len(dst) > 255:
        dst = H(b"H2C-OVERSIZE-DST-" + dst)
    ell = (len_in_bytes + b_in_bytes - 1) // b_in_bytes
    dst_prime = dst + bytes([len(dst)])
    if 8 * b_in_bytes < 2 * security_param or ell > 255:
        raise ValueError("ExpandMessage")
    z_pad = bytes(r_in_bytes)
    l_i_b = len_in_bytes.to_bytes(2, "big")
    b0 = H(z_pad + msg + l_i_b + b"\x00" + dst_prime)
    bv = [H(b0 + b"\x01" + dst_prime)]
    for i in range(1, ell):
        x = bytes(p ^ q for p, q in zip(b0, bv[i - 1]))
        bv.append(H(x + bytes([i + 1]) + dst_prime))
    return b"".join(bv)[:len_in_bytes]


def hash_to_field(msg: bytes, dst: bytes = DST, count=2, size=48, hash_name="keccak256"):
    """hasher.rs:84-128: 48-byte big-endian chunks mod p."""
    em = expand_message_xmd(msg, dst, count * size, hash_name)
    return [int.from_bytes(em[size * i:size * (i + 1)], "big") % P for i in range(count)]


def hash_to_curve(msg: bytes, dst: bytes = DST):
    """g1.rs:307-331: map(u0) + map(u1) with the complete projective add."""
    u0, u1 = hash_to_field(msg, dst)
    a = svdw_map_to_point(u0)
    b = svdw_map_to_point(u1)
    assert g1_is_on_curve_affine(*a) and g1_is_on_curve_affine(*b)
    return proj_add(F1, (a[0], a[1], 1), (b[0], b[1], 1))


def sign(k: int, msg: bytes):
    """lib.rs:179-187"""
    return proj_scalar_mul(F1, hash_to_curve(msg), k)


def verify(pubkey_proj, msg: bytes, sig_proj) -> bool:
    """lib.rs:223-236: two full pairings compared in Gt."""
    h = hash_to_curve(msg)
    lhs = pairing(sig_proj, proj_from_affine(F2, G2_GEN_AFF))
    rhs = pairing(h, pubkey_proj)
    return lhs == rhs
