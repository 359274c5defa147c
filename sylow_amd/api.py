"""Host-side mirror of sylow's public items for the hot path, batch-first.

Names and argument meaning follow src/lib.rs:71-84,105-236 (`pairing`, `glued_pairing`, `sign`,
`verify`, `G1Affine`/`G1Projective`, `G2Affine`/`G2Projective`, `Gt`, `Fp`), so tests written against
the reference read the same here -- except that every object is a BATCH of n values and every call
runs on the GPU through the C ABI.  Error behaviour mirrors `GroupError` (groups/group.rs:38-47):
constructors that validate raise `GroupError` with the reference's variant names.

Fp batches are numpy uint64 [n, 4] (little-endian limbs of the canonical value, `Fp::value().to_words()`).
"""
from __future__ import annotations

import numpy as np

from .engine import Engine

P = 0x30644E72E131A029B85045B68181585D97816A916871CA8D3C208C16D87CFD47
R_ORDER = 0x30644E72E131A029B85045B68181585D2833E84879B9709143E1F593F0000001      # fp.rs:60-65
_M64 = (1 << 64) - 1
_G2 = (0x1800DEEF121F1E76426A00665E5C4479674322D4F75EDADD46DEBD5CD992F6ED,
       0x198E9393920D483A7260BFB731FB5D25F1AA493335A9E71297E485B7AEF312C2,
       0x12C85EA5DB8C6DEB4AAB71808DCB408FE3D1E7690C43D37B4CE6CC0166FA7DAA,
       0x090689D0585FF075EC9E99AD690C3395BC4B313370B38EF355ACDADCD122975B)
DST = b"WARLOCK-CHAOS-V01-CS01-SHA-256"   # src/lib.rs:90

_engine: Engine | None = None


def engine() -> Engine:
    global _engine
    if _engine is None:
        _engine = Engine(0)
    return _engine


def set_engine(e: Engine) -> None:
    global _engine
    _engine = e


class GroupError(Exception):
    """groups/group.rs:38-47"""
    NOT_ON_CURVE, NOT_IN_SUBGROUP, CANNOT_HASH_TO_GROUP, DECODE_ERROR = "NotOnCurve", "NotInSubgroup", "CannotHashToGroup", "DecodeError"


def _ints(arr) -> list:
    arr = np.asarray(arr, dtype=np.uint64).reshape(-1, 4)
    return [sum(int(arr[i, k]) << (64 * k) for k in range(4)) for i in range(arr.shape[0])]


def fp(values) -> np.ndarray:
    """Fp::new on a list of Python ints (any 256-bit value; reduced on the device like the reference)."""
    vals = [int(v) for v in np.atleast_1d(np.asarray(values, dtype=object))]
    out = np.zeros((len(vals), 4), dtype=np.uint64)
    for i, v in enumerate(vals):
        for k in range(4):
            out[i, k] = (v >> (64 * k)) & _M64
    return out


def _row(vals):
    return np.array([[(v >> (64 * k)) & _M64 for v in vals for k in range(4)]], dtype=np.uint64)


def _raise_status(st):
    """per-element status bytes (include/sylow_hip.h) -> the reference's error for the first failing element"""
    names = {1: GroupError.NOT_ON_CURVE, 2: GroupError.NOT_IN_SUBGROUP, 3: GroupError.CANNOT_HASH_TO_GROUP, 4: GroupError.DECODE_ERROR}
    bad = np.nonzero(np.asarray(st))[0]
    if len(bad):
        raise GroupError(names[int(st[bad[0]])])


def fp_from_be_bytes(blobs):
    """Fp::from_be_bytes (fp.rs:686-719): (values [n, 4], is_some [n]) -- both halves of the CtOption (value = v mod p)."""
    v, st = engine().fp_from_be_bytes(list(blobs))
    return v, st == 0


def fp_to_be_bytes(values):                         # fp.rs:727-737
    return engine().fp_to_be_bytes(values)


class _Points:
    WIDTH = 0

    def __init__(self, xy: np.ndarray, infinity: np.ndarray | None = None):
        self.xy = np.ascontiguousarray(xy, dtype=np.uint64).reshape(-1, self.WIDTH)
        n = self.xy.shape[0]
        self.infinity = np.zeros(n, dtype=np.uint8) if infinity is None else np.ascontiguousarray(infinity, dtype=np.uint8).reshape(n)

    def __len__(self):
        return self.xy.shape[0]

    def __eq__(self, other):                       # GroupAffine ct_eq (group.rs:226-235), elementwise
        both_inf = (self.infinity & other.infinity).astype(bool)
        same = (~self.infinity.astype(bool)) & (~other.infinity.astype(bool)) & (self.xy == other.xy).all(axis=1)
        return both_inf | same

    def is_zero(self):
        return self.infinity.astype(bool)



def _random_scalars(n, seed):
    """n scalars in [0, r) as [n, 4] limbs.  seed None: the operating system's generator (what the reference's `rand(&mut OsRng)` call
    sites use); an int: the reproducible xoshiro test stream -- NOT for secrets or nonces."""
    if seed is None:
        import secrets
        return fp([secrets.randbelow(R_ORDER) for _ in range(n)])
    return engine().fr_add(engine().xoshiro_fp_soa(seed, n).T.copy(), np.zeros((n, 4), dtype=np.uint64))

class G1Affine(_Points):
    """Batch of G1 points in affine form (g1.rs:30); identity = (0, 1, infinity)."""
    WIDTH = 8

    @classmethod
    def generator(cls, n=1):                       # g1.rs:54-60
        return cls(np.repeat(_row([1, 2]), n, 0))

    @classmethod
    def zero(cls, n=1):
        return cls(np.repeat(_row([0, 1]), n, 0), np.ones(n, dtype=np.uint8))

    @classmethod
    def hash_to_curve(cls, msgs, dst: bytes = DST):     # g1.rs:307-331 with XMDExpander<Keccak256>(dst, 128)
        xy, inf = engine().hash_to_g1(list(msgs), dst)
        return cls(xy, inf)

    def __mul__(self, k):                          # Mul<&Fp> (group.rs:639-667)
        xy, inf = engine().g1_scalar_mul(self.xy, k, self.infinity)
        return G1Affine(xy, inf)

    def __add__(self, other):                      # Add (group.rs:528-599)
        xy, inf = engine().g1_add(self.xy, other.xy, self.infinity, other.infinity)
        return G1Affine(xy, inf)

    def __sub__(self, other):                      # Sub (group.rs:614-624): self + (-other)
        xy, inf = engine().g1_sub(self.xy, other.xy, self.infinity, other.infinity)
        return G1Affine(xy, inf)

    def __neg__(self):
        y = engine().fp_neg(self.xy[:, 4:])
        return G1Affine(np.concatenate([self.xy[:, :4], y], axis=1), self.infinity)

    def double(self):                              # GroupProjective::double (group.rs:339-386)
        xy, inf = engine().g1_double(self.xy, self.infinity)
        return G1Affine(xy, inf)

    @classmethod
    def rand(cls, n=1, seed=None):                 # GroupTrait::rand (g1.rs:293-305): generator * random scalar; an int seed is test-only
        xy, inf = engine().g1_generator_mul(_random_scalars(n, seed))
        return cls(xy, inf)

    def to_be_bytes(self):                         # g1.rs:151-180
        return engine().g1_to_be_bytes(self.xy, self.infinity)

    @classmethod
    def from_be_bytes(cls, blobs):                 # g1.rs:224-280: CtOption none -> GroupError(DecodeError / NotOnCurve)
        xy, inf, st = engine().g1_from_be_bytes(list(blobs))
        _raise_status(st)
        return cls(xy, inf)


G1Projective = G1Affine   # results are compared after normalisation: one batch type serves both names


class G2Affine(_Points):
    WIDTH = 16
    in_subgroup = False                            # set on values whose r-torsion membership is established

    def _checked(self, flag=True):
        self.in_subgroup = bool(flag)
        return self

    @classmethod
    def generator(cls, n=1):                       # g2.rs:47-77
        return cls(np.repeat(_row(_G2), n, 0))._checked()

    @classmethod
    def zero(cls, n=1):
        return cls(np.repeat(_row([0, 0, 1, 0]), n, 0), np.ones(n, dtype=np.uint8))._checked()

    @classmethod
    def new(cls, xy):                              # G2Projective::new (g2.rs:460-525): on-curve + subgroup
        pts = cls(xy)
        st = engine().g2_subgroup_check(pts.xy, pts.infinity)
        if (st == 1).any():
            raise GroupError(GroupError.NOT_ON_CURVE)
        if (st == 2).any():
            raise GroupError(GroupError.NOT_IN_SUBGROUP)
        return pts._checked()

    def __mul__(self, k):
        # values built the way the reference allows (generator, new, from_be_bytes, and what the group law makes of them) are in
        # the r-torsion and take the endomorphism-split product; a raw G2Affine(xy) is treated as an arbitrary twist point
        xy, inf = engine().g2_scalar_mul(self.xy, k, self.infinity, subgroup=self.in_subgroup)
        return G2Affine(xy, inf)._checked(self.in_subgroup)

    def __neg__(self):
        y = np.concatenate([engine().fp_neg(self.xy[:, 8:12]), engine().fp_neg(self.xy[:, 12:16])], axis=1)
        return G2Affine(np.concatenate([self.xy[:, :8], y], axis=1), self.infinity)._checked(self.in_subgroup)

    def __add__(self, other):
        xy, inf = engine().g2_add(self.xy, other.xy, self.infinity, other.infinity)
        return G2Affine(xy, inf)._checked(self.in_subgroup and other.in_subgroup)

    def __sub__(self, other):                      # Sub (group.rs:614-624)
        xy, inf = engine().g2_sub(self.xy, other.xy, self.infinity, other.infinity)
        return G2Affine(xy, inf)._checked(self.in_subgroup and other.in_subgroup)

    def double(self):
        xy, inf = engine().g2_double(self.xy, self.infinity)
        return G2Affine(xy, inf)._checked(self.in_subgroup)

    def precompute(self) -> "G2PreComputed":        # pairing.rs:676
        return G2PreComputed(self)

    def endomorphism(self) -> "G2Affine":           # GroupTrait::endomorphism = psi (g2.rs:140-152); panics upstream if the image is off-curve
        xy, inf, st = engine().g2_psi(self.xy, self.infinity)
        _raise_status(st)
        return G2Affine(xy, inf)._checked(self.in_subgroup)

    @classmethod
    def rand(cls, n=1, seed=None):                 # GroupTrait::rand (g2.rs:204-240): a random r-torsion point; an int seed is test-only
        xy, inf = engine().g2_generator_mul(_random_scalars(n, seed))
        return cls(xy, inf)._checked()

    def to_be_bytes(self):                         # g2.rs:319-359
        return engine().g2_to_be_bytes(self.xy, self.infinity)

    @classmethod
    def from_be_bytes(cls, blobs):                 # g2.rs:361-433 (decode + on-curve + subgroup)
        xy, inf, st = engine().g2_from_be_bytes(list(blobs))
        _raise_status(st)
        return cls(xy, inf)._checked()


G2Projective = G2Affine



class _Ext:
    """Batch of extension-field elements, canonical limbs [n, 4 * degree] in the reference's nesting order; the operators of
    FieldExtension<D, N, F> (extensions.rs:41-238) run on the GPU."""
    DEGREE = 0

    def __init__(self, v):
        self.v = np.ascontiguousarray(v, dtype=np.uint64).reshape(-1, 4 * self.DEGREE)

    def __len__(self): return self.v.shape[0]
    def __eq__(self, o): return (self.v == o.v).all(axis=1)
    def __add__(self, o): return type(self)(engine().fext_op("add", self.v, o.v))
    def __sub__(self, o): return type(self)(engine().fext_op("sub", self.v, o.v))
    def __neg__(self): return type(self)(engine().fext_op("neg", self.v))
    def scale(self, k): return type(self)(engine().fext_op("scale", self.v, np.ascontiguousarray(k, dtype=np.uint64).reshape(-1, 4)))


class Fp2(_Ext):
    DEGREE = 2
    def __mul__(self, o): return Fp2(engine().fp2_mul(self.v, o.v))
    def square(self): return Fp2(engine().fp2_sqr(self.v))
    def inv(self): return Fp2(engine().fp2_inv(self.v))
    def residue_mul(self): return Fp2(engine().fp2_residue_mul(self.v))          # x (9 + u), fp2.rs:99-107
    def frobenius(self, exponent: int): return Fp2(engine().fp2_frobenius(self.v, exponent))   # fp2.rs:119-133


class Fp6(_Ext):
    DEGREE = 6
    def __mul__(self, o): return Fp6(engine().fp6_mul(self.v, o.v))
    def square(self): return Fp6(engine().fp6_sqr(self.v))                       # fp6.rs:213-236
    def inv(self): return Fp6(engine().fp6_inv(self.v))
    def residue_mul(self): return Fp6(engine().fp6_residue_mul(self.v))          # x v, fp6.rs:189-192
    def frobenius(self, exponent: int): return Fp6(engine().fp6_frobenius(self.v, exponent))   # fp6.rs:205-211


class Fp12(_Ext):
    DEGREE = 12
    def __mul__(self, o): return Fp12(engine().fp12_mul(self.v, o.v))
    def square(self): return Fp12(engine().fp12_sqr(self.v))
    def inv(self): return Fp12(engine().fp12_inv(self.v))
    def frobenius(self, exponent: int): return Fp12(engine().fp12_frobenius(self.v, exponent))  # exponent in {1, 2, 3}
    def sparse_mul(self, ell): return Fp12(engine().fp12_sparse_mul(self.v, ell))               # fp12.rs:426-503, ell = [n, 24]


class Gt:
    """Batch of target-group elements (groups/gt.rs): 12 Fp each; `+` is the group law (Fp12 product)."""

    def __init__(self, v: np.ndarray):
        self.v = np.ascontiguousarray(v, dtype=np.uint64).reshape(-1, 48)

    @classmethod
    def identity(cls, n=1):                        # gt.rs:263-265
        v = np.zeros((n, 48), dtype=np.uint64)
        v[:, 0] = 1
        return cls(v)

    def __len__(self):
        return self.v.shape[0]

    def __eq__(self, other):                       # gt.rs:139-159
        return (self.v == other.v).all(axis=1)

    def __add__(self, other):                      # gt.rs: Add = Fp12 multiplication
        return Gt(engine().fp12_mul(self.v, other.v))

    def __neg__(self):                             # gt.rs:107-114: unitary inverse (conjugate)
        v = self.v.copy()
        v[:, 24:] = np.concatenate([engine().fp_neg(self.v[:, 24 + 4 * j:28 + 4 * j]) for j in range(6)], axis=1)
        return Gt(v)

    def __mul__(self, k):                          # Mul<&Fr> (gt.rs:161-187); k: Fr values [n, 4]
        return Gt(engine().gt_pow(self.v, k))


class MillerLoopResult:
    """Batch of raw Miller values (pairing.rs:72): public but NOT unique -- the engine replays the reference's
    line formulas and digit schedule, so these match `G2PreComputed::miller_loop` bit for bit."""

    def __init__(self, v: np.ndarray):
        self.v = np.ascontiguousarray(v, dtype=np.uint64).reshape(-1, 48)

    def __eq__(self, other):
        return (self.v == other.v).all(axis=1)

    def __mul__(self, other):                      # Mul<&MillerLoopResult> (pairing.rs:98-128): Fp12 product
        return MillerLoopResult(engine().fp12_mul(self.v, other.v))

    def final_exponentiation(self) -> "Gt":        # pairing.rs:245-492
        return Gt(engine().final_exp(self.v))


class Fr:
    """Batch of scalar-field elements (fp.rs:556-565), canonical limbs [n, 4]; arithmetic runs on the GPU."""

    def __init__(self, v):
        self.v = np.ascontiguousarray(v, dtype=np.uint64).reshape(-1, 4)

    @classmethod
    def from_ints(cls, values):
        return cls(fp(values))

    def __len__(self): return self.v.shape[0]
    def __add__(self, o): return Fr(engine().fr_add(self.v, o.v))
    def __sub__(self, o): return Fr(engine().fr_sub(self.v, o.v))
    def __mul__(self, o): return Fr(engine().fr_mul(self.v, o.v))
    def __neg__(self): return Fr(engine().fr_neg(self.v))
    def inv(self): return Fr(engine().fr_inv(self.v))
    def __eq__(self, o): return (self.v == o.v).all(axis=1)

    @classmethod
    def from_be_bytes(cls, blobs):                 # Fr::from_be_bytes (fp.rs:746-778): none for v >= r
        v, st = engine().fr_from_be_bytes(list(blobs))
        _raise_status(st)
        return cls(v)

    def to_be_bytes(self):
        return engine().fr_to_be_bytes(self.v)

    @classmethod
    def rand(cls, n=1, seed=None):                 # FieldExtensionTrait::rand; seed None = OS generator, an int = reproducible test stream
        return cls(_random_scalars(n, seed))


def aggregate(points: "G1Affine", weights: "Fr", n_jobs: int, n_terms: int) -> "G1Affine":
    """sum_i weights[j,i] * points[j,i] per job (examples/threshold_signing.rs:124-143); rows are term-major
    (row i*n_jobs + j is term i of job j)."""
    xy, inf = engine().g1_lincomb(points.xy, weights.v, n_jobs, n_terms, points.infinity)
    return G1Affine(xy, inf)


class G2PreComputed:
    """G2Affine::precompute() (pairing.rs:556,676-708): q plus the 87 line-coefficient triples, [n, 87*24] words."""

    def __init__(self, q: "G2Affine"):
        self.q = q
        self.coeffs = engine().g2_precompute(q.xy)

    def miller_loop(self, g1: G1Affine, table_idx=None) -> MillerLoopResult:     # pairing.rs:590-619
        """Consumes the cached tables (no G2 arithmetic): element i pairs g1[i] with table table_idx[i] (default: table i), so
        ONE precomputed key serves any number of G1 points."""
        if table_idx is None and len(g1) != len(self.q):
            raise ValueError(f"{len(g1)} G1 points against {len(self.q)} precomputed tables: pass table_idx (one table index per point)")
        return MillerLoopResult(engine().miller_loop_precomputed(self.coeffs, g1.xy, table_idx))


def glued_miller_loop(g2s, g1s: G1Affine, offsets=None) -> MillerLoopResult:
    """glued_miller_loop(&[G2PreComputed], &[G1Affine]) -> MillerLoopResult (pairing.rs:970-1022).  `g2s` is a G2PreComputed or
    G2Affine batch; without `offsets` the whole batch is one job."""
    q = g2s.q if isinstance(g2s, G2PreComputed) else g2s
    if offsets is None:
        offsets = [0, min(len(g1s), len(q))]               # zip truncates (pairing.rs:975)
    if isinstance(g2s, G2PreComputed) and len(g1s) == len(q):          # the cached tables are consumed as they are
        return MillerLoopResult(engine().glued_miller_loop_precomputed(g2s.coeffs, g1s.xy, offsets))
    return MillerLoopResult(engine().glued_miller_loop(g1s.xy, q.xy, offsets))


def pairing(p: G1Affine, q: G2Affine) -> Gt:
    """pairing(&G1Projective, &G2Projective) -> Gt (pairing.rs:870-893), n independent values."""
    return Gt(engine().pairing(p.xy, q.xy, p.infinity, q.infinity))


def glued_pairing(g1s: G1Affine, g2s: G2Affine, offsets=None, evm_infinity: bool = False) -> Gt:
    """glued_pairing(&[G1Projective], &[G2Projective]) -> Gt (pairing.rs:1029-1037).  Without `offsets` the
    whole batch is ONE product (the reference's shape); with offsets, job j multiplies pairs
    [offsets[j], offsets[j+1])."""
    if offsets is None:            # one product over the whole batch: spread over the GPU, one final exponentiation
        gt, _ = engine().pairing_product(g1s.xy, g2s.xy, g1s.infinity, g2s.infinity, skip_infinity=evm_infinity)
        return Gt(gt)
    gt, _ = engine().multi_pairing(g1s.xy, g2s.xy, offsets, g1s.infinity, g2s.infinity, skip_infinity=evm_infinity)
    return Gt(gt)


def verify_same_signer(pubkey: G2Affine, msgs, sig: G1Affine) -> np.ndarray:
    """examples/verify_multiple_messages_same_signer.rs:41-60: one key, many (message, signature) pairs."""
    assert len(pubkey) == 1
    return engine().bls_verify_same_signer(pubkey.xy, list(msgs), sig.xy, pubkey.infinity, sig.infinity).astype(bool)


def aggregate_verify(pubkey: G2Affine, msgs, sig: G1Affine) -> bool:
    """The batch check of examples/verify_multiple_messages_same_signer.rs:41-60 / threshold_signing.rs:92-121: the product of the
    2n pairs (sig_i, G2gen), (-H(msg_i), pk_i) == Gt::identity(), as one boolean; `pubkey` holds one key per message or ONE key."""
    _, ok = engine().bls_aggregate_verify(pubkey.xy, list(msgs), sig.xy, pubkey.infinity, sig.infinity)
    return bool(ok)


class KeyTable:
    """One signer's `G2PreComputed` kept ON THE DEVICE across calls (examples/verify_multiple_messages_same_signer.rs:41-60):
    built once, then every `verify` is two table-driven Miller loops with no G2 arithmetic."""

    def __init__(self, pubkey: G2Affine):
        assert len(pubkey) == 1
        self.infinity = pubkey.infinity
        self.table = engine().g2_line_table(pubkey.xy)

    def verify(self, msgs, sig: G1Affine) -> np.ndarray:
        return engine().bls_verify_line_table(self.table, list(msgs), sig.xy, self.infinity, sig.infinity).astype(bool)


def batch_verify(pubkey: G2Affine, msgs, sig: G1Affine, weight_bits: int = 128, seed=None) -> bool:
    """Sound one-boolean batch verification (the small-exponent test): prod_i [e(sig_i, G2gen) e(-H(m_i), pk_i)]^(w_i) == identity
    with fresh non-zero `weight_bits`-bit weights from the operating system's generator (an int `seed` draws reproducible weights:
    tests only).  True when every signature is valid; a batch with an invalid one passes with probability <= 2^-weight_bits PROVIDED
    the keys lie in G2 proper: keys whose r-torsion membership is not established (a G2Affine built from raw coordinates) go through
    the subgroup check first and a failing one raises, exactly where G2Projective::new (g2.rs:460-525) would have refused the key."""
    if not 1 <= int(weight_bits) <= 128:
        raise ValueError("weight_bits must be in 1..128")
    if not pubkey.in_subgroup:
        st = engine().g2_subgroup_check(pubkey.xy, pubkey.infinity)
        if np.any(st != 0):
            raise ValueError(f"batch_verify: public key {int(np.flatnonzero(st != 0)[0])} is not a point of G2 (status {int(st[st != 0][0])})")
    n = len(sig)
    mask = (1 << weight_bits) - 1
    if seed is None:
        import secrets
        w = []
        while len(w) < n:                                  # uniform over [1, 2^weight_bits): zero is redrawn, no bit is forced
            v = secrets.randbits(weight_bits)
            if v:
                w.append(v)
        w = fp(w)
    else:
        w = fp([(int(v) & mask) or 1 for v in _ints(engine().xoshiro_fp_soa(seed, n).T)])
    _, ok = engine().bls_batch_verify_weighted(pubkey.xy, list(msgs), sig.xy, w, pubkey.infinity, sig.infinity)
    return ok


class KeyPair:
    """KeyPair (lib.rs:105-137), a batch of them: secret_key = Fp::new(Fr::rand().value()) -- a scalar below r held as an Fp --
    and public_key = G2Projective::generator() * secret_key."""

    def __init__(self, secret_key: np.ndarray, public_key: "G2Affine"):
        self.secret_key, self.public_key = secret_key, public_key

    @classmethod
    def generate(cls, n=1, seed=None):
        """`seed` None: the operating system's generator like the reference's OsRng; an int: reproducible (tests)."""
        sk = _random_scalars(n, seed)
        xy, inf = engine().g2_generator_mul(sk)          # fixed-base table of the generator
        return cls(sk, G2Affine(xy, inf)._checked())

    def __len__(self):
        return self.secret_key.shape[0]


def sign(k, msgs) -> G1Affine:
    """sign(&Fp, &[u8]) (lib.rs:179-187): H(msg) * k."""
    xy, inf = engine().bls_sign(k, list(msgs))
    return G1Affine(xy, inf)


def verify(pubkey: G2Affine, msgs, sig: G1Affine) -> np.ndarray:
    """verify(&G2Projective, &[u8], &G1Projective) (lib.rs:223-236): elementwise bool."""
    return engine().bls_verify(pubkey.xy, list(msgs), sig.xy, pubkey.infinity, sig.infinity).astype(bool)
