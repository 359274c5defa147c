"""CPU tests: both oracles (C restatement + Python big-int restatement) against every literal
known-answer vector the reference's own tests hold for the hot path, and against
each other.  These pin the oracle; the GPU tests then compare the HIP path with the oracle."""
import numpy as np
import pytest

from helpers import P, SEED, Xoshiro, ints, limbs, pack
from oracle import pyref as R

I = lambda s: int(s, 16)
G2_PROJ = list(R.G2_GEN_AFF[0]) + list(R.G2_GEN_AFF[1]) + [1, 0]


def test_fp_kats_both_oracles(kats, coracle):
    for name, op, f in (("fp_add", "add", lambda a, b: (a + b) % P), ("fp_sub", "sub", lambda a, b: (a - b) % P),
                        ("fp_mul", "mul", lambda a, b: a * b % P)):
        for a, b, c in kats[name]["cases"]:
            assert ints(coracle.fp_op(op, limbs([I(a)]), limbs([I(b)])))[0] == I(c), (name, a, b)
            assert f(I(a) % P, I(b) % P) == I(c)
    # Fp::new reduces (fp.rs:878-884); from_be_bytes rejects >= p (fp.rs:821-826)
    assert R.fp_new(P) == 0
    assert R.fp_from_be_bytes((P - 1 + 10).to_bytes(32, "big")) is None
    assert R.fp_from_be_bytes((P - 1).to_bytes(32, "big")) == P - 1
    # inv(0) = 0 (fp.rs:1126-1132)
    assert ints(coracle.fp_op("inv", limbs([0, 1, 2])))[:2] == [0, 1]


def test_fp2_fp6_kats(kats, coracle):
    for a, b, c in kats["fp2_mul"]["cases"]:
        A, B, Cc = [I(x) for x in a], [I(x) for x in b], [I(x) for x in c]
        assert ints(coracle.fp2_op("mul", pack(A, 8), pack(B, 8))) == Cc
        assert list(R.fp2_mul(tuple(A), tuple(B))) == Cc
    for a, b, c in kats["fp2_div"]["cases"]:
        A, B, Cc = [I(x) for x in a], [I(x) for x in b], [I(x) for x in c]
        assert ints(coracle.fp2_op("div", pack(A, 8), pack(B, 8))) == Cc
        assert list(R.fp2_mul(tuple(A), R.fp2_inv(tuple(B)))) == Cc
    f6 = lambda v: ((v[0], v[1]), (v[2], v[3]), (v[4], v[5]))
    flat6 = lambda a: [x for c in a for x in c]
    for a, b, c in kats["fp6_mul"]["cases"]:
        A, B, Cc = [I(x) for x in a], [I(x) for x in b], [I(x) for x in c]
        assert ints(coracle.fp6_op("mul", pack(A, 24), pack(B, 24))) == Cc
        assert flat6(R.fp6_mul(f6(A), f6(B))) == Cc
    for a, b, c in kats["fp6_div"]["cases"]:
        A, B, Cc = [I(x) for x in a], [I(x) for x in b], [I(x) for x in c]
        assert ints(coracle.fp6_op("div", pack(A, 24), pack(B, 24))) == Cc
        assert flat6(R.fp6_mul(f6(A), R.fp6_inv(f6(B)))) == Cc


def test_constants_against_reference_literals(kats, coracle):
    flat2 = lambda t: [x for c in t for x in c]
    nonunit = lambda v: [x for x in v if x not in (0, 1)]
    for which, name, tbl in ((0, "FROBENIUS_COEFF_FP6_C1", R.FROB_FP6_C1), (1, "FROBENIUS_COEFF_FP6_C2", R.FROB_FP6_C2),
                             (2, "FROBENIUS_COEFF_FP12_C1", R.FROB_FP12_C1)):
        lit = nonunit([I(x) for x in kats[name]["nonunit_words"]])
        assert nonunit(flat2(tbl)) == lit, name
        assert ints(coracle.constants(which)) == flat2(tbl), name
    for which, key, val in ((3, "const_FP2_TWIST_CURVE_CONSTANT", R.TWIST_B), (4, "const_EPS_EXP0", R.EPS_EXP0),
                            (5, "const_EPS_EXP1", R.EPS_EXP1)):
        lit = [I(x) for x in kats[key]["value"]]
        assert list(val) == lit and ints(coracle.constants(which)) == lit, key
    assert [I(kats["const_TWO_INV"]["value"][0])] == [R.TWO_INV] == ints(coracle.constants(7))
    assert I(kats["const_P_MINUS_1_OVER_2"]["value"][0]) == (P - 1) // 2
    assert I(kats["const_P_MINUS_3_OVER_4"]["value"][0]) == (P - 3) // 4
    assert I(kats["const_BLS_X"]["value"][0]) == R.BLS_X
    assert list(R.G2_GEN_AFF[0]) == [I(x) for x in kats["const_G2_X"]["value"]]
    assert list(R.G2_GEN_AFF[1]) == [I(x) for x in kats["const_G2_Y"]["value"]]
    s = kats["svdw_constants"]
    assert [R.SVDW[n] for n in ("c1", "c2", "c3", "c4", "z")] == [I(s[n]) for n in ("c1", "c2", "c3", "c4", "z")] == ints(coracle.constants(6))
    assert kats["ate_loop_count_naf"]["value"] == R.ATE_LOOP_COUNT_NAF
    assert kats["dst"].encode() == R.DST


def test_gt_generator_and_pairing_kat(kats, coracle):
    g1 = pack([1, 2, 1], 12)
    g2 = pack(G2_PROJ, 24)
    gen = [I(x) for x in kats["gt_generator"]["value"]]
    assert ints(coracle.pairing(g1, g2)) == gen                                            # pairing.rs:1052-1057
    assert R.fp12_flatten(R.pairing((1, 2, 1), R.proj_from_affine(R.F2, R.G2_GEN_AFF))) == gen
    pk = kats["pairing_kat"]                                                               # pairing.rs:1122-1189
    a, b = I(pk["a"]), I(pk["b"])
    p = coracle.g1_scalar_mul(g1, limbs([a]))
    q = coracle.g2_scalar_mul(g2, limbs([b]))
    exp = [I(x) for x in pk["gt"]]
    assert ints(coracle.pairing(p, q)) == exp
    pr = R.proj_scalar_mul(R.F1, (1, 2, 1), a)
    qr = R.proj_scalar_mul(R.F2, R.proj_from_affine(R.F2, R.G2_GEN_AFF), b)
    assert R.fp12_flatten(R.pairing(pr, qr)) == exp
    # the projective representatives themselves agree between the two restatements (same formulas)
    assert ints(p) == list(pr) and ints(q) == [x for c in qr for x in c]


def test_pairing_identities_and_infinity(coracle):
    """pairing.rs:1101-1120, and the glued-loop infinity behaviour documented in SURVEY.md N5."""
    one = R.fp12_flatten(R.FP12_ONE)
    g1, g2 = pack([1, 2, 1], 12), pack(G2_PROJ, 24)
    assert ints(coracle.pairing(pack([0, 1, 0], 12), g2)) == one
    assert ints(coracle.pairing(g1, pack([0, 0, 1, 0, 0, 0], 24))) == one
    # glued with P = inf: that pair is neutral; with Q = inf: accumulator collapses to 0 (reference defect)
    gp = coracle.glued_pairing(np.concatenate([g1, pack([0, 1, 0], 12)]), np.concatenate([g2, g2]), [0, 2])
    assert ints(gp) == ints(coracle.pairing(g1, g2))
    gq = coracle.glued_pairing(np.concatenate([g1, g1]), np.concatenate([g2, pack([0, 0, 1, 0, 0, 0], 24)]), [0, 2])
    assert ints(gq) == [0] * 12
    assert ints(coracle.glued_pairing(g1[:0], g2[:0], [0, 0])) == one                     # pairing.rs:1218-1219


def test_c_oracle_vs_python_oracle_random(coracle):
    rng = Xoshiro(SEED + 10)
    f = [rng.fp() for _ in range(12)]
    g = [rng.fp() for _ in range(12)]
    F, G = R.fp12_unflatten(f), R.fp12_unflatten(g)
    assert ints(coracle.fp12_op("mul", pack(f, 48), pack(g, 48))) == R.fp12_flatten(R.fp12_mul(F, G))
    assert ints(coracle.fp12_op("sqr", pack(f, 48))) == R.fp12_flatten(R.fp12_square(F))
    assert ints(coracle.fp12_op("inv", pack(f, 48))) == R.fp12_flatten(R.fp12_inv(F))
    for e in (1, 2, 3):
        assert ints(coracle.fp12_op("frobenius", pack(f, 48), arg=e)) == R.fp12_flatten(R.fp12_frobenius(F, e))
    ell = [(rng.fp(), rng.fp()) for _ in range(3)]
    exp = R.fp12_sparse_mul_as_written(F, *ell)
    assert exp == R.fp12_sparse_mul(F, *ell)
    assert ints(coracle.fp12_sparse_mul(pack(f, 48), pack([x for e in ell for x in e], 24))) == R.fp12_flatten(exp)
    a, b = rng.fp(), rng.fp()
    p1 = R.affine_from_proj(R.F1, R.proj_scalar_mul(R.F1, (1, 2, 1), a))
    q1 = R.affine_from_proj(R.F2, R.proj_scalar_mul(R.F2, R.proj_from_affine(R.F2, R.G2_GEN_AFF), b))
    ml = R.miller_loop(R.g2_precompute(q1), p1)
    mlc = coracle.miller_loop(pack([p1[0], p1[1]], 8), pack(list(q1[0]) + list(q1[1]), 16))
    assert ints(mlc) == R.fp12_flatten(ml)
    fe = R.final_exponentiation(ml)
    assert ints(coracle.final_exponentiation(mlc)) == R.fp12_flatten(fe)
    assert ints(coracle.fp12_op("cyclotomic_squared", pack(R.fp12_flatten(fe), 48))) == R.fp12_flatten(R.cyclotomic_squared(fe))
    # precompute table: 87 triples
    co = coracle.g2_precompute(pack(list(q1[0]) + list(q1[1]), 16))
    assert ints(co) == [x for ell3 in R.g2_precompute(q1) for e in ell3 for x in e]
    # G2 subgroup check agrees: generator multiples are in the subgroup
    assert coracle.g2_projective_new(pack(list(q1[0]) + list(q1[1]) + [1, 0], 24))[0] == 0
    assert R.g2_projective_new((q1[0], q1[1], R.FP2_ONE)) == "ok"


def test_keccak_and_xmd(kats, coracle):
    # public Keccak-256 KATs (the reference holds none: parity of this leg is unpinned upstream)
    assert R.keccak256(b"").hex() == "c5d2460186f7233c927e7db2dcc703c0e500b653ca82273b7bfad8045d85a470"
    assert R.keccak256(b"abc").hex() == "4e03657aea45a94fc7d47ba826c8d667c0d1e6e33a64a036ec44f58fa12d6c45"
    for m in (b"", b"abc", bytes(range(135)), bytes(range(136)), bytes(range(137)), b"x" * 1000):
        assert coracle.keccak256(m) == R.keccak256(m)
    # RFC 9380 expand_message_xmd(SHA-256) vectors from hasher.rs:345-388 through the same XMD routine
    x = kats["xmd_sha256"]
    for fn, dst, ln in zip(("short_xmd_hashmap", "long_xmd_hashmap"), x["dsts"], x["lens"]):
        for msg, exp in x[fn]["pairs"]:
            assert R.expand_message_xmd(msg.encode(), dst.encode(), ln, "sha256").hex() == exp
    for m in (b"", b"abc", bytes(range(200)), (20).to_bytes(4, "big")):
        assert coracle.expand_message_xmd_keccak(m, R.DST, 96) == R.expand_message_xmd(m, R.DST, 96)
    assert coracle.expand_message_xmd_keccak(b"abc", b"d" * 300, 96) == R.expand_message_xmd(b"abc", b"d" * 300, 96)


def test_svdw_hash_sign_verify(coracle):
    rng = Xoshiro(SEED + 11)
    u = [rng.fp() for _ in range(16)] + [0, 1, P - 1]
    got = ints(coracle.svdw_map(limbs(u)))
    for i, ui in enumerate(u):
        x, y = R.svdw_map_to_point(ui)
        assert (got[2 * i], got[2 * i + 1]) == (x, y)
        assert R.g1_is_on_curve_affine(x, y)                     # svdw.rs:317-351
    msgs = [b"", (20).to_bytes(4, "big"), b"hello world", bytes(range(250))]
    h = coracle.hash_to_curve(msgs)
    for i, m in enumerate(msgs):
        assert ints(h[i:i + 1]) == list(R.hash_to_curve(m))
    sk = [rng.fp() for _ in msgs]
    sig = coracle.sign(limbs(sk), msgs)
    assert ints(sig[:1]) == list(R.sign(sk[0], msgs[0]))
    g2 = pack(G2_PROJ, 24)
    pk = coracle.g2_scalar_mul(np.repeat(g2, len(msgs), 0), limbs(sk))
    assert coracle.verify(pk, msgs, sig).tolist() == [1, 1, 1, 1]                          # pairing.rs:1059-1072
    assert coracle.verify(pk, msgs[::-1], sig).tolist() == [0, 0, 0, 0]
    pk0 = R.proj_scalar_mul(R.F2, R.proj_from_affine(R.F2, R.G2_GEN_AFF), sk[0])
    assert R.verify(pk0, msgs[0], R.sign(sk[0], msgs[0]))


def test_eip197_pairing_vector(kats, coracle):
    """examples/reth_bn128.rs:389-407: two pairs multiplying to one."""
    v = {e["line"]: bytes.fromhex(e["hex"]) for e in kats["eip_vectors_raw"]["hex_literals"]}
    inp, expected = v[389], v[406]
    assert expected == (1).to_bytes(32, "big")
    g1s, g2s = [], []
    for i in range(0, len(inp), 192):
        c = inp[i:i + 192]
        ax, ay = int.from_bytes(c[0:32], "big"), int.from_bytes(c[32:64], "big")
        bay, bax, bby, bbx = [int.from_bytes(c[64 + 32 * j:96 + 32 * j], "big") for j in range(4)]
        assert R.g1_is_on_curve_affine(ax, ay)
        g1s += [ax, ay, 1]
        g2s += [bax, bay, bbx, bby, 1, 0]
    assert coracle.g2_projective_new(pack(g2s, 24)).tolist() == [0, 0]
    assert ints(coracle.glued_pairing(pack(g1s, 12), pack(g2s, 24), [0, 2])) == R.fp12_flatten(R.FP12_ONE)
    # ecAdd / ecMul vectors (reth_bn128.rs:230-243, :312-324) through the group law
    a = v[230]
    p1 = (int.from_bytes(a[0:32], "big"), int.from_bytes(a[32:64], "big"), 1)
    p2 = (int.from_bytes(a[64:96], "big"), int.from_bytes(a[96:128], "big"), 1)
    s = R.affine_from_proj(R.F1, R.proj_add(R.F1, p1, p2))
    assert R.g1_to_be_bytes(s) == v[238]
    m = v[312]
    pm = (int.from_bytes(m[0:32], "big"), int.from_bytes(m[32:64], "big"), 1)
    k = int.from_bytes(m[64:96], "big")
    assert R.g1_to_be_bytes(R.affine_from_proj(R.F1, R.proj_scalar_mul(R.F1, pm, k % P))) == v[319]
    assert R.g1_to_be_bytes_scrubbed(R.affine_zero(R.F1)) == v[257]                       # zero-sum case encodes as zeros


def test_byte_formats_roundtrip():
    rng = Xoshiro(SEED + 12)
    p = R.affine_from_proj(R.F1, R.proj_scalar_mul(R.F1, (1, 2, 1), rng.fp()))
    assert R.affine_from_proj(R.F1, R.g1_from_be_bytes(R.g1_to_be_bytes(p))) == p
    assert R.g1_from_be_bytes(R.g1_to_be_bytes(R.affine_zero(R.F1))) == R.proj_zero(R.F1)
    assert R.g1_from_be_bytes(b"\x11" * 64) is None                                        # off-curve (reth_bn128.rs:294-307)
    q = R.affine_from_proj(R.F2, R.proj_scalar_mul(R.F2, R.proj_from_affine(R.F2, R.G2_GEN_AFF), rng.fp()))
    assert R.affine_from_proj(R.F2, R.g2_from_be_bytes(R.g2_to_be_bytes(q))) == q


def test_generator_line_constants_nonzero(coracle):
    """The fused verify kernel lets a dead pair (a point at infinity) step the G2 generator with ZERO G1 coordinates: its lines reduce to
    their constant coefficient ell_0, an Fp2 value the final exponentiation kills -- provided it is never zero.  The schedule is fixed,
    so the 87 constants of the generator are checked once, here (pairing.rs:676-708 via both restatements)."""
    co = R.g2_precompute(R.G2_GEN_AFF)
    assert len(co) == 87 and all(ell[0] != (0, 0) for ell in co)
    flat = ints(coracle.g2_precompute(pack(list(R.G2_GEN_AFF[0]) + list(R.G2_GEN_AFF[1]), 16)))
    assert flat == [x for ell3 in co for e in ell3 for x in e]
