"""The reference's own pairing / signature tests (src/pairing.rs:1038-1251, src/lib.rs:29-42), replayed
on batches through the host-side mirror of sylow's API (sylow_amd/api.py)."""
import numpy as np
import pytest

from helpers import SEED, Xoshiro

pytestmark = pytest.mark.gpu
MSG = (20).to_bytes(4, "big")          # `20_i32.to_be_bytes()`, pairing.rs:1046 / benches/sig.rs:7


@pytest.fixture(scope="module")
def api(engine):
    from sylow_amd import api
    api.set_engine(engine)
    return api


def test_gt_generator(api, kats):
    """pairing.rs:1052-1057"""
    gt = api.pairing(api.G1Projective.generator(), api.G2Projective.generator())
    assert [sum(int(gt.v[0, 4 * i + k]) << (64 * k) for k in range(4)) for i in range(12)] == [int(x, 16) for x in kats["gt_generator"]["value"]]


def test_signatures(api):
    """pairing.rs:1059-1072 on a batch of 64 keys"""
    rng = Xoshiro(SEED + 50)
    n = 64
    private_key = api.fp([rng.fp() for _ in range(n)])
    hashed_message = api.G1Projective.hash_to_curve([MSG] * n)
    signature = hashed_message * private_key
    public_key = api.G2Projective.generator(n) * private_key
    lhs = api.pairing(signature, api.G2Projective.generator(n))
    rhs = api.pairing(hashed_message, public_key)
    assert (lhs == rhs).all()
    # lib.rs:29-42 doc-test: sign / verify
    sig = api.sign(private_key, [MSG] * n)
    assert (sig == signature).all()
    assert api.verify(public_key, [MSG] * n, sig).all()
    assert not api.verify(public_key, [b"Hello, World!"] * n, sig).any()


def test_identities(api):
    """pairing.rs:1101-1120"""
    g1z, g2 = api.G1Projective.zero(), api.G2Projective.generator()
    assert (api.pairing(g1z, g2) == api.Gt.identity()).all()
    assert (api.pairing(api.G1Projective.generator(), api.G2Projective.zero()) == api.Gt.identity()).all()
    g, h = api.G1Projective.generator(), api.G2Projective.generator()
    q = api.pairing(g, -h)
    r = api.pairing(-g, h)
    assert (q == r).all()
    assert ((q + api.pairing(g, h)) == api.Gt.identity()).all()      # p = -pairing(g, h)


def test_batches(api):
    """pairing.rs:1216-1242: glued_pairing(&[], &[]) == identity; e(sP, Q) products == e(P, sQ) products"""
    rng = Xoshiro(SEED + 51)
    empty = api.glued_pairing(api.G1Projective(np.zeros((0, 8), np.uint64)), api.G2Projective(np.zeros((0, 16), np.uint64)))
    assert (empty == api.Gt.identity()).all()
    RANGE = 50
    p = api.G1Projective.generator(RANGE) * api.fp([rng.fp() for _ in range(RANGE)])
    q = api.G2Projective.generator(RANGE) * api.fp([rng.fp() for _ in range(RANGE)])
    s = api.fp([rng.fp() for _ in range(RANGE)])
    b_batch = api.glued_pairing(p * s, q)
    c_batch = api.glued_pairing(p, q * s)
    assert (b_batch == c_batch).all() and not (b_batch == api.Gt.identity()).any()


def test_group_errors(api):
    """G2Projective::new error variants (g2.rs:460-525; groups/mod.rs invalid-subgroup fixtures)"""
    good = api.G2Projective.generator(2)
    assert len(api.G2Affine.new(good.xy)) == 2
    bad = good.xy.copy()
    bad[0, 8] ^= np.uint64(1)
    with pytest.raises(api.GroupError, match="NotOnCurve"):
        api.G2Affine.new(bad)


def test_precompute_and_miller_types(api, coracle):
    """G2Affine::precompute / G2PreComputed::miller_loop / MillerLoopResult::final_exponentiation (pairing.rs:556-619)"""
    rng = Xoshiro(SEED + 52)
    n = 16
    p = api.G1Projective.generator(n) * api.fp([rng.fp() for _ in range(n)])
    q = api.G2Projective.generator(n) * api.fp([rng.fp() for _ in range(n)])
    pre = q.precompute()
    assert np.array_equal(pre.coeffs, coracle.g2_precompute(q.xy))
    ml = pre.miller_loop(p)
    assert np.array_equal(ml.v, coracle.miller_loop(p.xy, q.xy))
    assert (ml.final_exponentiation() == api.pairing(p, q)).all()
    # MillerLoopResult default * x == x (pairing.rs:1244-1250)
    one = np.zeros((n, 48), dtype=np.uint64); one[:, 0] = 1
    assert ((api.MillerLoopResult(one) * ml) == ml).all()


def test_shared_secret(api):
    """pairing.rs:1074-1099: three-party Diffie-Hellman through Gt * Fr, 32 triples at once"""
    from oracle import pyref as R
    rng = Xoshiro(SEED + 55)
    n = 32
    sk = [api.fp([rng.fp() % R.R_ORDER for _ in range(n)]) for _ in range(3)]          # alice, bob, carol
    pk1 = [api.G1Projective.generator(n) * k for k in sk]
    pk2 = [api.G2Projective.generator(n) * k for k in sk]
    alice_ss = api.pairing(pk1[1], pk2[2]) * sk[0]
    bob_ss = api.pairing(pk1[2], pk2[0]) * sk[1]
    carol_ss = api.pairing(pk1[0], pk2[1]) * sk[2]
    assert (alice_ss == bob_ss).all() and (bob_ss == carol_ss).all()
    assert not (alice_ss == api.Gt.identity(n)).any()


def test_bilinearity(api):
    """pairing.rs:1192-1213 on 40 random (p, q, s) triples"""
    from oracle import pyref as R
    rng = Xoshiro(SEED + 56)
    n = 40
    fr = lambda: api.fp([rng.fp() % R.R_ORDER for _ in range(n)])
    p, q, s = api.G1Projective.generator(n) * fr(), api.G2Projective.generator(n) * fr(), fr()
    a = api.pairing(p, q) * s
    assert (a == api.pairing(p * s, q)).all() and (a == api.pairing(p, q * s)).all()
    t = api.fp([R.R_ORDER - 1] * n)                                                       # -Fr::ONE
    assert not (a == api.Gt.identity(n)).any()
    assert (((a * t) + a) == api.Gt.identity(n)).all()


def test_rand_endomorphism_and_wire_formats(api):
    """GroupTrait::{rand, endomorphism} and the wire formats through the API mirror: g2.rs:140-152 (psi(Q) = [p]Q on the
    r-torsion), g1.rs:151-280 / g2.rs:319-433 round trips, fp.rs:686-778 rejections (the shapes of groups/mod.rs's
    `test_g1_serialisation` / `test_g2_serialisation` and fp.rs:821-826)."""
    from helpers import P
    from oracle import pyref as R
    n = 24
    p, q = api.G1Projective.rand(n, seed=11), api.G2Projective.rand(n, seed=12)
    assert not p.is_zero().any() and not q.is_zero().any()
    assert (api.G1Projective.rand(n, seed=11) == p).all() and not (api.G1Projective.rand(n, seed=13) == p).all()
    assert len(api.G2Affine.new(q.xy)) == n                                            # rand lands in the r-torsion (g2.rs:204-240)
    assert (q.endomorphism() == q * api.fp([P % R.R_ORDER] * n)).all()                 # psi acts as multiplication by p mod r
    assert (api.G2Affine.zero(3).endomorphism().is_zero()).all()                       # psi(identity) = identity (g2.rs:141-143)
    assert (api.G1Affine.from_be_bytes(p.to_be_bytes()) == p).all()
    assert (api.G2Affine.from_be_bytes(q.to_be_bytes()) == q).all()
    blob = bytearray(p.to_be_bytes()[0]); blob[63] ^= 1                                # y + 1: off the curve
    with pytest.raises(api.GroupError, match="NotOnCurve"):
        api.G1Affine.from_be_bytes([bytes(blob)])
    with pytest.raises(api.GroupError, match="DecodeError"):
        api.G1Affine.from_be_bytes([P.to_bytes(32, "big") + (2).to_bytes(32, "big")])   # x = p: not a canonical residue
    v, is_some = api.fp_from_be_bytes([P.to_bytes(32, "big"), (P - 1).to_bytes(32, "big")])
    assert is_some.tolist() == [False, True] and api.fp_to_be_bytes(v) == [(0).to_bytes(32, "big"), (P - 1).to_bytes(32, "big")]
    with pytest.raises(api.GroupError, match="DecodeError"):
        api.Fr.from_be_bytes([R.R_ORDER.to_bytes(32, "big")])                          # the case reth_bn128.rs:150 unwraps
    s = api.Fr.rand(n, seed=5)
    assert (api.Fr.from_be_bytes(s.to_be_bytes()) == s).all()


def test_key_table_and_cached_precompute(api):
    """examples/verify_multiple_messages_same_signer.rs:41-60 with the key's table cached across calls, and one cached
    G2PreComputed serving many G1 points (pairing.rs:556-619)."""
    rng = Xoshiro(SEED + 57)
    sk = api.fp([rng.fp()])
    pk = api.G2Projective.generator(1) * sk
    key = api.KeyTable(pk)
    for rnd in range(2):
        msgs = [bytes([rnd, i]) for i in range(9)]
        sig = api.sign(np.repeat(sk, 9, 0), msgs)
        assert key.verify(msgs, sig).all()
        assert not key.verify(msgs[::-1], sig)[:4].any()
    pre = pk.precompute()
    p = api.G1Projective.rand(7, seed=3)
    ml = pre.miller_loop(p, table_idx=np.zeros(7, dtype=np.uint64))
    assert (ml.final_exponentiation() == api.pairing(p, api.G2Affine(np.repeat(pk.xy, 7, 0)))).all()
    g = api.glued_miller_loop(api.G2Affine(np.repeat(pk.xy, 7, 0)).precompute(), p)
    assert (g.final_exponentiation() == api.glued_pairing(p, api.G2Affine(np.repeat(pk.xy, 7, 0)))).all()


def test_keypair_sign_verify_round_trip(api):
    """lib.rs:29-42 (`KeyPair::generate` -> `sign` -> `verify`) on a batch, both generators"""
    from helpers import ints
    for seed in (7, None):
        kp = api.KeyPair.generate(9, seed=seed)
        assert len(kp) == 9 and all(0 <= k < api.R_ORDER for k in ints(kp.secret_key))
        assert np.array_equal(kp.public_key.xy, (api.G2Projective.generator(9) * kp.secret_key).xy)
        msgs = [bytes([i]) * (i + 1) for i in range(9)]
        sig = api.sign(kp.secret_key, msgs)
        assert api.verify(kp.public_key, msgs, sig).all()
        assert not api.verify(kp.public_key, msgs[::-1], sig)[:4].any()
    a, b = api.KeyPair.generate(3, seed=11), api.KeyPair.generate(3, seed=11)
    assert np.array_equal(a.secret_key, b.secret_key)


def test_aggregate_verify_api(api):
    kp = api.KeyPair.generate(12, seed=3)
    msgs = [bytes([i, 7]) for i in range(12)]
    sig = api.sign(kp.secret_key, msgs)
    assert api.aggregate_verify(kp.public_key, msgs, sig) is True
    assert api.aggregate_verify(kp.public_key, msgs[::-1], sig) is False
    one = api.KeyPair.generate(1, seed=4)
    sig1 = api.sign(np.repeat(one.secret_key, 12, 0), msgs)
    assert api.aggregate_verify(one.public_key, msgs, sig1) is True
    assert api.aggregate_verify(one.public_key, msgs, sig) is False


def test_batch_verify_api(api):
    """The small-exponent batch test of the mirror: accepts a valid batch, rejects one with a swapped signature (seeded and OS-drawn
    weights), validates weight_bits, and refuses a key that was built from raw coordinates outside G2 proper."""
    kp = api.KeyPair.generate(9, seed=5)
    msgs = [bytes([i, 1, 2]) for i in range(9)]
    sig = api.sign(kp.secret_key, msgs)
    assert api.batch_verify(kp.public_key, msgs, sig) and api.batch_verify(kp.public_key, msgs, sig, weight_bits=64, seed=11)
    bad = api.G1Affine(np.roll(sig.xy, 1, axis=0), sig.infinity)
    assert not api.batch_verify(kp.public_key, msgs, bad) and not api.batch_verify(kp.public_key, msgs, bad, seed=12)
    for wb in (0, 129):
        with pytest.raises(ValueError):
            api.batch_verify(kp.public_key, msgs, sig, weight_bits=wb)
    raw = api.G2Affine(kp.public_key.xy.copy(), kp.public_key.infinity)           # membership not established: checked, and passes
    assert not raw.in_subgroup and api.batch_verify(raw, msgs, sig)
    off = raw.xy.copy()
    off[0, :4] = np.array([5, 0, 0, 0], dtype=np.uint64)                          # not on the twist
    with pytest.raises(ValueError):
        api.batch_verify(api.G2Affine(off, raw.infinity), msgs, sig)


def test_mirror_sub_and_tower_classes(api, coracle):
    """Sub for points (group.rs:614-624) and the Fp2 / Fp6 / Fp12 value classes of the mirror: operators and the small items
    (residue_mul, frobenius, square) against the oracle."""
    from helpers import rand_fp_array
    rng = Xoshiro(SEED + 310)
    p, q = api.G1Projective.rand(6, seed=21), api.G1Projective.rand(6, seed=22)
    assert ((p - q) + q == p).all() and (p - p).is_zero().all()
    a, b = api.G2Projective.rand(4, seed=23), api.G2Projective.rand(4, seed=24)
    assert ((a - b) + b == a).all() and (a - a).is_zero().all()
    x2, y2 = api.Fp2(rand_fp_array(rng, 8, 2)), api.Fp2(rand_fp_array(rng, 8, 2))
    assert np.array_equal((x2 * y2).v, coracle.fp2_op("mul", x2.v, y2.v)) and np.array_equal((x2 - y2).v, coracle.fp2_op("sub", x2.v, y2.v))
    assert np.array_equal(x2.residue_mul().v, coracle.fp2_op("mul_xi", x2.v)) and np.array_equal(x2.frobenius(3).v, coracle.fp2_frobenius(x2.v, 3))
    assert ((x2 * x2.inv()).v[:, 0] == 1).all()
    x6 = api.Fp6(rand_fp_array(rng, 8, 6))
    assert np.array_equal(x6.square().v, coracle.fp6_op("sqr", x6.v)) and np.array_equal(x6.residue_mul().v, coracle.fp6_residue_mul(x6.v))
    assert np.array_equal(x6.frobenius(5).v, coracle.fp6_frobenius(x6.v, 5)) and np.array_equal((-x6).v, coracle.fp6_op("neg", x6.v))
    x12, y12 = api.Fp12(rand_fp_array(rng, 4, 12)), api.Fp12(rand_fp_array(rng, 4, 12))
    assert np.array_equal((x12 * y12).v, coracle.fp12_op("mul", x12.v, y12.v)) and np.array_equal(x12.square().v, coracle.fp12_op("sqr", x12.v))
    assert np.array_equal(x12.frobenius(2).v, coracle.fp12_op("frobenius", x12.v, arg=2))
    assert np.array_equal(((x12 + y12) - y12).v, x12.v)
