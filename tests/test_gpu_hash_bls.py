"""GPU parity: Keccak-256 XMD -> hash_to_field -> SvdW -> hash_to_curve, BLS sign and verify vs the oracle.
(The reference pins no literal for this chain -- SURVEY.md §8c3; the oracle is pinned by public Keccak
KATs, RFC 9380 SHA-256 vectors through the same XMD routine and round trips: tests/test_oracle_kats.py.)"""
import numpy as np
import pytest

from helpers import P, SEED, Xoshiro, ints, limbs, pack
from oracle import pyref as R

pytestmark = pytest.mark.gpu
G2 = list(R.G2_GEN_AFF[0]) + list(R.G2_GEN_AFF[1])


def messages():
    rng = np.random.default_rng(5)
    lens = [0, 1, 2, 4, 31, 32, 33, 93, 94, 95, 96, 97, 135, 136, 137, 200, 271, 272, 273, 500]
    return [rng.integers(0, 256, size=l, dtype=np.uint8).tobytes() for l in lens] + [(20).to_bytes(4, "big")]


def test_hash_to_g1_vs_oracle(engine, coracle):
    msgs = messages()
    xy, inf = engine.hash_to_g1(msgs)
    exp_xy, exp_inf = coracle.g1_to_affine(coracle.hash_to_curve(msgs))
    assert np.array_equal(xy, exp_xy) and np.array_equal(inf, exp_inf)
    for i in (0, 5, 20):                                        # and against the independent Python restatement
        a = R.affine_from_proj(R.F1, R.hash_to_curve(msgs[i]))
        assert ints(xy[i:i + 1]) == [a[0], a[1]]
        assert R.g1_is_on_curve_affine(a[0], a[1])
    # custom and oversize (> 255 bytes -> hashed) domain separation tags (hasher.rs:157-173)
    for dst in (b"QUUX-V01-CS02-with-BN254G1_XMD:KECCAK-256_SVDW_RO_", b"x" * 255, b"y" * 256, b"z" * 400):
        xy2, _ = engine.hash_to_g1(msgs[:6], dst)
        e2, _ = coracle.g1_to_affine(coracle.hash_to_curve(msgs[:6], dst))
        assert np.array_equal(xy2, e2), len(dst)


def test_sign_verify_vs_oracle(engine, coracle):
    rng = Xoshiro(SEED + 30)
    msgs = messages()
    n = len(msgs)
    sk = limbs([rng.fp() for _ in range(n)])
    sig_xy, sig_inf = engine.bls_sign(sk, msgs)
    exp_xy, exp_inf = coracle.g1_to_affine(coracle.sign(sk, msgs))
    assert np.array_equal(sig_xy, exp_xy) and np.array_equal(sig_inf, exp_inf)
    pk_xy, pk_inf = engine.g2_scalar_mul(np.repeat(pack(G2, 16), n, 0), sk)
    bad_sig, _ = engine.g1_add(sig_xy, np.repeat(pack([1, 2], 8), n, 0))
    plant = np.zeros(n, dtype=bool)
    plant[[1, 7, 13]] = True
    mixed = np.where(plant[:, None], bad_sig, sig_xy)
    pk_proj = np.concatenate([pk_xy, np.repeat(pack([1, 0], 8), n, 0)], axis=1)
    sig_proj = np.concatenate([mixed, np.repeat(limbs([1]), n, 0)], axis=1)
    inf = np.ones(n, dtype=np.uint8)
    for two in (False, True):        # the default entry point (one final exponentiation) and the literal two-pairing form
        verify = lambda *a, **k: engine.bls_verify(*a, two_pairings=two, **k)
        assert verify(pk_xy, msgs, sig_xy).tolist() == [1] * n                         # lib.rs:29-42 round trip
        # planted corruption pattern: wrong message, sig + G1gen, wrong key
        got = verify(pk_xy, msgs, mixed)
        assert got.tolist() == (~plant).astype(int).tolist()
        assert np.array_equal(got, coracle.verify(pk_proj[:8], msgs[:8], sig_proj[:8]).tolist() + got[8:].tolist())
        assert verify(pk_xy, msgs[::-1], sig_xy).sum() <= 1                            # only a palindromic position could match
        assert verify(np.roll(pk_xy, 1, axis=0), msgs, sig_xy).tolist() == [0] * n
        # infinity signature / key: pairing() maps them to the identity (pairing.rs:876-886) -> both sides must be 1 to pass
        assert verify(pk_xy, msgs, sig_xy, sig_inf=inf).tolist() == [0] * n
        assert verify(pk_xy, msgs, sig_xy, pk_inf=inf, sig_inf=inf).tolist() == [1] * n


def test_single_verification_equals_the_batch_kernel(engine):
    """A batch of ONE verification takes the latency route (a one-element aggregate on whole wavefronts): element by element it must give the
    flag the batch kernel gives for the same tuple -- valid, corrupted, every combination of identity flags, a key outside the r-torsion."""
    rng = Xoshiro(SEED + 31)
    msgs = messages()[:8]
    n = len(msgs)
    sk = limbs([rng.fp() for _ in range(n)])
    sig, _ = engine.bls_sign(sk, msgs)
    pk, _ = engine.g2_scalar_mul(np.repeat(pack(G2, 16), n, 0), sk)
    bad, _ = engine.g1_add(sig, np.repeat(pack([1, 2], 8), n, 0))
    mixed = sig.copy(); mixed[[1, 5]] = bad[[1, 5]]
    pk2 = pk.copy(); pk2[3] = pk[4]                                       # wrong key
    pinf = np.array([0, 0, 1, 0, 0, 1, 0, 1], np.uint8); sinf = np.array([0, 0, 0, 1, 0, 1, 0, 0], np.uint8)
    for keys, sigs, kw in ((pk, sig, {}), (pk, mixed, {}), (pk2, sig, {}), (pk, mixed, dict(pk_inf=pinf, sig_inf=sinf))):
        batch = engine.bls_verify(keys, msgs, sigs, **kw)
        for i in range(n):
            one_kw = {k: v[i:i + 1] for k, v in kw.items()}
            assert engine.bls_verify(keys[i:i + 1], msgs[i:i + 1], sigs[i:i + 1], **one_kw)[0] == batch[i], (i, kw.keys())


def test_verify_batch_planted_pattern_large(engine):
    """2^12 tuples, 1/64 corrupted at PRNG-chosen indices: flags must equal the planted pattern, and
    the device-side AND (flags_all) must see it."""
    n = 1 << 12
    g = np.random.default_rng(77)
    msgs = [g.integers(0, 256, size=32, dtype=np.uint8).tobytes() for _ in range(n)]
    sk = g.integers(0, 1 << 62, size=(n, 4), dtype=np.uint64)
    sk[:, 3] &= np.uint64((1 << 60) - 1)
    sig_xy, _ = engine.bls_sign(sk, msgs)
    pk_xy, _ = engine.g2_scalar_mul(np.repeat(pack(G2, 16), n, 0), sk)
    plant = g.random(n) < 1 / 64
    bad_sig, _ = engine.g1_add(sig_xy, np.repeat(pack([1, 2], 8), n, 0))
    mixed = np.where(plant[:, None], bad_sig, sig_xy)
    ok = engine.bls_verify(pk_xy, msgs, mixed)
    assert np.array_equal(ok.astype(bool), ~plant)
    d_ok = engine.to_device(ok)
    assert engine.flags_all(d_ok) == 0
    assert engine.flags_all(engine.to_device(engine.bls_verify(pk_xy, msgs, sig_xy))) == 1


@pytest.mark.parametrize("n", [301, 4097])
def test_same_signer_on_both_routes(engine, n):
    """One key for every message at a size where a wavefront holds two Miller loops and all of them read the ONE key (301), and above the
    one-wavefront route's cap (4097: the line-table kernel): the flags of the per-element verifier with the key repeated."""
    g = np.random.default_rng(4321 + n)
    msgs = [g.integers(0, 256, size=int(g.integers(1, 50)), dtype=np.uint8).tobytes() for _ in range(n)]
    sk = g.integers(0, 1 << 62, size=(1, 4), dtype=np.uint64)
    pk, _ = engine.g2_scalar_mul(pack(G2, 16), sk)
    sig, _ = engine.bls_sign(np.repeat(sk, n, 0), msgs)
    bad, _ = engine.g1_add(sig, np.repeat(pack([1, 2], 8), n, 0))
    plant = g.random(n) < 0.1
    plant[[0, n - 1]] = [True, False]
    mixed = np.where(plant[:, None], bad, sig)
    sinf = (g.random(n) < 0.05).astype(np.uint8)
    got = engine.bls_verify_same_signer(pk, msgs, mixed, sig_inf=sinf)
    assert np.array_equal(got.astype(bool), ~plant & ~sinf.astype(bool))
    assert np.array_equal(got, engine.bls_verify(np.repeat(pk, n, 0), msgs, mixed, sig_inf=sinf, pipelined=False))
    assert np.array_equal(engine.bls_verify_same_signer(pk, msgs, mixed, pk_inf=[1], sig_inf=sinf).astype(bool), sinf.astype(bool))


@pytest.mark.parametrize("n", [513, 1500, 2047, 3333])
def test_verify_two_elements_per_wavefront_route(engine, coracle, n):
    """128 < n <= 4096 verifications: 2 n Miller loops and n final exponentiations, two per wavefront.  Planted corruption and identity
    flags in both halves; flags equal the planted pattern, the literal two-pairing form, and (a sample) the oracle."""
    g = np.random.default_rng(1234 + n)
    msgs = [g.integers(0, 256, size=int(g.integers(0, 70)), dtype=np.uint8).tobytes() for _ in range(n)]
    sk = g.integers(0, 1 << 62, size=(n, 4), dtype=np.uint64)
    sk[:, 3] &= np.uint64((1 << 60) - 1)
    sig_xy, _ = engine.bls_sign(sk, msgs)
    pk_xy, _ = engine.g2_scalar_mul(np.repeat(pack(G2, 16), n, 0), sk)
    plant = g.random(n) < 1 / 16
    plant[[0, 1, n - 1]] = [True, False, True]
    bad_sig, _ = engine.g1_add(sig_xy, np.repeat(pack([1, 2], 8), n, 0))
    mixed = np.where(plant[:, None], bad_sig, sig_xy)
    pinf, sinf = np.zeros(n, np.uint8), np.zeros(n, np.uint8)
    pinf[[7, 40, n - 2]] = 1
    sinf[[8, 40, n - 3]] = 1
    expect = (~plant).astype(np.uint8)
    expect[[7, 8, n - 2, n - 3]] = 0                                                  # one side the identity, the other not
    expect[40] = 1                                                                     # both sides the identity
    got = engine.bls_verify(pk_xy, msgs, mixed, pk_inf=pinf, sig_inf=sinf, pipelined=False)
    assert np.array_equal(got, expect)
    assert np.array_equal(got, engine.bls_verify(pk_xy, msgs, mixed, pk_inf=pinf, sig_inf=sinf))
    assert np.array_equal(got, engine.bls_verify(pk_xy, msgs, mixed, pk_inf=pinf, sig_inf=sinf, two_pairings=True))
    idx = [0, 1, 2, 31, 32, 33, n - 1]
    pk_proj = np.concatenate([pk_xy[idx], np.repeat(pack([1, 0], 8), len(idx), 0)], axis=1)
    sig_proj = np.concatenate([mixed[idx], np.repeat(limbs([1]), len(idx), 0)], axis=1)
    assert got[idx].tolist() == list(coracle.verify(pk_proj, [msgs[i] for i in idx], sig_proj))


def test_one_final_exponentiation_verify_equals_the_two_pairing_form(engine, coracle):
    """sylow_hip_bls_verify_batch (= _fused_: e(sig, G2gen) * e(-H, pk) == 1, one final exponentiation) gives the booleans of
    lib.rs:223-236 evaluated literally (sylow_hip_bls_verify_two_pairings_batch): planted corruption, identity inputs, and keys
    OUTSIDE the r-torsion (conj(miller(H, pk)) = miller(-H, pk) holds line by line, for any twist point)."""
    from test_gpu_groups import fp2_sqrt
    rng = Xoshiro(SEED + 31)
    msgs = messages() * 7                                    # 147 elements: spans three wavefronts with a ragged tail
    n = len(msgs)
    sk = limbs([rng.fp() for _ in range(n)])
    sig_xy, _ = engine.bls_sign(sk, msgs)
    pk_xy, _ = engine.g2_scalar_mul(np.repeat(pack(G2, 16), n, 0), sk)
    for fused in (False, True):
        assert engine.bls_verify(pk_xy, msgs, sig_xy, fused=fused).tolist() == [1] * n
    bad_sig, _ = engine.g1_add(sig_xy, np.repeat(pack([1, 2], 8), n, 0))
    plant = np.random.default_rng(9).random(n) < 0.2
    mixed = np.where(plant[:, None], bad_sig, sig_xy)
    got = engine.bls_verify(pk_xy, msgs, mixed)
    assert np.array_equal(got.astype(bool), ~plant)
    assert np.array_equal(got, engine.bls_verify(pk_xy, msgs, mixed, two_pairings=True))
    assert np.array_equal(got, engine.bls_verify(pk_xy, msgs, mixed, fused=True))
    assert engine.bls_verify(np.roll(pk_xy, 1, axis=0), msgs, sig_xy).sum() == 0
    # identity handling equals pairing()'s: random flags on both sides
    g = np.random.default_rng(10)
    pinf, sinf = (g.random(n) < 0.3).astype(np.uint8), (g.random(n) < 0.3).astype(np.uint8)
    a = engine.bls_verify(pk_xy, msgs, mixed, pk_inf=pinf, sig_inf=sinf)
    b = engine.bls_verify(pk_xy, msgs, mixed, pk_inf=pinf, sig_inf=sinf, two_pairings=True)
    assert np.array_equal(a, b)
    assert a[(pinf & sinf).astype(bool)].all()               # both sides identity -> 1 == 1
    # keys that are twist points outside G2, and signatures that are arbitrary curve points: same answers from both forms
    wild = []
    while len(wild) < 12:
        x = (rng.fp(), rng.fp())
        y = fp2_sqrt(R.fp2_add(R.fp2_mul(R.fp2_square(x), x), R.TWIST_B))
        if y is not None:
            wild.append(list(x) + list(y))
    wild = pack([v for q in wild for v in q], 16)
    assert engine.g2_subgroup_check(wild).tolist() == [2] * 12
    a = engine.bls_verify(wild, msgs[:12], sig_xy[:12])
    b = engine.bls_verify(wild, msgs[:12], sig_xy[:12], two_pairings=True)
    assert np.array_equal(a, b) and not a.any()
    wild_proj = np.concatenate([wild, np.repeat(pack([1, 0], 8), 12, 0)], axis=1)
    sig_proj = np.concatenate([sig_xy[:12], np.repeat(limbs([1]), 12, 0)], axis=1)
    assert np.array_equal(a[:4], coracle.verify(wild_proj[:4], msgs[:4], sig_proj[:4]))


def test_same_signer_and_precompute(engine, coracle):
    """G2Affine::precompute replayed coefficient by coefficient (pairing.rs:676-708), and the same-signer batch
    verify of examples/verify_multiple_messages_same_signer.rs against the per-element verifier."""
    rng = Xoshiro(SEED + 32)
    q, _ = engine.g2_scalar_mul(np.repeat(pack(G2, 16), 9, 0), limbs([rng.fp() for _ in range(9)]))
    q = np.concatenate([q, pack(G2, 16)])
    assert np.array_equal(engine.g2_precompute(q), coracle.g2_precompute(q))
    msgs = messages() * 4
    n = len(msgs)
    sk = limbs([rng.fp()])
    pk, _ = engine.g2_scalar_mul(pack(G2, 16), sk)
    sig, _ = engine.bls_sign(np.repeat(sk, n, 0), msgs)
    assert engine.bls_verify_same_signer(pk, msgs, sig).tolist() == [1] * n
    bad, _ = engine.g1_add(sig, np.repeat(pack([1, 2], 8), n, 0))
    plant = np.random.default_rng(3).random(n) < 0.25
    mixed = np.where(plant[:, None], bad, sig)
    got = engine.bls_verify_same_signer(pk, msgs, mixed)
    assert np.array_equal(got.astype(bool), ~plant)
    assert np.array_equal(got, engine.bls_verify(np.repeat(pk, n, 0), msgs, mixed))
    other, _ = engine.g2_scalar_mul(pack(G2, 16), limbs([rng.fp()]))
    assert engine.bls_verify_same_signer(other, msgs, sig).sum() == 0
    sinf = (np.random.default_rng(4).random(n) < 0.3).astype(np.uint8)
    assert np.array_equal(engine.bls_verify_same_signer(pk, msgs, mixed, sig_inf=sinf), engine.bls_verify(np.repeat(pk, n, 0), msgs, mixed, sig_inf=sinf))
    assert np.array_equal(engine.bls_verify_same_signer(pk, msgs, mixed, pk_inf=[1], sig_inf=sinf).astype(bool), sinf.astype(bool))


def test_hash_to_field_vs_oracle(engine, coracle):
    """Expander::hash_to_field (hasher.rs:84-128): Keccak-256 XMD + 48-byte reduction, ragged message lengths across the
    136-byte rate boundary, default and custom DST (incl. an oversize one, hasher.rs:157-173)."""
    from oracle import pyref as R
    msgs = [bytes((7 * i + j) & 0xFF for j in range(L)) for i, L in enumerate([0, 1, 3, 31, 32, 33, 55, 100, 135, 136, 137, 200, 271, 272, 273, 500])]
    for dst in (None, b"QUUX-V01-CS02-with-expander-SHA256-128", b"x" * 300):
        got = engine.hash_to_field(msgs, dst)
        d = dst if dst is not None else R.DST
        for i, m in enumerate(msgs):
            em = coracle.expand_message_xmd_keccak(m, d, 96)
            want = [int.from_bytes(em[:48], "big") % R.P, int.from_bytes(em[48:], "big") % R.P]
            have = [sum(int(got[i, 4 * c + k]) << (64 * k) for k in range(4)) for c in range(2)]
            assert have == want, (i, dst)


def test_svdw_map_and_compute_naf_entry_points(engine, coracle):
    """a30 / a6 of the scope table as entry points of their own: SvdW::unchecked_map_to_point (svdw.rs:180-262) against both oracles
    on edge and random field elements, and Fp::compute_naf (fp.rs:653-662) on raw 256-bit values incl. the wrap at 2^256."""
    rng = Xoshiro(SEED + 33)
    # +-1/2 are the u with tv1 tv2 = 0 (the inv0 case of svdw.rs:196); the device maps elements in neighbouring pairs that share one
    # inversion: (P - 2, -1/2), (1/2, 3) exercise a zero on either side, the last two a zero on both
    us = [0, 1, 2, P - 1, P - 2, (P - 1) // 2, (P + 1) // 2, 3, 4] + [rng.fp() for _ in range(120)] + [(P - 1) // 2, (P + 1) // 2, (P - 1) // 2]
    xy, st = engine.svdw_map(limbs(us))
    assert not st.any()
    assert np.array_equal(xy, coracle.svdw_map(limbs(us)))
    got = ints(xy)
    for i in (0, 1, 3, 9, 50):
        x, y = R.svdw_map_to_point(us[i])
        assert got[2 * i: 2 * i + 2] == [x, y] and R.g1_is_on_curve_affine(x, y)
    ks = [0, 1, 2, 3, 5, 7, (1 << 256) - 1, 1 << 255, (1 << 255) - 1, R.R_ORDER, P, int("a" * 64, 16), int("5" * 64, 16), int("6" * 64, 16)] + [rng.u256() for _ in range(86)]
    kw = np.array([[(k >> (64 * j)) & ((1 << 64) - 1) for j in range(4)] for k in ks], dtype=np.uint64)
    np_, nm_ = engine.fp_compute_naf(kw)
    for i, k in enumerate(ks):
        ep, em = R.compute_naf(k)
        assert sum(int(np_[i, j]) << (64 * j) for j in range(4)) == ep and sum(int(nm_[i, j]) << (64 * j) for j in range(4)) == em, hex(k)
        assert ep & em == 0 and (ep | em) & ((ep | em) << 1) & ((1 << 256) - 1) == 0 or k >= (1 << 255)      # non-adjacent form

def test_large_batch_verify_kernels_agree(engine):
    """Above the wide-route threshold bls_verify_batch is k_bls_verify_fused, whose key side runs on the isomorphic curves,
    and the literal two-pairing form is k_bls_verify on the reference's curves: same booleans for valid and planted-invalid signatures,
    identity flags on either side, and keys that are twist points OUTSIDE the r-torsion (phi is an isomorphism of the whole curve)."""
    from test_gpu_groups import fp2_sqrt
    rng = Xoshiro(SEED + 34)
    n = 1500
    sk = engine.xoshiro_fp_soa(SEED + 35, n).T.copy()
    msgs = [bytes([(7 * i) & 255, i >> 8]) * (1 + i % 23) for i in range(n)]
    sig_xy, _ = engine.bls_sign(sk, msgs)
    pk_xy, _ = engine.g2_scalar_mul(np.repeat(pack(G2, 16), n, 0), sk)
    wild = []
    while len(wild) < 40:
        x = (rng.fp(), rng.fp())
        y = fp2_sqrt(R.fp2_add(R.fp2_mul(R.fp2_square(x), x), R.TWIST_B))
        if y is not None:
            wild.append(list(x) + list(y))
    pk_xy[100:140] = pack([v for q in wild for v in q], 16)
    g = np.random.default_rng(12)
    plant = g.random(n) < 0.15
    mixed = np.where(plant[:, None], np.roll(sig_xy, 3, axis=0), sig_xy)
    pinf, sinf = (g.random(n) < 0.05).astype(np.uint8), (g.random(n) < 0.05).astype(np.uint8)
    for flags in ((None, None), (pinf, sinf)):
        a = engine.bls_verify(pk_xy, msgs, mixed, pk_inf=flags[0], sig_inf=flags[1], pipelined=False)
        b = engine.bls_verify(pk_xy, msgs, mixed, pk_inf=flags[0], sig_inf=flags[1], two_pairings=True, pipelined=False)
        assert np.array_equal(a, b)
    a = engine.bls_verify(pk_xy, msgs, mixed, pipelined=False).astype(bool)
    expect = ~plant
    expect[100:140] = False
    assert np.array_equal(a, expect)


def test_sign_edge_scalars_and_ragged_groups(engine, coracle):
    """bls_sign_batch (lib.rs:179-187) on the batch sizes around the eight-lanes-per-signature groups of sign_wide.hip (1, 7, 8, 9, 63 ... 67:
    ragged last wavefront, ragged last group) with planted edge scalars -- 0, 1, r - 1, r, r + 1 (the scalar is an Fp value and acts mod r:
    r gives the identity signature), p - 1, a value >= p (reduced like Fp::new), 2^128 +- 1 (GLV halves of extreme size) -- and message
    lengths on both sides of the Keccak rate (0, 1, 135, 136, 137, 300 bytes).  Every row against the oracle; under SYLOW_HIP_WIDE_TAIL=0
    (tests/test_gpu_routes.py) the same rows go through the one-lane kernel."""
    from helpers import P as PMOD
    R_ORDER = 0x30644E72E131A029B85045B68181585D2833E84879B9709143E1F593F0000001
    edge = [0, 1, R_ORDER - 1, R_ORDER, R_ORDER + 1, PMOD - 1, PMOD + 5, (1 << 256) - 1, (1 << 128) - 1, (1 << 128) + 1, 2]
    lens = [0, 1, 135, 136, 137, 300, 32, 4]
    rng = Xoshiro(SEED + 41)
    for n in (1, 7, 8, 9, 63, 64, 65, 67):
        sk_int = [edge[i % len(edge)] if i % 3 == 0 else rng.fp() for i in range(n)]
        msgs = [bytes((7 * i + j) & 255 for j in range(lens[(i + n) % len(lens)])) for i in range(n)]
        sk = limbs(sk_int)
        sig_xy, sig_inf = engine.bls_sign(sk, msgs)
        exp_xy, exp_inf = coracle.g1_to_affine(coracle.sign(limbs([k % PMOD for k in sk_int]), msgs))     # the engine reduces like Fp::new (fp.rs:199-201)
        assert np.array_equal(sig_xy, exp_xy) and np.array_equal(sig_inf, exp_inf), n
        zero_rows = [i for i, k in enumerate(sk_int) if k % PMOD % R_ORDER == 0]
        assert all(sig_inf[i] == 1 for i in zero_rows)
