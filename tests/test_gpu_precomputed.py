"""GPU parity: the consumers of a cached `G2PreComputed` (pairing.rs:556): G2PreComputed::miller_loop(&G1Affine)
(pairing.rs:590-619), glued_miller_loop(&[G2PreComputed], &[G1Affine]) (pairing.rs:970-1022), and the same-signer verifier
against a key table kept across calls (examples/verify_multiple_messages_same_signer.rs:41-60).  Raw Miller values are
compared bit for bit (they depend on the exact line formulas, which the tables carry)."""
import numpy as np
import pytest

from helpers import SEED, Xoshiro, limbs, pack
from oracle import pyref as R

pytestmark = pytest.mark.gpu
G1 = [1, 2]
G2 = list(R.G2_GEN_AFF[0]) + list(R.G2_GEN_AFF[1])


def points(engine, n, seed):
    rng = Xoshiro(seed)
    p, _ = engine.g1_scalar_mul(np.repeat(pack(G1, 8), n, 0), limbs([rng.fp() for _ in range(n)]))
    q, _ = engine.g2_scalar_mul(np.repeat(pack(G2, 16), n, 0), limbs([rng.fp() for _ in range(n)]))
    return p, q


def test_miller_loop_precomputed_vs_oracle_and_direct(engine, coracle):
    n = 37                                                      # ragged: not a multiple of the 32 pairs per wavefront
    p, q = points(engine, n, SEED + 80)
    coeffs = engine.g2_precompute(q)
    assert np.array_equal(coeffs[:3], coracle.g2_precompute(q[:3]).reshape(3, -1))
    f = engine.miller_loop_precomputed(coeffs, p)
    assert np.array_equal(f, coracle.miller_loop(p, q))         # the oracle's miller_loop IS precompute + G2PreComputed::miller_loop
    assert np.array_equal(f, engine.miller_loop(p, q))
    # one cached table serving many G1 points (table_idx), incl. a permutation
    idx = np.array([(7 * i + 3) % 5 for i in range(n)], dtype=np.uint64)
    f2 = engine.miller_loop_precomputed(coeffs[:5], p, table_idx=idx)
    assert np.array_equal(f2, coracle.miller_loop(p, q[idx.astype(int)]))
    # and the pairing through it: final_exponentiation(table loop) == pairing()
    assert np.array_equal(engine.final_exp(f), engine.pairing(p, q))


def test_glued_miller_loop_precomputed(engine, coracle):
    ks = [1, 2, 3, 4, 5, 0, 7, 2, 1, 9, 0, 3]                  # mixed job sizes in one wavefront, empty jobs
    n = sum(ks)
    off = np.concatenate([[0], np.cumsum(ks)]).astype(np.uint64)
    p, q = points(engine, n, SEED + 81)
    coeffs = engine.g2_precompute(q)
    f = engine.glued_miller_loop_precomputed(coeffs, p, off)
    assert np.array_equal(f, engine.glued_miller_loop(p, q, off))            # the on-the-fly loop, itself pinned to the oracle
    one = np.zeros(48, dtype=np.uint64); one[0] = 1
    assert np.array_equal(f[5], one) and np.array_equal(f[10], one)          # empty product (pairing.rs:1218-1219)
    # value check against the oracle's glued_pairing through the final exponentiation
    from test_gpu_multi_pairing import proj1, proj2
    keep = [j for j, k in enumerate(ks) if k]
    exp = coracle.glued_pairing(proj1(p), proj2(q), off)
    assert np.array_equal(engine.final_exp(f)[keep], exp[keep])
    # shared table through table_idx: every pair of every job against ONE key
    tidx = np.zeros(n, dtype=np.uint64)
    f1 = engine.glued_miller_loop_precomputed(coeffs[:1], p, off, table_idx=tidx)
    assert np.array_equal(f1, engine.glued_miller_loop(p, np.repeat(q[:1], n, 0), off))


def test_same_signer_with_cached_line_table(engine):
    rng = Xoshiro(SEED + 82)
    n = 70
    sk = limbs([rng.fp()])
    msgs = [bytes([i]) * (1 + i % 9) for i in range(n)]
    sig, _ = engine.bls_sign(np.repeat(sk, n, 0), msgs)
    pk, _ = engine.g2_scalar_mul(pack(G2, 16), sk)
    bad = sig.copy()
    bad[[3, 40]] = sig[[4, 41]]
    table = engine.g2_line_table(pk)                              # built ONCE ...
    want = engine.bls_verify_same_signer(pk, msgs, bad)
    assert want.tolist() == [0 if i in (3, 40) else 1 for i in range(n)]
    for _ in range(3):                                            # ... reused across calls
        assert np.array_equal(engine.bls_verify_line_table(table, msgs, bad), want)
    assert engine.bls_verify_line_table(table, msgs[:5], sig[:5]).tolist() == [1] * 5
    # identity key: pairing() semantics, the pair contributes 1 -> only e(sig, G2gen) == 1 could pass
    assert engine.bls_verify_line_table(table, msgs[:5], sig[:5], pk_inf=[1]).tolist() == [0] * 5


def test_line_table_is_the_precompute_divided_by_its_first_coefficient(engine, coracle):
    """The device-internal key table (sylow_hip_g2_line_table) against the oracle's G2Affine::precompute (pairing.rs:676-708):
    every line (ell_0, ell_vw, ell_vv) is stored as (ell_vw / ell_0, ell_vv / ell_0) in 9 x 29-bit digits of v * 2^261 mod p,
    followed by one unit word per line (1: the first coefficient is the field's one)."""
    from helpers import P
    rinv = pow(pow(2, 261, P), P - 2, P)
    rng = Xoshiro(SEED + 83)
    for trial in range(2):
        pk, _ = engine.g2_scalar_mul(pack(G2, 16), limbs([rng.fp()]))
        words = engine.g2_line_table(pk).download().astype(np.int64)
        assert words.shape == (87 * 37,)
        assert words[87 * 36:].tolist() == [1] * 87
        got = words[:87 * 36].reshape(87, 2, 2, 9)
        ells = coracle.from_limbs(coracle.g2_precompute(pk)[0])             # 87 x (ell_0, ell_vw, ell_vv) x (c0, c1)
        for line in range(87):
            e = ells[6 * line: 6 * line + 6]
            inv = R.fp2_inv((e[0], e[1]))
            for k in range(2):
                want = R.fp2_mul((e[2 + 2 * k], e[3 + 2 * k]), inv)
                for c in range(2):
                    digits = [int(d) for d in got[line, k, c]]
                    assert all(0 <= d < (1 << 29) for d in digits[:8])
                    v = sum(d << (29 * j) for j, d in enumerate(digits))
                    assert abs(v) < 0.51 * P
                    assert v * rinv % P == want[c], (trial, line, k, c)
