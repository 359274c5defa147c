"""GPU parity: one boolean / one Gt from a batch that is sharded over GPUs without Python in the loop:
raw partial Miller products per shard + product + ONE final exponentiation == glued_pairing over all pairs
(pairing.rs:970-1037), and the AND of flag vectors.  The RCCL entry points are exercised through a real one-rank
communicator (this box has one GPU); the two-rank logic is covered by tests/test_gpu_bench_ranks.py and the gloo CPU tests."""
import ctypes
import os

import numpy as np
import pytest

from helpers import SEED, Xoshiro, limbs, pack
from oracle import pyref as R
from test_gpu_multi_pairing import G1, G2, proj1, proj2

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pairs(engine):
    rng = Xoshiro(SEED + 90)
    n = 301
    p, _ = engine.g1_scalar_mul(np.repeat(pack(G1, 8), n, 0), limbs([rng.fp() for _ in range(n)]))
    q, _ = engine.g2_scalar_mul(np.repeat(pack(G2, 16), n, 0), limbs([rng.fp() for _ in range(n)]))
    return p, q


def test_sharded_product_equals_glued_pairing(engine, coracle, pairs):
    p, q = pairs
    n = p.shape[0]
    exp = coracle.glued_pairing(proj1(p), proj2(q), np.array([0, n], dtype=np.uint64))
    whole, is_one = engine.pairing_product(p, q)
    assert np.array_equal(whole, exp) and not is_one
    for shards in (1, 2, 3, 5):
        cuts = [n * s // shards for s in range(shards + 1)]
        if shards == 5:
            cuts[2] = cuts[1]                                     # an EMPTY shard: partial = 1
        parts = np.concatenate([engine.pairing_product_partial(p[a:b], q[a:b]) for a, b in zip(cuts, cuts[1:])], axis=0)
        gt, one = engine.fp12_product_final_exp(parts)
        assert np.array_equal(gt, exp) and not one, shards
    # the partial is the glued Miller value of the shard up to a factor in Fp* (isomorphic curves): its final exponentiation is the glued pairing
    part = engine.pairing_product_partial(p[:9], q[:9])
    assert np.array_equal(engine.final_exp(part), coracle.glued_pairing(proj1(p[:9]), proj2(q[:9]), np.array([0, 9], dtype=np.uint64)))
    # empty product and k = 0
    gt, one = engine.fp12_product_final_exp(np.zeros((0, 48), dtype=np.uint64))
    assert one and gt[0, 0] == 1 and not gt[0, 1:].any()
    # a product that IS one: e(P, Q) e(-P, Q), split over two shards
    from helpers import P as PMOD, ints
    negy = limbs([(PMOD - y) % PMOD for y in ints(p[:4, 4:8])])
    pn = np.concatenate([p[:4, :4], negy], axis=1)
    parts = np.concatenate([engine.pairing_product_partial(p[:4], q[:4]), engine.pairing_product_partial(pn, q[:4])], axis=0)
    gt, one = engine.fp12_product_final_exp(parts)
    assert one


def test_all_valid_and_product_all_single_rank(engine, pairs):
    p, q = pairs
    flags = engine.to_device(np.ones(1000, dtype=np.uint8))
    assert engine.all_valid(flags) == 1
    f = np.ones(1000, dtype=np.uint8); f[777] = 0
    assert engine.all_valid(engine.to_device(f)) == 0
    gt, one = engine.pairing_product_all(p[:50], q[:50])
    ref, _ = engine.pairing_product(p[:50], q[:50])
    assert np.array_equal(gt, ref)


def _rccl():
    import torch  # noqa: F401  (makes torch's librccl resolvable first: one RCCL per process)
    from sylow_amd.rccl import quiet_init_env, warm_file
    quiet_init_env()
    for name in (os.path.join(os.path.dirname(__import__("torch").__file__), "lib", "librccl.so"), "librccl.so.1", "/opt/rocm/lib/librccl.so.1"):
        try:
            if os.path.isfile(name):
                warm_file(name)          # a cold page cache turns the first communicator into minutes of 4 KB faults
            return ctypes.CDLL(name, mode=ctypes.RTLD_GLOBAL)
        except OSError:
            continue
    pytest.skip("no librccl on this box")


def test_rccl_entry_points_one_rank_communicator(engine, pairs):
    """sylow_hip_all_valid / sylow_hip_pairing_product_all through a REAL ncclComm_t (world size 1): the dlopen binding, the
    MIN all-reduce and the all-gather + product + final exponentiation path all execute."""
    lib = _rccl()

    class UniqueId(ctypes.Structure):
        _fields_ = [("internal", ctypes.c_char * 128)]

    uid = UniqueId()
    lib.ncclGetUniqueId.argtypes = [ctypes.POINTER(UniqueId)]
    lib.ncclCommInitRank.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, UniqueId, ctypes.c_int]
    lib.ncclCommDestroy.argtypes = [ctypes.c_void_p]
    assert lib.ncclGetUniqueId(ctypes.byref(uid)) == 0
    comm = ctypes.c_void_p()
    lib.ncclGetLastError.restype = ctypes.c_char_p
    rc = lib.ncclCommInitRank(ctypes.byref(comm), 1, uid, 0)
    assert rc == 0, (rc, lib.ncclGetLastError(None))
    try:
        p, q = pairs
        f = np.ones(4097, dtype=np.uint8)
        assert engine.all_valid(engine.to_device(f), comm=comm.value) == 1
        f[4096] = 0
        assert engine.all_valid(engine.to_device(f), comm=comm.value) == 0
        gt, one = engine.pairing_product_all(p[:77], q[:77], comm=comm.value)
        ref, _ = engine.pairing_product(p[:77], q[:77])
        assert np.array_equal(gt, ref) and not one
        # aggregate verification through the communicator (all-gather of one partial) == the local answer
        pk, msgs, sig = signed_batch(engine, 40, same_signer=False)
        gt_c, ok_c = engine.bls_aggregate_verify(pk, msgs, sig, comm=comm.value)
        gt_l, ok_l = engine.bls_aggregate_verify(pk, msgs, sig)
        assert ok_c == ok_l == 1 and np.array_equal(gt_c, gt_l)
    finally:
        engine.sync()
        lib.ncclCommDestroy(comm)


def signed_batch(engine, n, same_signer, seed=SEED + 91):
    rng = Xoshiro(seed)
    sk = limbs([rng.fp()] * n) if same_signer else limbs([rng.fp() for _ in range(n)])
    msgs = [bytes([i % 251, i // 251]) * (1 + i % 11) for i in range(n)]
    sig, _ = engine.bls_sign(sk, msgs)
    pk, _ = engine.g2_scalar_mul(np.repeat(pack(G2, 16), 1 if same_signer else n, 0), sk[:1] if same_signer else sk, subgroup=True)
    return pk, msgs, sig


def reference_shape(engine, pk, msgs, sig, pk_inf=None, sig_inf=None):
    """the 2n pairs (sig_i, G2gen), (-H(m_i), pk_i) as the reference's example builds them, through the generic glued product"""
    from helpers import P as PMOD, ints
    n = len(msgs)
    h, _ = engine.hash_to_g1(msgs)
    hneg = np.concatenate([h[:, :4], limbs([(PMOD - y) % PMOD for y in ints(h[:, 4:8])])], axis=1)
    pkn = np.repeat(pk, n, 0) if pk.shape[0] == 1 else pk
    pin = np.zeros(n, np.uint8) if pk_inf is None else (np.repeat(np.asarray(pk_inf, np.uint8), n) if pk.shape[0] == 1 else np.asarray(pk_inf, np.uint8))
    sin = np.zeros(n, np.uint8) if sig_inf is None else np.asarray(sig_inf, np.uint8)
    p = np.concatenate([sig, hneg]); q = np.concatenate([np.repeat(pack(G2, 16), n, 0), pkn])
    return engine.pairing_product(p, q, p_inf=np.concatenate([sin, np.zeros(n, np.uint8)]), q_inf=np.concatenate([np.zeros(n, np.uint8), pin]), skip_infinity=1)


@pytest.mark.parametrize("same_signer", [False, True])
def test_aggregate_verify_equals_the_glued_product_of_the_reference_shape(engine, coracle, same_signer):
    """sylow_hip_bls_aggregate_verify_batch (signatures summed in G1 first; one key: hashes summed too) yields the Gt value and the
    boolean of the reference's 2n-pair glued product (examples/verify_multiple_messages_same_signer.rs:41-60) -- valid batches,
    a wrong signature, a swapped message, identity signatures / keys, n = 1 and n = 0; and the oracle's glued_pairing."""
    n = 77                                                              # ragged: tree levels with odd counts
    pk, msgs, sig = signed_batch(engine, n, same_signer)
    gt, ok = engine.bls_aggregate_verify(pk, msgs, sig)
    ref, ref_ok = reference_shape(engine, pk, msgs, sig)
    assert ok == 1 and ref_ok and np.array_equal(gt, ref[0] if ref.ndim == 2 else ref)
    one = np.zeros(48, dtype=np.uint64); one[0] = 1
    assert np.array_equal(gt, one)
    # one wrong signature: sig_5 <- sig_6 (a valid point, the wrong signature)
    bad = sig.copy(); bad[5] = sig[6]
    gt_b, ok_b = engine.bls_aggregate_verify(pk, msgs, bad)
    ref_b, refok_b = reference_shape(engine, pk, msgs, bad)
    assert ok_b == 0 and not refok_b and np.array_equal(gt_b, np.asarray(ref_b).reshape(-1))
    # the oracle on the same 2n pairs (first 6 messages)
    from helpers import P as PMOD, ints
    m = 6
    h, _ = engine.hash_to_g1(msgs[:m])
    hneg = np.concatenate([h[:, :4], limbs([(PMOD - y) % PMOD for y in ints(h[:, 4:8])])], axis=1)
    pkm = np.repeat(pk, m, 0) if same_signer else pk[:m]
    exp = coracle.glued_pairing(proj1(np.concatenate([bad[:m], hneg])), proj2(np.concatenate([np.repeat(pack(G2, 16), m, 0), pkm])), np.array([0, 2 * m], dtype=np.uint64))
    gt_m, ok_m = engine.bls_aggregate_verify(pk if same_signer else pk[:m], msgs[:m], bad[:m])
    assert np.array_equal(gt_m, exp[0]) and ok_m == 0
    # swapped messages
    sw = list(msgs); sw[3], sw[4] = sw[4], sw[3]
    # (under ONE key the product does not see the order -- neither does the reference's 2n-pair product)
    assert engine.bls_aggregate_verify(pk, sw, sig)[1] == (1 if same_signer else 0) == int(reference_shape(engine, pk, sw, sig)[1])
    assert engine.bls_aggregate_verify(pk, [b"other"] + msgs[1:], sig)[1] == 0
    # identity flags: pairing() semantics on both sides, against the generic product with the same flags
    g = np.random.default_rng(12)
    sinf = (g.random(n) < 0.2).astype(np.uint8)
    pinf = np.array([0], np.uint8) if same_signer else (g.random(n) < 0.2).astype(np.uint8)
    gt_i, ok_i = engine.bls_aggregate_verify(pk, msgs, sig, pk_inf=pinf, sig_inf=sinf)
    ref_i, refok_i = reference_shape(engine, pk, msgs, sig, pk_inf=pinf, sig_inf=sinf)
    assert np.array_equal(gt_i, np.asarray(ref_i).reshape(-1)) and ok_i == int(refok_i)
    if same_signer:
        gt_k, ok_k = engine.bls_aggregate_verify(pk, msgs, sig, pk_inf=[1])        # identity key: only e(sum sig, G2gen) is left
        ref_k, _ = reference_shape(engine, pk, msgs, sig, pk_inf=[1])
        assert np.array_equal(gt_k, np.asarray(ref_k).reshape(-1)) and ok_k == 0
    # n = 1 and n = 0
    assert engine.bls_aggregate_verify(pk[:1], msgs[:1], sig[:1])[1] == 1
    assert engine.bls_aggregate_verify(pk[:1], msgs[:1], sig[1:2])[1] == 0
    gt0, ok0 = engine.bls_aggregate_verify(pk[:1] if same_signer else pk[:0], [], sig[:0])
    assert ok0 == 1 and np.array_equal(gt0, one)


def test_weighted_batch_verification_vs_oracle(engine, coracle):
    """sylow_hip_bls_batch_verify_weighted == the reference's glued_pairing over the 2n pairs (w_i sig_i, G2gen), (-w_i H(m_i), pk_i)
    (oracle scalar multiplications + glued_pairing), for distinct keys and for one key; a valid batch gives the identity for any
    weights; one wrong signature is caught; a zero weight removes that element from the test."""
    n = 9
    for same_signer in (False, True):
        pk, msgs, sig = signed_batch(engine, n, same_signer=same_signer, seed=SEED + 95)
        rng = Xoshiro(SEED + 96)
        w = limbs([rng.next() | (rng.next() << 64) | 1 for _ in range(n)])
        gt, ok = engine.bls_batch_verify_weighted(pk[:1] if same_signer else pk, msgs, sig, w)
        assert ok
        # the oracle's value of the same product
        one4 = np.zeros((n, 4), dtype=np.uint64); one4[:, 0] = 1
        z4 = np.zeros((n, 4), dtype=np.uint64)
        h_xy, _ = coracle.g1_to_affine(coracle.hash_to_curve(msgs))
        nh = h_xy.copy(); nh[:, 4:] = coracle.fp_op("neg", h_xy[:, 4:])
        wsig, _ = coracle.g1_to_affine(coracle.g1_scalar_mul(np.concatenate([sig, one4], axis=1), w))
        wnh, _ = coracle.g1_to_affine(coracle.g1_scalar_mul(np.concatenate([nh, one4], axis=1), w))
        g2 = np.repeat(pack(G2, 16), n, 0)
        p_all = np.concatenate([np.concatenate([wsig, one4], axis=1), np.concatenate([wnh, one4], axis=1)], axis=0)
        pkn = np.repeat(pk, n, 0) if pk.shape[0] == 1 else pk
        q_all = np.concatenate([np.concatenate([g2, one4, z4], axis=1), np.concatenate([pkn, one4, z4], axis=1)], axis=0)
        exp = coracle.glued_pairing(p_all, q_all, [0, 2 * n])
        assert np.array_equal(gt, exp)
        # one signature replaced by another valid point: the weighted test fails, and passes again once that element's weight is zero
        bad = sig.copy(); bad[4] = sig[5]
        _, ok_bad = engine.bls_batch_verify_weighted(pk[:1] if same_signer else pk, msgs, bad, w)
        assert not ok_bad
        w0 = w.copy(); w0[4] = 0
        _, ok_masked = engine.bls_batch_verify_weighted(pk[:1] if same_signer else pk, msgs, bad, w0)
        assert ok_masked
    # signatures permuted among the messages pass the UNWEIGHTED product (its documented blind spot) and fail the weighted one
    pk, msgs, sig = signed_batch(engine, n, same_signer=True, seed=SEED + 97)
    perm = sig[[1, 0] + list(range(2, n))]
    _, ok_plain = engine.bls_aggregate_verify(pk[:1], msgs, perm)
    _, ok_w = engine.bls_batch_verify_weighted(pk[:1], msgs, perm, limbs([3 + 2 * i for i in range(n)]))
    assert ok_plain and not ok_w


def test_fallback_path_and_collectives_in_one_process(engine, pairs):
    """The staggered launch's recompute fallback (STAGGER option 2: the parking blocks' flags muted) and the RCCL aggregates in ONE process,
    at the smallest skewed size (2^17): bls_verify_batch under the muted skew -> sylow_hip_all_valid over a real one-rank ncclComm_t (one wrong
    signature planted inside the parked range flips it), and pairing_product_all of 2^17 pairs on the same communicator equals the local product."""
    lib = _rccl()

    class UniqueId(ctypes.Structure):
        _fields_ = [("internal", ctypes.c_char * 128)]

    uid = UniqueId()
    lib.ncclGetUniqueId.argtypes = [ctypes.POINTER(UniqueId)]
    lib.ncclCommInitRank.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, UniqueId, ctypes.c_int]
    lib.ncclCommDestroy.argtypes = [ctypes.c_void_p]
    assert lib.ncclGetUniqueId(ctypes.byref(uid)) == 0
    comm = ctypes.c_void_p()
    assert lib.ncclCommInitRank(ctypes.byref(comm), 1, uid, 0) == 0
    n, d = 1 << 17, 64
    prev = engine.get_option("STAGGER")
    try:
        engine.set_option("STAGGER", 2)
        pk64, msgs64, sig64 = signed_batch(engine, d, same_signer=False, seed=SEED + 93)
        idx = np.arange(n) % d
        dpk, dsig = engine.to_device_soa(pk64[idx], 16), engine.to_device_soa(sig64[idx], 8)
        blob = b"".join(msgs64[i] for i in idx)
        off = np.zeros(n + 1, dtype=np.uint64)
        off[1:] = np.cumsum([len(msgs64[i]) for i in idx])
        dm, doff = engine.to_device(np.frombuffer(blob, dtype=np.uint8)), engine.to_device(off)
        ok = engine.empty((n,), np.uint8)
        engine._call("sylow_hip_bls_verify_batch", dpk.ptr, None, dm.ptr, doff.ptr, dsig.ptr, None, ok.ptr, n)
        assert engine.all_valid(ok, comm=comm.value) == 1 and ok.download().all()
        bad = sig64[idx].copy()
        bad[40000] = sig64[(idx[40000] + 1) % d]                       # element 40000 lies in a parked chunk (32768 .. 65535)
        dbad = engine.to_device_soa(bad, 8)
        engine._call("sylow_hip_bls_verify_batch", dpk.ptr, None, dm.ptr, doff.ptr, dbad.ptr, None, ok.ptr, n)
        flags = ok.download()
        assert engine.all_valid(ok, comm=comm.value) == 0 and flags.sum() == n - 1 and flags[40000] == 0
        p, q = pairs
        m = p.shape[0]
        pidx = np.arange(n) % m
        gt_c, one_c = engine.pairing_product_all(p[pidx], q[pidx], comm=comm.value)
        gt_l, one_l = engine.pairing_product(p[pidx], q[pidx])
        assert np.array_equal(gt_c, gt_l) and one_c == one_l
    finally:
        engine.set_option("STAGGER", prev)
        engine.sync()
        lib.ncclCommDestroy(comm)
