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
    # the partial IS the raw glued Miller value of the shard (product of the per-pair Miller values)
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
    for name in ("librccl.so.1", os.path.join(os.path.dirname(__import__("torch").__file__), "lib", "librccl.so"), "/opt/rocm/lib/librccl.so.1"):
        try:
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
    finally:
        engine.sync()
        lib.ncclCommDestroy(comm)
