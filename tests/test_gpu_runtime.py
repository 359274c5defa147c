"""GPU tests of the runtime around the kernels: the leased scratch workspace under concurrent host threads on separate HIP
streams (every result checked against the oracle), shutdown + reuse, device re-assertion, and a short run of the determinism
soak (tools/soak.py: random batch sizes through every pairing-based entry point, each call issued twice)."""
import os
import subprocess
import sys
import threading

import numpy as np
import pytest

from helpers import SEED, Xoshiro, limbs, pack
from test_gpu_multi_pairing import G1, G2, proj1, proj2

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _inputs(engine, n, seed):
    rng = Xoshiro(seed)
    p, _ = engine.g1_scalar_mul(np.repeat(pack(G1, 8), n, 0), limbs([rng.fp() for _ in range(n)]))
    q, _ = engine.g2_scalar_mul(np.repeat(pack(G2, 16), n, 0), limbs([rng.fp() for _ in range(n)]))
    return p, q


def _evm_blob(p, q):
    """EIP-197 encoding of pairs (P_i, Q_i): x | y | x.c1 | x.c0 | y.c1 | y.c0, 32-byte big-endian each"""
    def be(col):
        return col[:, ::-1].astype(">u8").view(np.uint8).reshape(len(col), 32)
    return np.concatenate([be(p[:, 0:4]), be(p[:, 4:8]), be(q[:, 4:8]), be(q[:, 0:4]), be(q[:, 12:16]), be(q[:, 8:12])], axis=1)


def test_two_threads_two_streams_workspace_users(engine, coracle):
    """pairing_product_batch, evm_ecpairing_batch, bls_verify_same_signer_batch and bls_aggregate_verify_batch interleaved from two host threads on two
    non-default streams: each call leases its own scratch block, so no result may ever differ from the oracle's."""
    import torch

    import sylow_amd
    n = 96
    p, q = _inputs(engine, n, SEED + 100)
    exp_prod = {m: coracle.glued_pairing(proj1(p[:m]), proj2(q[:m]), np.array([0, m], dtype=np.uint64)) for m in (17, 64, 96)}
    # ecPairing jobs of two pairs: e(P, Q) e(-P, Q) -> true; odd jobs use another P in the second pair -> false
    from helpers import P as PMOD, ints
    negy = limbs([(PMOD - y) % PMOD for y in ints(p[:, 4:8])])
    pn = np.concatenate([p[:, :4], negy], axis=1)
    pos, neg = _evm_blob(p, q), _evm_blob(pn, q)
    nj = 48
    jobs = np.concatenate([pos[:nj], neg[:nj]], axis=1).copy()
    jobs[1::2, 192:256] = pos[(np.arange(nj)[1::2] + 5) % n, :64]
    exp_evm = (np.arange(nj) % 2 == 0).astype(np.uint8)
    rng = Xoshiro(SEED + 101)
    sk = limbs([rng.fp()])
    msgs = [bytes([i]) * (1 + i % 7) for i in range(40)]
    sig, _ = engine.bls_sign(np.repeat(sk, 40, 0), msgs)
    pk, _ = engine.g2_scalar_mul(pack(G2, 16), sk)
    sig_bad = sig.copy(); sig_bad[11] = sig[12]
    exp_ver = np.ones(40, dtype=np.uint8); exp_ver[11] = 0

    errors = []

    def worker(tid):
        try:
            st = torch.cuda.Stream()
            eng = sylow_amd.Engine(0, stream=st.cuda_stream)
            for it in range(12):
                m = (17, 64, 96)[(it + tid) % 3]
                g, _ = eng.pairing_product(p[:m], q[:m])
                if not np.array_equal(g, exp_prod[m]):
                    errors.append(("product", tid, it, m))
                d_in = eng.to_device(jobs.reshape(-1)); d_off = eng.to_device(np.arange(nj + 1, dtype=np.uint64) * np.uint64(2))
                d_res, d_st = eng.empty((nj,), np.uint8), eng.empty((nj,), np.uint8)
                eng._call("sylow_hip_evm_ecpairing_batch", d_in.ptr, d_off.ptr, nj, 2 * nj, d_res.ptr, d_st.ptr)
                if not (np.array_equal(d_res.download(), exp_evm) and not d_st.download().any()):
                    errors.append(("evm", tid, it))
                if not np.array_equal(eng.bls_verify_same_signer(pk, msgs, sig_bad), exp_ver):
                    errors.append(("same_signer", tid, it))
                # aggregate verification nests three leases (hash / sums, and the two product trees)
                if eng.bls_aggregate_verify(pk, msgs, sig)[1] != 1 or eng.bls_aggregate_verify(pk, msgs, sig_bad)[1] != 0:
                    errors.append(("aggregate", tid, it))
                if eng.bls_aggregate_verify(np.repeat(pk, 40, 0), msgs, sig_bad if it % 2 else sig)[1] != (0 if it % 2 else 1):
                    errors.append(("aggregate_keys", tid, it))
        except Exception as e:  # noqa: BLE001
            errors.append(("exception", tid, repr(e)))

    th = [threading.Thread(target=worker, args=(t,)) for t in range(2)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errors, errors[:5]


def test_two_threads_staggered_launches(engine, coracle):
    """Two host threads on two non-default streams, each launching the skewed kernels (k_pairing / k_bls_verify_fused from 2^17 elements on:
    parked Miller values, per-block flags and finishing blocks in a leased block per call) at the same time -- two skewed grids share the GPU, so
    finishing blocks of one wait while the other holds slots.  Every row of every call against oracle values (64 distinct pairs / triples tiled)."""
    import torch

    import sylow_amd
    n, d = (1 << 17) + 40, 64
    rng = np.random.default_rng(616)
    p64, _ = engine.g1_scalar_mul(np.repeat(pack(G1, 8), d, 0), limbs([int(x) for x in rng.integers(1, 1 << 62, size=d)]))
    q64, _ = engine.g2_scalar_mul(np.repeat(pack(G2, 16), d, 0), limbs([int(x) for x in rng.integers(1, 1 << 62, size=d)]))
    gt64 = coracle.pairing(proj1(p64), proj2(q64))
    sk = limbs([int(x) for x in rng.integers(1, 1 << 62, size=d)])
    msgs64 = [bytes([i, 3 * i & 255]) * (1 + i % 4) for i in range(d)]
    sig64, _ = engine.bls_sign(sk, msgs64)
    pk64, _ = engine.g2_scalar_mul(np.repeat(pack(G2, 16), d, 0), sk)
    idx = np.arange(n) % d
    bad = np.array([1, 32768, 50000, 65535, 70000, n - 1])
    sig = sig64[idx].copy(); sig[bad] = sig64[(idx[bad] + 1) % d]
    want = np.ones(n, dtype=np.uint8); want[bad] = 0
    blob = np.frombuffer(b"".join(msgs64[i] for i in idx), dtype=np.uint8)
    off = np.zeros(n + 1, dtype=np.uint64); off[1:] = np.cumsum([len(msgs64[i]) for i in idx])
    errors = []

    def worker(tid):
        try:
            st = torch.cuda.Stream()
            eng = sylow_amd.Engine(0, stream=st.cuda_stream)
            dp, dq = eng.to_device_soa(p64[idx], 8), eng.to_device_soa(q64[idx], 16)
            dpk, dsig = eng.to_device_soa(pk64[idx], 16), eng.to_device_soa(sig, 8)
            dm, doff = eng.to_device(blob), eng.to_device(off)
            gt, ok = eng.empty((48, n)), eng.empty((n,), np.uint8)
            for it in range(3):
                if (it + tid) % 2:
                    eng._call("sylow_hip_pairing_batch", dp.ptr, None, dq.ptr, None, gt.ptr, n)
                    eng._call("sylow_hip_bls_verify_batch", dpk.ptr, None, dm.ptr, doff.ptr, dsig.ptr, None, ok.ptr, n)
                else:
                    eng._call("sylow_hip_bls_verify_batch", dpk.ptr, None, dm.ptr, doff.ptr, dsig.ptr, None, ok.ptr, n)
                    eng._call("sylow_hip_pairing_batch", dp.ptr, None, dq.ptr, None, gt.ptr, n)
                if not np.array_equal(eng.from_device_soa(gt), gt64[idx]):
                    errors.append(("pairing", tid, it))
                if not np.array_equal(ok.download(), want):
                    errors.append(("verify", tid, it))
        except Exception as e:  # noqa: BLE001
            errors.append(("exception", tid, repr(e)))

    th = [threading.Thread(target=worker, args=(t,)) for t in range(2)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errors, errors[:5]


def test_shutdown_then_reuse(engine, coracle):
    p, q = _inputs(engine, 20, SEED + 102)
    before, _ = engine.pairing_product(p, q)
    msgs = [b"a", b"bc"]
    rng = Xoshiro(SEED + 103)
    sk = limbs([rng.fp(), rng.fp()])
    sig, _ = engine.bls_sign(sk, msgs)
    pk, _ = engine.g2_scalar_mul(np.repeat(pack(G2, 16), 2, 0), sk)
    assert engine.bls_verify(pk, msgs, sig, fused=True).tolist() == [1, 1]         # builds the generator line table
    engine.shutdown()                                                              # frees blocks, events and the table
    after, _ = engine.pairing_product(p, q)                                        # state is rebuilt on demand
    assert np.array_equal(before, after)
    assert engine.bls_verify(pk, msgs, sig, fused=True).tolist() == [1, 1]
    engine.shutdown()
    engine.shutdown()                                                              # idempotent
    # trim: idle, completed scratch blocks are handed back; the next call leases afresh and computes the same
    again, _ = engine.pairing_product(p, q)
    engine.sync()
    engine.trim(0)
    engine.trim(1 << 40)                                                           # nothing is above the threshold: a no-op
    final, _ = engine.pairing_product(p, q)
    assert np.array_equal(before, again) and np.array_equal(before, final)


def test_set_device_is_reasserted_and_checked(engine):
    lib = engine.lib
    assert lib.sylow_hip_set_device(0) == 0
    assert lib.sylow_hip_set_device(lib.sylow_hip_device_count()) != 0             # out of range -> HIP error code, not a crash
    assert lib.sylow_hip_set_device(-1) == -2
    ids = (__import__("ctypes").c_int32 * 1)(0)
    assert lib.sylow_hip_init_devices(ids, 1) == 0
    assert lib.sylow_hip_init_devices(None, 0) == -2
    a = limbs([3, 5])
    assert engine.fp_mul(a[:1], a[1:]).tolist() == limbs([15]).tolist()


def test_soak_short():
    """~20 s of tools/soak.py as a fresh child process (it asserts determinism and oracle parity internally)."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "soak.py"), "20"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "soak ok" in r.stdout


def test_soak_large_short():
    """~25 s of tools/soak_large.py as a fresh child process: the LARGE-batch kernels (k_pairing, the fused verifier, the line-table multi-pair
    route, byte-level ecPairing at 2^14 jobs) called over and over -- every repetition bit-identical to the first, the first checked against the
    oracle / the planted pattern."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "soak_large.py"), "25", "16"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "soak_large ok" in r.stdout


def test_scratch_limit_bounds_the_line_tables_and_keeps_results(engine, coracle):
    """sylow_hip_set_scratch_limit: the multi-pair routes' line tables (19.5 KB per pair, by default whole rounds of 2^16 jobs up to 12 GB) shrink
    to the host's bound -- job slices below one round, batch-wide products on the in-register schedule -- and every result stays the same:
    4096 three-pair jobs and a 140 000-pair product (three pairs per lane pair) under a 64 MB bound against the unbounded run and (a sample) the oracle; device memory held
    by the library afterwards stays far below the 2.9 GB of tables the unbounded calls lease."""
    import torch
    from helpers import pack, limbs
    from oracle import pyref as R
    g1, g2 = [1, 2], list(R.G2_GEN_AFF[0]) + list(R.G2_GEN_AFF[1])
    nj, k = 4096, 3
    n = nj * k
    rng = np.random.default_rng(77)
    a = limbs([int(x) for x in rng.integers(1, 1 << 62, size=n)])
    b = limbs([int(x) for x in rng.integers(1, 1 << 62, size=n)])
    p, _ = engine.g1_scalar_mul(np.repeat(pack(g1, 8), n, 0), a)
    q, _ = engine.g2_scalar_mul(np.repeat(pack(g2, 16), n, 0), b)
    off = np.arange(nj + 1, dtype=np.uint64) * np.uint64(k)
    gt0, one0 = engine.multi_pairing(p, q, off, skip_infinity=True)
    pp, qq = np.tile(p, (12, 1))[:140000], np.tile(q, (12, 1))[:140000]
    prod0, _ = engine.pairing_product(pp, qq, skip_infinity=True)
    engine.sync(); engine.trim(0)
    free0 = torch.cuda.mem_get_info()[0]
    try:
        engine.set_scratch_limit(64 << 20)
        gt1, one1 = engine.multi_pairing(p, q, off, skip_infinity=True)
        prod1, _ = engine.pairing_product(pp, qq, skip_infinity=True)
        engine.sync()
        held = free0 - torch.cuda.mem_get_info()[0]
    finally:
        engine.set_scratch_limit(0)
    assert np.array_equal(gt0, gt1) and np.array_equal(one0, one1) and np.array_equal(prod0, prod1)
    # what the library holds after the bounded calls: the 64 MB of tables + the n-proportional buffers (raw Miller values, product tree) + whatever
    # the driver keeps of the freed staging arrays -- far from the 2.9 GB of tables the same two calls lease without a bound
    assert held <= (512 << 20), held
    idx = np.sort(rng.choice(nj, 24, replace=False))
    rows = (idx[:, None] * k + np.arange(k)[None, :]).reshape(-1)
    one4 = np.zeros((rows.size, 4), dtype=np.uint64); one4[:, 0] = 1
    exp = coracle.glued_pairing(np.concatenate([p[rows], one4], axis=1), np.concatenate([q[rows], one4, np.zeros((rows.size, 4), dtype=np.uint64)], axis=1),
                                np.arange(25, dtype=np.uint64) * np.uint64(k))
    assert np.array_equal(gt1[idx], exp)
    engine.trim(0)


def test_options_and_clock_probe_on_the_metric_kernels(engine, coracle):
    """Round 6: the route selectors are ABI options (sylow_hip_set_option), and the metric's kernels carry a live clock probe
    (sylow_hip_clock_probe).  One batch of 2^17 + 77 pairings under STAGGER = 0 / 1 / 2 gives identical Gt values (2: every finishing
    block takes its recompute fallback); with the probe armed the wavefront count is the launch's, the sustained clock is a plausible
    engine clock, the results do not change, and a disarmed probe leaves the accumulator alone."""
    n, d = (1 << 17) + 77, 32
    rng = np.random.default_rng(616)
    a = limbs([int(x) for x in rng.integers(1, 1 << 62, size=d)])
    b = limbs([int(x) for x in rng.integers(1, 1 << 62, size=d)])
    p32, _ = engine.g1_scalar_mul(np.repeat(pack(G1, 8), d, 0), a)
    q32, _ = engine.g2_scalar_mul(np.repeat(pack(G2, 16), d, 0), b)
    one4 = np.zeros((d, 4), dtype=np.uint64); one4[:, 0] = 1
    gt32 = coracle.pairing(np.concatenate([p32, one4], axis=1), np.concatenate([q32, one4, np.zeros((d, 4), dtype=np.uint64)], axis=1))
    idx = np.arange(n) % d
    dp, dq, dg = engine.to_device_soa(p32[idx], 8), engine.to_device_soa(q32[idx], 16), engine.empty((48, n))
    prev = engine.get_option("STAGGER")
    acc = engine.empty((256,)).upload(np.zeros(256, dtype=np.uint64))
    khz = engine.wall_clock_khz()
    assert 1000 <= khz <= 1000000
    try:
        outs = {}
        for mode in (0, 1, 2):
            engine.set_option("STAGGER", mode)
            assert engine.get_option("STAGGER") == mode
            if mode == 1:
                engine.clock_probe(acc)
            engine._call("sylow_hip_pairing_batch", dp.ptr, None, dq.ptr, None, dg.ptr, n)
            engine.sync()
            if mode == 1:
                engine.clock_probe(None)
            outs[mode] = engine.from_device_soa(dg)
        assert np.array_equal(outs[0], gt32[idx])
        assert np.array_equal(outs[1], outs[0]) and np.array_equal(outs[2], outs[0])
        words = acc.download()
        mhz, ticks, waves, longest_ms = engine.clock_probe_summary(words, khz)
        blocks = (2 * n + 255) // 256
        cus = (waves // 4 - blocks)                                  # the skewed launch adds one finishing block per parked block
        assert waves % 4 == 0 and cus in (0, 128, 256, 304, 512), (waves, blocks)
        assert 500.0 < mhz < 3500.0, mhz
        assert 0.5 < longest_ms < 200.0, longest_ms
        # disarmed: another launch adds nothing
        engine._call("sylow_hip_pairing_batch", dp.ptr, None, dq.ptr, None, dg.ptr, n)
        engine.sync()
        assert np.array_equal(acc.download(), words)
    finally:
        engine.clock_probe(None)
        engine.set_option("STAGGER", prev)
