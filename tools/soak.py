"""Soak / determinism harness: random batch sizes through every pairing-based entry point, each call issued twice (results must be
bit-identical) and a slice checked against the oracle.  `python tools/soak.py [seconds [seed]]`"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from helpers import Xoshiro, limbs, pack
from oracle import coracle as C, pyref as R
import sylow_amd
from test_gpu_multi_pairing import proj1, proj2, G1, G2

eng = sylow_amd.Engine(0)
rng = Xoshiro(int(sys.argv[2], 0) if len(sys.argv) > 2 else 0xC0FFEE)
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0      # tests/test_gpu_runtime.py runs 20 s of it
t0 = time.time()
NMAX = 4096
base_k = limbs([rng.fp() for _ in range(NMAX)])
P_all, _ = eng.g1_scalar_mul(np.repeat(pack(G1, 8), NMAX, 0), base_k)
Q_all, _ = eng.g2_scalar_mul(np.repeat(pack(G2, 16), NMAX, 0), limbs([rng.fp() for _ in range(NMAX)]))
rounds = checks = 0
while time.time() - t0 < budget:
    n = 1 + rng.next() % NMAX
    s = rng.next() % (NMAX - n + 1)
    p, q = P_all[s:s + n], Q_all[s:s + n]
    pinf = np.array([(rng.next() % 17 == 0) for _ in range(n)], np.uint8)
    qinf = np.array([(rng.next() % 19 == 0) for _ in range(n)], np.uint8)
    a = eng.pairing(p, q, p_inf=pinf, q_inf=qinf); b = eng.pairing(p, q, p_inf=pinf, q_inf=qinf)
    assert np.array_equal(a, b), ("pairing nondeterministic", n)
    rows = sorted(set(list(range(min(n, 3))) + [int(rng.next() % n) for _ in range(3)] + [n - 1]))    # both halves of a packed wavefront, the tail
    exp = C.final_exponentiation(C.miller_loop(p[rows], q[rows]))
    one = np.zeros(48, np.uint64); one[0] = 1
    exp[(pinf[rows] | qinf[rows]).astype(bool)] = one
    assert np.array_equal(a[rows], exp), ("pairing parity", n)
    # glued jobs with random sizes
    sizes = []
    left = n
    while left > 0:
        k = min(left, 1 + rng.next() % 7); sizes.append(k); left -= k
    off = np.concatenate([[0], np.cumsum(sizes)]).astype(np.uint64)
    g1, o1 = eng.multi_pairing(p, q, off, p_inf=pinf, q_inf=qinf, skip_infinity=True)
    g2, o2 = eng.multi_pairing(p, q, off, p_inf=pinf, q_inf=qinf, skip_infinity=True)
    assert np.array_equal(g1, g2) and np.array_equal(o1, o2), ("multi_pairing nondeterministic", n)
    j = int(rng.next() % len(sizes))
    lo, hi = int(off[j]), int(off[j + 1])
    keep = [i for i in range(lo, hi) if not (pinf[i] or qinf[i])]
    e = C.glued_pairing(proj1(p[keep]), proj2(q[keep]), np.array([0, len(keep)], dtype=np.uint64))
    assert np.array_equal(g1[j:j + 1], e), ("multi_pairing parity", n, j)
    # batch-wide product
    x1, _ = eng.pairing_product(p, q, p_inf=pinf, q_inf=qinf, skip_infinity=True)
    x2, _ = eng.pairing_product(p, q, p_inf=pinf, q_inf=qinf, skip_infinity=True)
    assert np.array_equal(x1, x2), ("product nondeterministic", n)
    prod = g1[0:1]
    for jj in range(1, g1.shape[0]):
        prod = C.fp12_op("mul", prod, g1[jj:jj + 1])
    assert np.array_equal(x1, prod), ("product != product of jobs", n)
    # BLS: sign, then the three verifiers with planted corruptions
    nv = min(n, 2048 if rounds % 4 == 3 else 512)     # every fourth round up to the cap of the one-wavefront verification route
    sk = base_k[s:s + nv]
    msgs = [bytes([(i * 7 + rounds) & 255] * (1 + (i % 40))) for i in range(nv)]
    sig, sinf = eng.bls_sign(sk, msgs)
    sig2, _ = eng.bls_sign(sk, msgs)
    assert np.array_equal(sig, sig2), ("sign nondeterministic", nv)
    ks = [int(rng.next() % nv) for _ in range(3)]          # a slice of the signatures and hashes against the oracle (eight-lane routes up to 16384)
    es, _ = C.g1_to_affine(C.sign(sk[ks], [msgs[i] for i in ks]))
    assert np.array_equal(sig[ks], es), ("sign parity", nv, ks)
    hx, _ = eng.hash_to_g1([msgs[i] for i in ks])
    eh, _ = C.g1_to_affine(C.hash_to_curve([msgs[i] for i in ks]))
    assert np.array_equal(hx, eh), ("hash_to_g1 parity", nv, ks)
    pk, _ = eng.g2_scalar_mul(np.repeat(pack(G2, 16), nv, 0), sk)
    badi = set(int(rng.next() % nv) for _ in range(3))
    sigb = sig.copy()
    for i in badi: sigb[i] = sig[(i + 1) % nv] if nv > 1 else P_all[0]
    want = [0 if (i in badi and nv > 1) else 1 for i in range(nv)]
    if nv == 1: want = [0] if badi else [1]
    for fused in (False, True):       # False: the literal two-pairing form, True: the default (one final exponentiation)
        v1 = eng.bls_verify(pk, msgs, sigb, two_pairings=not fused); v2 = eng.bls_verify(pk, msgs, sigb, two_pairings=not fused)
        assert np.array_equal(v1, v2), ("verify nondeterministic", fused, n)
        assert v1.tolist() == want or nv == 1, ("verify flags", fused, n, v1.tolist()[:8], want[:8])
    # the endomorphism-split G2 product against the generic one, and aggregate verification against the per-element flags
    k2 = limbs([rng.fp() for _ in range(nv)])
    ga, gai = eng.g2_scalar_mul(pk, k2, subgroup=True); gb, gbi = eng.g2_scalar_mul(pk, k2)
    assert np.array_equal(ga, gb) and np.array_equal(gai, gbi), ("g2 split product", n)
    ok_all = eng.bls_aggregate_verify(pk, msgs, sig)[1]
    ok_bad = eng.bls_aggregate_verify(pk, msgs, sigb)[1]
    # the product only sees the SUM of the signatures (like the reference's glued product): a permutation of them still passes
    same_multiset = sorted(map(bytes, sigb)) == sorted(map(bytes, sig))
    assert ok_all == 1 and ok_bad == (1 if same_multiset else 0), ("aggregate", n, ok_all, ok_bad, same_multiset)
    gt_a, _ = eng.bls_aggregate_verify(pk, msgs, sigb); gt_b, _ = eng.bls_aggregate_verify(pk, msgs, sigb)
    assert np.array_equal(gt_a, gt_b), ("aggregate nondeterministic", n)
    rounds += 1; checks += 15
print(f"soak ok: {rounds} rounds, {checks} cross-checks, {time.time() - t0:.0f} s")
