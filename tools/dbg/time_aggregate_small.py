import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, sylow_amd
from helpers import Xoshiro, limbs, pack
from test_gpu_pairing import G1, G2
eng = sylow_amd.Engine(0)
rng = Xoshiro(5)
for n in (1, 8, 64, 512):
    sk = limbs([rng.fp() for _ in range(n)])
    msgs = [bytes([i & 255] * 20) for i in range(n)]
    sig, _ = eng.bls_sign(sk, msgs)
    pk, _ = eng.g2_scalar_mul(np.repeat(pack(G2, 16), n, 0), sk)
    sig1, _ = eng.bls_sign(np.repeat(sk[:1], n, 0), msgs)
    w = limbs([rng.fp() >> 130 for _ in range(n)])
    ops = {"aggregate_verify": lambda: eng.bls_aggregate_verify(pk, msgs, sig), "aggregate_same_signer": lambda: eng.bls_aggregate_verify(pk[:1], msgs, sig1),
           "batch_verify_weighted": lambda: eng.bls_batch_verify_weighted(pk, msgs, sig, w), "verify_same_signer": lambda: eng.bls_verify_same_signer(pk[:1], msgs, sig1),
           "verify": lambda: eng.bls_verify(pk, msgs, sig, pipelined=False)}
    out = []
    for name, fn in ops.items():
        r = fn(); eng.sync()
        t0 = time.perf_counter()
        for _ in range(5): fn()
        eng.sync()
        out.append("%s %.2f" % (name, (time.perf_counter() - t0) / 5 * 1e3))
    print("n=%d: " % n + "  ".join(out))
