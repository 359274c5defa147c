"""Wall latency of ONE element through the Python layer for a broad set of entry points (upload + kernel + download; sorted): finds the
operations whose single call is far from its batch cost."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import numpy as np, sylow_amd
from helpers import Xoshiro, limbs, pack
from test_gpu_pairing import G1, G2
eng = sylow_amd.Engine(0)
rng = Xoshiro(99)
k = limbs([rng.fp()])
P, _ = eng.g1_scalar_mul(pack(G1, 8), k)
Q, _ = eng.g2_scalar_mul(pack(G2, 16), limbs([rng.fp()]))
gt = eng.pairing(P, Q)
f2 = limbs([rng.fp(), rng.fp()]).reshape(1, 8)
msgs = [b"abc"]
sig, _ = eng.bls_sign(k, msgs)
pk, _ = eng.g2_scalar_mul(pack(G2, 16), k)
g1b, g2b = eng.g1_to_be_bytes(P), eng.g2_to_be_bytes(Q)
off = np.array([0, 2], dtype=np.uint64)
P2, Q2 = np.concatenate([P, P]), np.concatenate([Q, Q])
ops = {
    "fp_mul": lambda: eng.fp_mul(k, k), "fp_inv": lambda: eng.fp_inv(k), "fp_sqrt": lambda: eng.fp_sqrt(k), "fp2_inv": lambda: eng.fp2_inv(f2),
    "fp12_mul": lambda: eng.fp12_mul(gt, gt), "fp12_inv": lambda: eng.fp12_inv(gt), "gt_pow": lambda: eng.gt_pow(gt, k),
    "g1_add": lambda: eng.g1_add(P, P), "g1_scalar_mul": lambda: eng.g1_scalar_mul(P, k), "g1_generator_mul": lambda: eng.g1_generator_mul(k),
    "g2_add": lambda: eng.g2_add(Q, Q), "g2_scalar_mul": lambda: eng.g2_scalar_mul(Q, k), "g2_scalar_mul_subgroup": lambda: eng.g2_scalar_mul(Q, k, subgroup=True),
    "g2_generator_mul": lambda: eng.g2_generator_mul(k), "g2_subgroup_check": lambda: eng.g2_subgroup_check(Q), "g2_psi": lambda: eng.g2_psi(Q),
    "g1_from_be_bytes": lambda: eng.g1_from_be_bytes(g1b), "g2_from_be_bytes": lambda: eng.g2_from_be_bytes(g2b),
    "hash_to_field": lambda: eng.hash_to_field(msgs), "hash_to_g1": lambda: eng.hash_to_g1(msgs), "bls_sign": lambda: eng.bls_sign(k, msgs),
    "bls_verify": lambda: eng.bls_verify(pk, msgs, sig, pipelined=False), "bls_verify_two_pairings": lambda: eng.bls_verify(pk, msgs, sig, two_pairings=True),
    "pairing": lambda: eng.pairing(P, Q, pipelined=False), "miller_loop": lambda: eng.miller_loop(P, Q), "final_exp": lambda: eng.final_exp(gt),
    "multi_pairing_1job_2pairs": lambda: eng.multi_pairing(P2, Q2, off, skip_infinity=True), "pairing_product_2": lambda: eng.pairing_product(P2, Q2, skip_infinity=True),
}
res = []
for name, fn in ops.items():
    fn(); eng.sync()
    t0 = time.perf_counter()
    for _ in range(10): fn()
    eng.sync()
    res.append(((time.perf_counter() - t0) / 10 * 1e3, name))
for t, name in sorted(res, reverse=True): print("%-28s %8.3f ms" % (name, t))
