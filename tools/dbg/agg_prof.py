import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np, sylow_amd
from helpers import Xoshiro, limbs, pack
from test_gpu_pairing import G1, G2
eng = sylow_amd.Engine(0)
rng = Xoshiro(5)
n = 64
sk = limbs([rng.fp() for _ in range(n)])
msgs = [bytes([i & 255] * 20) for i in range(n)]
sig, _ = eng.bls_sign(sk, msgs)
pk, _ = eng.g2_scalar_mul(np.repeat(pack(G2, 16), n, 0), sk)
w = limbs([rng.fp() >> 130 for _ in range(n)])
eng.sync()
eng._call("sylow_hip_fp_neg_batch", eng.to_device_soa(sk, 4).ptr, eng.empty((4, n)).ptr, n)   # marker
for _ in range(3): eng.bls_aggregate_verify(pk, msgs, sig)
eng._call("sylow_hip_fp_neg_batch", eng.to_device_soa(sk, 4).ptr, eng.empty((4, n)).ptr, n)   # marker
for _ in range(3): eng.bls_batch_verify_weighted(pk, msgs, sig, w)
eng.sync()
