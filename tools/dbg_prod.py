"""Stress harness for sylow_hip_pairing_product_batch: repeats the batch-wide product and counts mismatches against the oracle
(this is how the stream-ordered-allocator corruption was found and the workspace fix verified)."""
import sys, numpy as np
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
from helpers import SEED, Xoshiro, limbs, pack
from oracle import coracle as C, pyref as R
import sylow_amd
from test_gpu_multi_pairing import proj1, proj2, G1, G2
eng = sylow_amd.Engine(0)
rng = Xoshiro(5)
N = 301
p, _ = eng.g1_scalar_mul(np.repeat(pack(G1, 8), N, 0), limbs([rng.fp() for _ in range(N)]))
q, _ = eng.g2_scalar_mul(np.repeat(pack(G2, 16), N, 0), limbs([rng.fp() for _ in range(N)]))


for n in (64, 128, 301):
    pinf = np.zeros(n, np.uint8); qinf = np.zeros(n, np.uint8)
    exp = C.glued_pairing(proj1(p[:n]), proj2(q[:n]), np.array([0, n], dtype=np.uint64))
    bad = {"noflags": 0, "zeroflags_skip": 0, "zeroflags_replay": 0}
    for rep in range(40):
        g, _ = eng.pairing_product(p[:n], q[:n]); bad["noflags"] += not np.array_equal(g, exp)
        g, _ = eng.pairing_product(p[:n], q[:n], p_inf=pinf, q_inf=qinf, skip_infinity=True); bad["zeroflags_skip"] += not np.array_equal(g, exp)
        g, _ = eng.pairing_product(p[:n], q[:n], p_inf=pinf, q_inf=qinf); bad["zeroflags_replay"] += not np.array_equal(g, exp)
    print(n, bad)
