"""Stress harness for the scratch-workspace users (the round-1 "allocator incident"): repeats sylow_hip_pairing_product_batch and
counts results that differ from the oracle, on the default stream and on an explicit stream.
    python tools/dbg_prod.py                         # the leased workspace (product path)
    SYLOW_HIP_WS_ASYNC=1 python tools/dbg_prod.py    # the same calls on hipMallocAsync / hipFreeAsync blocks
Round 1 measured ~7 % wrong products with hipMallocAsync on the legacy default stream and replaced the allocator."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import ctypes
import numpy as np
SYSTEM_RT = os.environ.get("SYLOW_HIP_SYSTEM_RUNTIME") == "1"       # /opt/rocm's HIP runtime instead of the one torch ships (no torch import)
if not SYSTEM_RT:
    import torch
from helpers import Xoshiro, limbs, pack
from oracle import coracle as C
import sylow_amd
from test_gpu_multi_pairing import proj1, proj2, G1, G2

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = Xoshiro(5)
N = 301
eng0 = sylow_amd.Engine(0)
if SYSTEM_RT:
    hip = ctypes.CDLL("libamdhip64.so.7")                 # already loaded by libsylow_hip.so: same runtime instance
    sp = ctypes.c_void_p()
    assert hip.hipStreamCreate(ctypes.byref(sp)) == 0
    streams = {"default-stream": None, "explicit-stream": sp.value}
else:
    torch.cuda.set_device(0)
    streams = {"default-stream": None, "explicit-stream": torch.cuda.Stream().cuda_stream}
p, _ = eng0.g1_scalar_mul(np.repeat(pack(G1, 8), N, 0), limbs([rng.fp() for _ in range(N)]))
q, _ = eng0.g2_scalar_mul(np.repeat(pack(G2, 16), N, 0), limbs([rng.fp() for _ in range(N)]))
mode = ("hipMallocAsync blocks" if os.environ.get("SYLOW_HIP_WS_ASYNC") == "1" else "leased workspace") + (", ROCm system runtime" if SYSTEM_RT else ", torch's HIP runtime")
total_bad = 0
for sname, st in streams.items():
    eng = sylow_amd.Engine(0, stream=st)
    for n in (64, 128, 301):
        pinf = np.zeros(n, np.uint8); qinf = np.zeros(n, np.uint8)
        exp = C.glued_pairing(proj1(p[:n]), proj2(q[:n]), np.array([0, n], dtype=np.uint64))
        bad = {"noflags": 0, "zeroflags_skip": 0, "zeroflags_replay": 0}
        for rep in range(reps):
            g, _ = eng.pairing_product(p[:n], q[:n]); bad["noflags"] += not np.array_equal(g, exp)
            g, _ = eng.pairing_product(p[:n], q[:n], p_inf=pinf, q_inf=qinf, skip_infinity=True); bad["zeroflags_skip"] += not np.array_equal(g, exp)
            g, _ = eng.pairing_product(p[:n], q[:n], p_inf=pinf, q_inf=qinf); bad["zeroflags_replay"] += not np.array_equal(g, exp)
        total_bad += sum(bad.values())
        print(f"[{mode}] {sname} n={n} reps={reps} mismatches={bad}", flush=True)
print(f"[{mode}] TOTAL mismatches {total_bad} of {2 * 3 * 3 * reps}")
