#!/bin/bash
# usage: ab_verify.sh libA.so libB.so [rounds]  -- alternate two builds of libsylow_hip.so on ONE box (separate processes):
# bls_verify / fused / same-signer at 2^20
A=$1; B=$2; R=${3:-2}
for r in $(seq $R); do
  for L in $A $B; do
    SYLOW_HIP_LIB=$L python - <<PY
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, torch, sylow_amd
from bench import limbs_row, G2
eng = sylow_amd.Engine(0)
nv = 1 << 20
rng = np.random.default_rng(7)
msgs = rng.integers(0, 256, size=(nv, 32), dtype=np.uint8)
dm, doff = eng.to_device(msgs.reshape(-1)), eng.to_device(np.arange(nv + 1, dtype=np.uint64) * np.uint64(32))
sk = eng.empty((4, nv)).upload(eng.xoshiro_fp_soa(99, nv))
g2 = eng.empty((16, nv)).upload(np.repeat(limbs_row(G2).T, nv, axis=1))
pk, pki, sig, sigi, ok = eng.empty((16, nv)), eng.empty((nv,), np.uint8), eng.empty((8, nv)), eng.empty((nv,), np.uint8), eng.empty((nv,), np.uint8)
eng._call("sylow_hip_g2_scalar_mul_batch", g2.ptr, None, sk.ptr, pk.ptr, pki.ptr, nv)
eng._call("sylow_hip_bls_sign_batch", sk.ptr, dm.ptr, doff.ptr, sig.ptr, sigi.ptr, nv)
def timed(fn, reps=3):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps
tv = timed(lambda: eng._call("sylow_hip_bls_verify_two_pairings_batch", pk.ptr, None, dm.ptr, doff.ptr, sig.ptr, None, ok.ptr, nv))
assert ok.download().all()
tf = timed(lambda: eng._call("sylow_hip_bls_verify_fused_batch", pk.ptr, None, dm.ptr, doff.ptr, sig.ptr, None, ok.ptr, nv))
assert ok.download().all()
# same signer: one key, its signatures
sk1 = eng.empty((4, nv)).upload(np.repeat(eng.xoshiro_fp_soa(5, 1), nv, axis=1))
eng._call("sylow_hip_g2_scalar_mul_batch", g2.ptr, None, sk1.ptr, pk.ptr, pki.ptr, nv)
eng._call("sylow_hip_bls_sign_batch", sk1.ptr, dm.ptr, doff.ptr, sig.ptr, sigi.ptr, nv)
pk1 = eng.empty((16, 1)).upload(pk.download()[:, :1].copy())
ts = timed(lambda: eng._call("sylow_hip_bls_verify_same_signer_batch", pk1.ptr, None, dm.ptr, doff.ptr, sig.ptr, None, ok.ptr, nv))
assert ok.download().all()
print("%-12s verify %.2f ms  fused %.2f ms  same-signer %.2f ms" % (os.path.basename(os.environ["SYLOW_HIP_LIB"]), tv, tf, ts))
PY
  done
done
