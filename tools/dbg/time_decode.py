"""G2 wire-format decode (sylow_hip_g2_from_be_bytes_batch: canonical check + twist equation + subgroup) and the EIP-197 pair decode behind
the byte-level ecPairing call, at 2^18 encodings, for the library SYLOW_HIP_LIB names."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch, sylow_amd
from bench import make_points
eng = sylow_amd.Engine(0)
def timed(fn, reps=3):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps
n = 1 << 18
p, q, ka, kb = make_points(eng, n, 5)
b1, b2 = eng.empty((n * 64,), np.uint8), eng.empty((n * 128,), np.uint8)
eng._call("sylow_hip_g1_to_be_bytes_batch", p.ptr, None, b1.ptr, n)
eng._call("sylow_hip_g2_to_be_bytes_batch", q.ptr, None, b2.ptr, n)
o2, o2i, st = eng.empty((16, n)), eng.empty((n,), np.uint8), eng.empty((n,), np.uint8)
name = os.path.basename(os.environ.get("SYLOW_HIP_LIB", "current"))
print("%s g2_from_bytes 2^18: %.2f ms" % (name, timed(lambda: eng._call("sylow_hip_g2_from_be_bytes_batch", b2.ptr, o2.ptr, o2i.ptr, st.ptr, n))))
assert not st.download().any()
# ecPairing: 2^16 jobs of two pairs (P, Q), (-P, Q)
nj = 1 << 16
ny = eng.empty((4, n)); eng._call("sylow_hip_fp_neg_batch", p.ptr + 4 * n * 8, ny.ptr, n)
pneg = eng.empty((8, n)).upload(np.concatenate([p.download()[:4], ny.download()], axis=0))
b1n = eng.empty((n * 64,), np.uint8); eng._call("sylow_hip_g1_to_be_bytes_batch", pneg.ptr, None, b1n.ptr, n)
g1b, g1nb, g2b = (x.download().reshape(n, -1)[:nj] for x in (b1, b1n, b2))
jobs = np.concatenate([g1b, g2b, g1nb, g2b], axis=1)
d_in = eng.to_device(np.ascontiguousarray(jobs).reshape(-1)); d_off = eng.to_device(np.arange(nj + 1, dtype=np.uint64) * np.uint64(2))
d_res, d_st = eng.empty((nj,), np.uint8), eng.empty((nj,), np.uint8)
print("%s ecpairing bytes 2^16 k=2: %.2f ms" % (name, timed(lambda: eng._call("sylow_hip_evm_ecpairing_batch", d_in.ptr, d_off.ptr, nj, 2 * nj, d_res.ptr, d_st.ptr))))
assert d_res.download().all() and not d_st.download().any()
