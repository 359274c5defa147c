"""The multi-pair routes over job counts and job sizes: multi_pairing_batch (SoA points), the byte-level ecPairing adapter and the batch-wide
product, with a checksum of the Gt values (A/B runs of route variants must agree on it).  Round 5 used it for the two-chain slicing experiment
(profiles/r05_ab/multi_two_chains.log; the switch it toggled, SYLOW_HIP_MULTI_PIPE, was removed with the variant)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch, sylow_amd
from bench import make_points, limbs_row, G2
eng = sylow_amd.Engine(0)
def timed(fn, reps=4):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps
N = 1 << 19
p, q, ka, kb = make_points(eng, N, 5)
b1, b2 = eng.empty((N * 64,), np.uint8), eng.empty((N * 128,), np.uint8)
eng._call("sylow_hip_g1_to_be_bytes_batch", p.ptr, None, b1.ptr, N)
eng._call("sylow_hip_g2_to_be_bytes_batch", q.ptr, None, b2.ptr, N)
blob_all = np.concatenate([b1.download().reshape(N, 64), b2.download().reshape(N, 128)], axis=1)
out = []
for lgj, k in ((16, 2), (16, 4), (17, 2), (17, 4), (18, 2)):
    nj = 1 << lgj
    n = nj * k
    off = eng.to_device(np.arange(nj + 1, dtype=np.uint64) * np.uint64(k))
    pk, qk = eng.empty((8, n)).upload(np.ascontiguousarray(p.download()[:, :n])), eng.empty((16, n)).upload(np.ascontiguousarray(q.download()[:, :n]))
    gt, iso = eng.empty((48, nj)), eng.empty((nj,), np.uint8)
    t0 = timed(lambda: eng._call("sylow_hip_multi_pairing_batch", pk.ptr, None, qk.ptr, None, off.ptr, nj, n, 1, gt.ptr, iso.ptr))
    d_in = eng.to_device(blob_all[:n].reshape(-1))
    res, st = eng.empty((nj,), np.uint8), eng.empty((nj,), np.uint8)
    t1 = timed(lambda: eng._call("sylow_hip_evm_ecpairing_batch", d_in.ptr, off.ptr, nj, n, res.ptr, st.ptr))
    chk = int(gt.download()[:, :64].astype(np.uint64).sum() % 1000003)
    out.append("jobs 2^%d k=%d: multi_pairing %.2f ms (%.2f M jobs/s)   ecPairing from bytes %.2f ms (%.2f M jobs/s)   [gt checksum %d, status bad %d]"
               % (lgj, k, t0, nj / t0 / 1e3, t1, nj / t1 / 1e3, chk, int(st.download().any())))
    del pk, qk, gt, iso, d_in, res, st, off
g1o, io = eng.empty((48, 1)), eng.empty((1,), np.uint8)
t2 = timed(lambda: eng._call("sylow_hip_pairing_product_batch", p.ptr, None, q.ptr, None, N, 1, g1o.ptr, io.ptr), 3)
out.append("pairing_product 2^19 pairs: %.2f ms (%.2f M pairs/s)  [checksum %d]" % (t2, N / t2 / 1e3, int(g1o.download().astype(np.uint64).sum() % 1000003)))
print("\n".join(out))
