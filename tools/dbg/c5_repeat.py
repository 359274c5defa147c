"""Repeat the byte-level ecPairing configuration (2^16 jobs, k pairs) and count calls whose result pattern is wrong."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, sylow_amd
from bench import make_points, SEED
eng = sylow_amd.Engine(0)
n3 = 1 << 18
p3, q3, ka, kb = make_points(eng, n3, SEED + 3)
nj = 1 << 16
npts = 2 * nj
ny = eng.empty((4, n3))
eng._call("sylow_hip_fp_neg_batch", p3.ptr + 4 * n3 * 8, ny.ptr, n3)
pneg = eng.empty((8, n3)).upload(np.concatenate([p3.download()[:4], ny.download()], axis=0))
b1, b1n, b2 = eng.empty((n3 * 64,), np.uint8), eng.empty((n3 * 64,), np.uint8), eng.empty((n3 * 128,), np.uint8)
eng._call("sylow_hip_g1_to_be_bytes_batch", p3.ptr, None, b1.ptr, n3)
eng._call("sylow_hip_g1_to_be_bytes_batch", pneg.ptr, None, b1n.ptr, n3)
eng._call("sylow_hip_g2_to_be_bytes_batch", q3.ptr, None, b2.ptr, n3)
g1b, g1nb, g2b = (x.download().reshape(n3, -1)[:npts] for x in (b1, b1n, b2))
pos, neg = np.concatenate([g1b, g2b], axis=1), np.concatenate([g1nb, g2b], axis=1)
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
for k in (2, 4):
    jobs = np.concatenate([pos[:nj], neg[:nj]], axis=1) if k == 2 else np.concatenate([pos[0:2 * nj:2], neg[0:2 * nj:2], pos[1:2 * nj:2], neg[1:2 * nj:2]], axis=1)
    d_in = eng.to_device(np.ascontiguousarray(jobs).reshape(-1))
    d_off = eng.to_device(np.arange(nj + 1, dtype=np.uint64) * np.uint64(k))
    d_res, d_st = eng.empty((nj,), np.uint8), eng.empty((nj,), np.uint8)
    bad_calls, bad_jobs = 0, []
    for r in range(reps):
        eng._call("sylow_hip_evm_ecpairing_batch", d_in.ptr, d_off.ptr, nj, k * nj, d_res.ptr, d_st.ptr)
        res, st = d_res.download(), d_st.download()
        if not res.all() or st.any():
            bad_calls += 1
            bad_jobs.append((int((res == 0).sum()), int((st != 0).sum()), np.flatnonzero(res == 0)[:8].tolist()))
    print("k=%d: %d of %d calls wrong %s" % (k, bad_calls, reps, bad_jobs[:4]))
