"""Byte-level ecPairing (sylow_hip_evm_ecpairing_batch, examples/reth_bn128.rs:156-217) against the number of jobs: k pairs per job, jobs that
multiply to one.  usage: time_ecpairing_sizes.py [k]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import sylow_amd
from bench import SEED, make_points
eng = sylow_amd.Engine(0)
k = int(sys.argv[1]) if len(sys.argv) > 1 else 2
njmax = 1 << 16
npts = njmax * k // 2
p, q, ka, kb = make_points(eng, npts, SEED + 5)
ny = eng.empty((4, npts))
eng._call("sylow_hip_fp_neg_batch", p.ptr + 4 * npts * 8, ny.ptr, npts)
pneg = eng.empty((8, npts)).upload(np.concatenate([p.download()[:4], ny.download()], axis=0))
b1, b1n, b2 = eng.empty((npts * 64,), np.uint8), eng.empty((npts * 64,), np.uint8), eng.empty((npts * 128,), np.uint8)
eng._call("sylow_hip_g1_to_be_bytes_batch", p.ptr, None, b1.ptr, npts)
eng._call("sylow_hip_g1_to_be_bytes_batch", pneg.ptr, None, b1n.ptr, npts)
eng._call("sylow_hip_g2_to_be_bytes_batch", q.ptr, None, b2.ptr, npts)
g1b, g1nb, g2b = (x.download().reshape(npts, -1) for x in (b1, b1n, b2))
pos, neg = np.concatenate([g1b, g2b], axis=1), np.concatenate([g1nb, g2b], axis=1)
half = k // 2
for nj in (1, 4, 16, 64, 256, 1024, 4096, 16384, 65536):
    rows = []
    for j in range(half):
        rows += [pos[j * njmax:j * njmax + nj], neg[j * njmax:j * njmax + nj]]
    jobs = np.ascontiguousarray(np.concatenate(rows, axis=1))
    d_in, d_off = eng.to_device(jobs.reshape(-1)), eng.to_device(np.arange(nj + 1, dtype=np.uint64) * np.uint64(k))
    d_res, d_st = eng.empty((nj,), np.uint8), eng.empty((nj,), np.uint8)
    eng._call("sylow_hip_evm_ecpairing_batch", d_in.ptr, d_off.ptr, nj, k * nj, d_res.ptr, d_st.ptr); eng.sync()
    best = 1e9
    for _ in range(5):
        t0 = time.perf_counter()
        eng._call("sylow_hip_evm_ecpairing_batch", d_in.ptr, d_off.ptr, nj, k * nj, d_res.ptr, d_st.ptr); eng.sync()
        best = min(best, time.perf_counter() - t0)
    print("k = %d  jobs = %6d   %.3f ms   %.3f M jobs/s   all one: %d  status clean: %d" % (k, nj, best * 1e3, nj / best / 1e6, int(d_res.download().all()), int(not d_st.download().any())), flush=True)
