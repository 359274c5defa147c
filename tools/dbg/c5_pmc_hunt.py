"""Hunt for the rare wrong ecPairing result seen twice inside `rocprofv3 --pmc` passes of tools/prof_configs.py (docs/DESIGN_LOG.md): the same
memory-pool history as the profile driver (G1 / G2 scalar multiplications with leased window tables first), then R rounds of
  3 x sylow_hip_evm_ecpairing_batch back to back (2^16 jobs of 2 pairs, every result must be 1, every status 0)
  1 x sylow_hip_multi_pairing_batch on the same points with Gt out (must be bit-identical to the first round's Gt)
Every deviation is printed in full: which jobs (run lengths), which words of Gt, whether a repeat of the call reproduces it.
    python3 tools/dbg/c5_pmc_hunt.py [rounds]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, sylow_amd
from bench import make_points, SEED

def runs(idx):
    idx = np.asarray(idx)
    if idx.size == 0: return []
    cut = np.flatnonzero(np.diff(idx) != 1)
    starts = np.concatenate([[0], cut + 1]); ends = np.concatenate([cut, [idx.size - 1]])
    return [(int(idx[a]), int(idx[b])) for a, b in zip(starts, ends)]

eng = sylow_amd.Engine(0)
R = int(sys.argv[1]) if len(sys.argv) > 1 else 100
n = 1 << 20
p, q, ka, kb = make_points(eng, n, SEED + 3)
o1, o1i = eng.empty((8, n)), eng.empty((n,), np.uint8)
eng._call("sylow_hip_g1_scalar_mul_batch", p.ptr, None, ka.ptr, o1.ptr, o1i.ptr, n)
o2, o2i = eng.empty((16, n)), eng.empty((n,), np.uint8)
eng._call("sylow_hip_g2_scalar_mul_subgroup_batch", q.ptr, None, ka.ptr, o2.ptr, o2i.ptr, n)
eng._call("sylow_hip_g2_scalar_mul_batch", q.ptr, None, ka.ptr, o2.ptr, o2i.ptr, n)
del o1, o1i, o2, o2i
n3 = 1 << 18
p3 = eng.empty((8, n3)).upload(np.ascontiguousarray(p.download()[:, :n3]))
q3 = eng.empty((16, n3)).upload(np.ascontiguousarray(q.download()[:, :n3]))
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
jobs = np.concatenate([pos[:nj], neg[:nj]], axis=1)
d_in = eng.to_device(np.ascontiguousarray(jobs).reshape(-1))
d_off = eng.to_device(np.arange(nj + 1, dtype=np.uint64) * np.uint64(2))
d_res, d_st = eng.empty((nj,), np.uint8), eng.empty((nj,), np.uint8)
gtj, iso = eng.empty((48, nj)), eng.empty((nj,), np.uint8)
ec = lambda: eng._call("sylow_hip_evm_ecpairing_batch", d_in.ptr, d_off.ptr, nj, 2 * nj, d_res.ptr, d_st.ptr)
mp = lambda: eng._call("sylow_hip_multi_pairing_batch", p3.ptr, None, q3.ptr, None, d_off.ptr, nj, 2 * nj, 1, gtj.ptr, iso.ptr)
ec(); eng.sync()
assert d_res.download().all() and not d_st.download().any(), "first call already wrong"
mp(); eng.sync()
gt0, iso0 = gtj.download().copy(), iso.download().copy()
bad_ec = bad_mp = 0
t0 = time.time()
for r in range(R):
    ec(); ec(); ec(); eng.sync()
    res, st = d_res.download(), d_st.download()
    if not res.all() or st.any():
        bad_ec += 1
        z = np.flatnonzero(res == 0)
        print("round %d: ecPairing wrong: %d zero results, runs %s; %d nonzero statuses %s" % (r, z.size, runs(z)[:12], int((st != 0).sum()), runs(np.flatnonzero(st != 0))[:6]), flush=True)
        again = []
        for _ in range(3):
            ec(); eng.sync()
            again.append(int((d_res.download() == 0).sum()))
        print("   repeats of the single call: zero results %s" % again, flush=True)
    mp(); eng.sync()
    g, i1 = gtj.download(), iso.download()
    if not np.array_equal(g, gt0) or not np.array_equal(i1, iso0):
        bad_mp += 1
        diff = (g != gt0)
        jobs_bad = np.flatnonzero(diff.any(axis=0)); words_bad = np.flatnonzero(diff.any(axis=1))
        print("round %d: multi_pairing Gt differs: jobs %s (count %d), words %s, flags differing %d" % (r, runs(jobs_bad)[:12], jobs_bad.size, words_bad.tolist(), int((i1 != iso0).sum())), flush=True)
        j = int(jobs_bad[0])
        print("   job %d: got %s\n           want %s" % (j, [hex(int(v)) for v in g[:6, j]], [hex(int(v)) for v in gt0[:6, j]]), flush=True)
print("rounds %d: ecPairing wrong %d, multi_pairing wrong %d, %.1f s" % (R, bad_ec, bad_mp, time.time() - t0), flush=True)
