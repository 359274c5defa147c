"""Does MIXING the two phases of a pairing on one SIMD help?  A one-round launch (2^16 pairings: every wavefront starts together, so the two
wavefronts of a SIMD are in the same phase all the way) runs ~10 % below the steady-state rate of a 2^20 launch, where the co-resident
wavefronts are in different phases.  Test: Miller loops and final exponentiations of HALF a round each, concurrently on two streams (every
CU then holds a block of each kernel, if the dispatcher interleaves them), against the same kernels alone at full occupancy."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch, sylow_amd
from bench import make_points
n = 1 << 16
h = n // 2
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
e0 = sylow_amd.Engine(0)
e1, e2 = sylow_amd.Engine(0, stream=s1.cuda_stream), sylow_amd.Engine(0, stream=s2.cuda_stream)
p, q, _, _ = make_points(e0, n, 99)
f, g = e0.empty((48, n)), e0.empty((48, n))
ph, qh = e0.empty((8, h)).upload(np.ascontiguousarray(p.download()[:, :h])), e0.empty((16, h)).upload(np.ascontiguousarray(q.download()[:, :h]))
fh, gh, f2 = e0.empty((48, h)), e0.empty((48, h)), e0.empty((48, h))
e0._call("sylow_hip_miller_loop_batch", p.ptr, q.ptr, f.ptr, n)
e0._call("sylow_hip_miller_loop_batch", ph.ptr, qh.ptr, f2.ptr, h)
torch.cuda.synchronize()
def timed(fn, reps=6):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps
tm = timed(lambda: e0._call("sylow_hip_miller_loop_batch", p.ptr, q.ptr, f.ptr, n))
tf = timed(lambda: e0._call("sylow_hip_final_exp_batch", f.ptr, g.ptr, n))
tp = timed(lambda: e0._call("sylow_hip_pairing_batch", p.ptr, None, q.ptr, None, g.ptr, n))
tmh = timed(lambda: e0._call("sylow_hip_miller_loop_batch", ph.ptr, qh.ptr, fh.ptr, h))
tfh = timed(lambda: e0._call("sylow_hip_final_exp_batch", f2.ptr, gh.ptr, h))
def both():
    e1._call("sylow_hip_miller_loop_batch", ph.ptr, qh.ptr, fh.ptr, h)
    e2._call("sylow_hip_final_exp_batch", f2.ptr, gh.ptr, h)
tb = timed(both)
print("n = 2^16 (one round of the resident wavefronts):  Miller %.3f ms, final exp %.3f ms, sum %.3f; fused k_pairing %.3f ms" % (tm, tf, tm + tf, tp))
print("half rounds alone (one wavefront per SIMD):        Miller %.3f ms, final exp %.3f ms" % (tmh, tfh))
print("half a round of each, concurrently on two streams: %.3f ms   (same work as (Miller + final exp) / 2 = %.3f ms at full single-phase occupancy)" % (tb, (tm + tf) / 2))
print("mixing gain: %.1f %%" % (100 * ((tm + tf) / 2 / tb - 1)))
