import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch, sylow_amd
eng = sylow_amd.Engine(0)
def timed(fn, reps=5):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps
n = 1 << 20
rng = np.random.default_rng(7)
for mlen in (32, 4, 100, 200):
    msgs = rng.integers(0, 256, size=(n, mlen), dtype=np.uint8)
    dm, doff = eng.to_device(msgs.reshape(-1)), eng.to_device(np.arange(n + 1, dtype=np.uint64) * np.uint64(mlen))
    u = eng.empty((8, n)); hx, hi = eng.empty((8, n)), eng.empty((n,), np.uint8)
    t1 = timed(lambda: eng._call("sylow_hip_hash_to_field_batch", dm.ptr, doff.ptr, None, 0, u.ptr, n))
    t2 = timed(lambda: eng._call("sylow_hip_hash_to_g1_batch", dm.ptr, doff.ptr, None, 0, hx.ptr, hi.ptr, n))
    print("msg_len %3d: hash_to_field %.2f ms, hash_to_g1 %.2f ms" % (mlen, t1, t2))
um = eng.empty((4, n)).upload(eng.xoshiro_fp_soa(5, n)); st = eng.empty((n,), np.uint8); xy = eng.empty((8, n))
print("svdw_map alone: %.2f ms" % timed(lambda: eng._call("sylow_hip_svdw_map_batch", um.ptr, xy.ptr, st.ptr, n)))
print("fp_inv: %.2f ms, fp_is_square: %.2f ms, fp_sqrt: %.2f ms" % (
    timed(lambda: eng._call("sylow_hip_fp_inv_batch", um.ptr, xy.ptr, n)),
    timed(lambda: eng._call("sylow_hip_fp_is_square_batch", um.ptr, st.ptr, n)),
    timed(lambda: eng._call("sylow_hip_fp_sqrt_batch", um.ptr, xy.ptr, st.ptr, n))))
