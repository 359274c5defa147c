"""Wire-format kernels at 2^20 rows: HIP-event time and the rate against their algorithmic bytes (in + out)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch, sylow_amd
from bench import make_points
eng = sylow_amd.Engine(0)
def timed(fn, reps=10):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
p, q, ka, kb = make_points(eng, n, 5)
g1b, g2b, fb = eng.empty((n * 64,), np.uint8), eng.empty((n * 128,), np.uint8), eng.empty((n * 32,), np.uint8)
st, inf = eng.empty((n,), np.uint8), eng.empty((n,), np.uint8)
p2, q2, k2 = eng.empty((8, n)), eng.empty((16, n)), eng.empty((4, n))
rows = [("g1_to_be_bytes", lambda: eng._call("sylow_hip_g1_to_be_bytes_batch", p.ptr, None, g1b.ptr, n), 64 + 64),
        ("g1_from_be_bytes", lambda: eng._call("sylow_hip_g1_from_be_bytes_batch", g1b.ptr, p2.ptr, inf.ptr, st.ptr, n), 64 + 66),
        ("g2_to_be_bytes", lambda: eng._call("sylow_hip_g2_to_be_bytes_batch", q.ptr, None, g2b.ptr, n), 128 + 128),
        ("g2_from_be_bytes", lambda: eng._call("sylow_hip_g2_from_be_bytes_batch", g2b.ptr, q2.ptr, inf.ptr, st.ptr, n), 128 + 130),
        ("fp_to_be_bytes", lambda: eng._call("sylow_hip_fp_to_be_bytes_batch", ka.ptr, fb.ptr, n), 64),
        ("fp_from_be_bytes", lambda: eng._call("sylow_hip_fp_from_be_bytes_batch", fb.ptr, k2.ptr, st.ptr, n), 65)]
for name, fn, bytes_per in rows:
    t = timed(fn)
    print("%-18s %8.3f ms  %7.1f GB/s algorithmic  status ok %d" % (name, t, n * bytes_per / t / 1e6, int((st.download() == 0).all())))
assert np.array_equal(p2.download(), p.download()) and np.array_equal(q2.download(), q.download()) and np.array_equal(k2.download(), ka.download())
print("round trips identical")
