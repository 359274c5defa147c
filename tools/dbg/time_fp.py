import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch, sylow_amd
eng = sylow_amd.Engine(0)
def timed(fn, reps):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps
out = []
for lg, reps in ((20, 200), (24, 20), (26, 6)):
    m = 1 << lg
    a = eng.empty((4, m)).upload(eng.xoshiro_fp_soa(7, m)); b = eng.empty((4, m)).upload(eng.xoshiro_fp_soa(8, m)); o = eng.empty((4, m))
    for name in ("mul", "add"):
        t = timed(lambda: eng._call(f"sylow_hip_fp_{name}_batch", a.ptr, b.ptr, o.ptr, m), reps)
        out.append("2^%d %s %.2f TB/s" % (lg, name, 96 * m / t / 1e9))
    del a, b, o
print(" | ".join(out))
