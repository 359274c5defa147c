"""hash_to_g1_batch over batch sizes (the command line, or a default ladder; each size draws its messages from its own seed) for the route the environment selects (SYLOW_HIP_WIDE_TAIL=0: one lane per message everywhere)"""
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
for n in ([int(a) for a in sys.argv[1:]] or (1, 64, 1024, 4096, 8192, 12288, 16384, 20480, 65536)):
    rng = np.random.default_rng(7 + n)
    msgs = rng.integers(0, 256, size=(n, 32), dtype=np.uint8)
    dm, doff = eng.to_device(msgs.reshape(-1)), eng.to_device(np.arange(n + 1, dtype=np.uint64) * np.uint64(32))
    hx, hi = eng.empty((8, n)), eng.empty((n,), np.uint8)
    t = timed(lambda: eng._call("sylow_hip_hash_to_g1_batch", dm.ptr, doff.ptr, None, 0, hx.ptr, hi.ptr, n))
    print("n=%6d hash_to_g1 %.3f ms  checksum %d" % (n, t, int(hx.download().sum() % 1000003)))
