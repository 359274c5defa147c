"""Raw-ctypes timing of sylow_hip_pairing_batch / sylow_hip_bls_verify_batch at 2^20 for ANY build of libsylow_hip.so (no prototype table: old builds
lack newer symbols).  usage: ab_raw.py lib.so [reps]"""
import ctypes, os, sys, time
import numpy as np
lib = ctypes.CDLL(os.path.abspath(sys.argv[1]))
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
vp, sz = ctypes.c_void_p, ctypes.c_size_t
def chk(rc, what):
    if rc != 0:
        lib.sylow_hip_last_error.restype = ctypes.c_char_p
        raise RuntimeError(f"{what}: {rc} {lib.sylow_hip_last_error()}")
chk(lib.sylow_hip_init(0), "init")
def dmalloc(nbytes):
    p = vp(); lib.sylow_hip_malloc.argtypes = [ctypes.POINTER(vp), sz]; chk(lib.sylow_hip_malloc(ctypes.byref(p), nbytes), "malloc"); return p
n = 1 << 20
ks = np.empty((4, n), dtype=np.uint64)
lib.sylow_hip_host_xoshiro_fp.argtypes = [ctypes.c_uint64, vp, sz, sz]
chk(lib.sylow_hip_host_xoshiro_fp(12345, ks.ctypes.data, n, n), "xoshiro")
dk = dmalloc(ks.nbytes)
lib.sylow_hip_memcpy_h2d.argtypes = [vp, vp, sz, vp]
chk(lib.sylow_hip_memcpy_h2d(dk, ks.ctypes.data, ks.nbytes, None), "h2d")
p, pi, q, qi, gt = dmalloc(64 * n), dmalloc(n), dmalloc(128 * n), dmalloc(n), dmalloc(384 * n)
lib.sylow_hip_g1_generator_mul_batch.argtypes = [vp, vp, vp, sz, vp]
lib.sylow_hip_g2_generator_mul_batch.argtypes = [vp, vp, vp, sz, vp]
chk(lib.sylow_hip_g1_generator_mul_batch(dk, p, pi, n, None), "g1gen")
chk(lib.sylow_hip_g2_generator_mul_batch(dk, q, qi, n, None), "g2gen")
lib.sylow_hip_pairing_batch.argtypes = [vp, vp, vp, vp, vp, sz, vp]
lib.sylow_hip_stream_sync.argtypes = [vp]
chk(lib.sylow_hip_pairing_batch(p, None, q, None, gt, n, None), "pairing"); lib.sylow_hip_stream_sync(None)
t0 = time.perf_counter()
for _ in range(reps):
    chk(lib.sylow_hip_pairing_batch(p, None, q, None, gt, n, None), "pairing")
lib.sylow_hip_stream_sync(None)
print("%-20s pairing 2^20: %.2f ms" % (os.path.basename(sys.argv[1]), (time.perf_counter() - t0) / reps * 1e3))

# BLS verify at 2^20: pk = sk * G2gen (fixed-base table), sig = sign(sk, msg), 32-byte messages
msgs = np.random.default_rng(7).integers(0, 256, size=(n, 32), dtype=np.uint8)
off = np.arange(n + 1, dtype=np.uint64) * np.uint64(32)
dm, doff = dmalloc(msgs.nbytes), dmalloc(off.nbytes)
chk(lib.sylow_hip_memcpy_h2d(dm, msgs.ctypes.data, msgs.nbytes, None), "h2d")
chk(lib.sylow_hip_memcpy_h2d(doff, off.ctypes.data, off.nbytes, None), "h2d")
sig, sigi, ok = dmalloc(64 * n), dmalloc(n), dmalloc(n)
lib.sylow_hip_bls_sign_batch.argtypes = [vp, vp, vp, vp, vp, sz, vp]
lib.sylow_hip_bls_verify_batch.argtypes = [vp, vp, vp, vp, vp, vp, vp, sz, vp]
chk(lib.sylow_hip_bls_sign_batch(dk, dm, doff, sig, sigi, n, None), "sign")
chk(lib.sylow_hip_bls_verify_batch(q, None, dm, doff, sig, None, ok, n, None), "verify"); lib.sylow_hip_stream_sync(None)
t0 = time.perf_counter()
for _ in range(reps):
    chk(lib.sylow_hip_bls_verify_batch(q, None, dm, doff, sig, None, ok, n, None), "verify")
lib.sylow_hip_stream_sync(None)
okh = np.empty(n, dtype=np.uint8)
lib.sylow_hip_memcpy_d2h.argtypes = [vp, vp, sz, vp]
chk(lib.sylow_hip_memcpy_d2h(okh.ctypes.data, ok, n, None), "d2h"); lib.sylow_hip_stream_sync(None)
print("%-20s verify  2^20: %.2f ms  all ok: %d" % (os.path.basename(sys.argv[1]), (time.perf_counter() - t0) / reps * 1e3, int(okh.all())))
