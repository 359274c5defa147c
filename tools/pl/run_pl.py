"""Parity + throughput of the lane-pair tower (tools/pl/pl_test.hip) against the oracle and the single-lane path."""
import ctypes, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import SEED, Xoshiro
from oracle import coracle as C, pyref as R
from sylow_amd.engine import Engine

lib = ctypes.CDLL(os.path.join(ROOT, "tools", "pl", "libpl_test.so"))
lib.pl_op.restype = ctypes.c_float
lib.pl_op.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_int]
lib.pl_pairing.restype = ctypes.c_float
lib.pl_pairing.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_int]
eng = Engine()
rng = Xoshiro(SEED + 777)
OPS = {"mul": 0, "sqr": 1, "inv": 2, "frob1": 3, "frob2": 4, "frob3": 5, "sparse": 6, "cycsqr": 7, "expz": 8, "final": 9}

def rand12(n): return C.to_limbs([rng.fp() for _ in range(12 * n)]).reshape(n, 48)

def run_op(pair, op, a, b=None, iters=1, reps=1):
    n = a.shape[0]
    da = eng.to_device_soa(a, 48); db = eng.to_device_soa(b, 48) if b is not None else None
    do = eng.empty((48, n))
    ms = lib.pl_op(pair, OPS[op], da.ptr, db.ptr if db is not None else None, do.ptr, n, iters, reps)
    assert ms >= 0
    return eng.from_device_soa(do), ms

def parity():
    n = 70
    a, b = rand12(n), rand12(n)
    a[0] = 0; a[1, :] = 0; a[1, 0] = 1          # zero and one
    exp = {"mul": C.fp12_op("mul", a, b), "sqr": C.fp12_op("sqr", a), "inv": C.fp12_op("inv", a),
           "frob1": C.fp12_op("frobenius", a, arg=1), "frob2": C.fp12_op("frobenius", a, arg=2), "frob3": C.fp12_op("frobenius", a, arg=3),
           "sparse": C.fp12_sparse_mul(a, b[:, :24])}
    for op, e in exp.items():
        for mode in (1, 2):
            if mode == 2 and op == "inv": continue
            got, _ = run_op(mode, op, a, b if op in ("mul", "sparse") else None)
            print(f"mode {mode} {op:7s} parity:", np.array_equal(got, e)); assert np.array_equal(got, e), op
    # cyclotomic ops need subgroup members: easy part of random values via the oracle
    cyc = C.fp12_op("mul", C.fp12_op("frobenius", C.fp12_op("mul", C.fp12_op("conj", a[2:]), C.fp12_op("inv", a[2:])), arg=2),
                    C.fp12_op("mul", C.fp12_op("conj", a[2:]), C.fp12_op("inv", a[2:])))
    ref, _ = run_op(0, "expz", cyc)
    for mode in (1, 2):
        got, _ = run_op(mode, "cycsqr", cyc)
        assert np.array_equal(got, C.fp12_op("sqr", cyc)); print(f"mode {mode} cycsqr  parity: True")
        got, _ = run_op(mode, "cycsqr", cyc, iters=40); e = cyc
        for _ in range(40): e = C.fp12_op("sqr", e)
        assert np.array_equal(got, e); print(f"mode {mode} cycsqr x40 parity: True")
        got, _ = run_op(mode, "expz", cyc)
        assert np.array_equal(got, ref); print(f"mode {mode} expz == single-lane expz: True")
        got, _ = run_op(mode, "final", a[2:])
        assert np.array_equal(got, C.final_exponentiation(a[2:])); print(f"mode {mode} final_exponentiation parity: True")
        got, _ = run_op(mode, "mul", a, b, iters=25); e = a
        for _ in range(25): e = C.fp12_op("mul", e, b)
        assert np.array_equal(got, e); print(f"mode {mode} mul x25 parity: True")
        got, _ = run_op(mode, "sqr", a, iters=25); e = a
        for _ in range(25): e = C.fp12_op("sqr", e)
        assert np.array_equal(got, e); print(f"mode {mode} sqr x25 parity: True")

def pairing_inputs(n):
    ks = C.to_limbs([rng.fp() % R.R_ORDER for _ in range(2 * n)])
    g1 = np.tile(C.to_limbs([1, 2, 1]).reshape(1, 12), (n, 1)); g2 = np.tile(C.to_limbs(list(R.G2_GEN_AFF[0]) + list(R.G2_GEN_AFF[1]) + [1, 0]).reshape(1, 24), (n, 1))
    p_aff, _ = C.g1_to_affine(C.g1_scalar_mul(g1, ks[:n])); q_aff, _ = C.g2_to_affine(C.g2_scalar_mul(g2, ks[n:]))
    return p_aff, q_aff

def pairing_parity():
    n = 33
    p, q = pairing_inputs(n)
    dp, dq = eng.to_device_soa(p, 8), eng.to_device_soa(q, 16)
    for mode in (1, 2, 3):
        dg, df = eng.empty((48, n)), eng.empty((48, n))
        assert lib.pl_pairing(mode, dp.ptr, dq.ptr, dg.ptr, df.ptr, n, 1, 0) >= 0
        f, g = eng.from_device_soa(df), eng.from_device_soa(dg)
        print(f"mode {mode} miller raw parity:", np.array_equal(f, C.miller_loop(p, q)))
        print(f"mode {mode} pairing Gt parity:", np.array_equal(g, C.final_exponentiation(C.miller_loop(p, q))))
        assert np.array_equal(f, C.miller_loop(p, q)) and np.array_equal(g, C.final_exponentiation(C.miller_loop(p, q)))

def bench():
    n = 1 << 18
    a, b = rand12(64), rand12(64)
    a = np.tile(a, (n // 64, 1)); b = np.tile(b, (n // 64, 1))
    for op, iters in (("mul", 64), ("sqr", 64), ("sparse", 64), ("cycsqr", 64)):
        r = []
        for pair in (0, 1, 2):
            _, ms = run_op(pair, op, a, b if op in ("mul", "sparse") else None, iters=iters, reps=3)
            r.append(ms)
        print(f"{op:7s} x{iters}: single-lane {r[0]:8.3f} ms   lane-pair {r[1]:8.3f} ms   lane-pair/29 {r[2]:8.3f} ms")
    n = 1 << 20
    p, q = pairing_inputs(64)
    p = np.tile(p, (n // 64, 1)); q = np.tile(q, (n // 64, 1))
    dp, dq = eng.to_device_soa(p, 8), eng.to_device_soa(q, 16)
    dg, df = eng.empty((48, n)), eng.empty((48, n))
    for pair in (0, 1, 2, 3):
        ms = lib.pl_pairing(pair, dp.ptr, dq.ptr, dg.ptr, df.ptr, n, 3, 0)
        m1 = lib.pl_pairing(pair, dp.ptr, dq.ptr, dg.ptr, df.ptr, n, 3, 1)
        m2 = lib.pl_pairing(pair, dp.ptr, dq.ptr, dg.ptr, df.ptr, n, 3, 2)
        print(f"pairing n=2^20 mode {pair}: {ms:8.2f} ms  -> {n / ms / 1e3:.3f} M pairings/s   (miller {m1:7.2f} ms, final exp {m2:7.2f} ms)")

if __name__ == "__main__":
    what = sys.argv[1:] or ["parity", "pairing", "bench"]
    if "parity" in what: parity()
    if "pairing" in what: pairing_parity()
    if "bench" in what: bench()
