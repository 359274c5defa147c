"""ctypes front-end for oracle/libsylow_oracle.so (the C restatement).  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
All arrays are numpy uint64, array-of-structs, 4 little-endian limbs per Fp (canonical values).
Shapes: Fp (n,4); Fp2 (n,8); Fp6 (n,24); Fp12 (n,48); G1 proj (n,12); G2 proj (n,24);
G1 affine (n,8)+inf(n,); G2 affine (n,16)+inf(n,).
"""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libsylow_oracle.so")
_lib = None

P_INT = 0x30644E72E131A029B85045B68181585D97816A916871CA8D3C208C16D87CFD47


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "sylow_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libsylow_oracle.so"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        _lib = ctypes.CDLL(_LIB_PATH)
        _lib.oracle_constants.restype = ctypes.c_size_t
        # warm the one-time init on this thread (so later multi-threaded timing is race-free)
        buf = np.zeros(96, dtype=np.uint64)
        _lib.oracle_constants(ctypes.c_int(2), _p(buf))
    return _lib


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def _u64(a, width=None):
    a = np.ascontiguousarray(a, dtype=np.uint64)
    if width is not None:
        a = a.reshape(-1, width)
    return a


# ---- integer <-> limb helpers -------------------------------------------------------------
def to_limbs(vals) -> np.ndarray:
    """list of python ints -> (n,4) uint64 little-endian limbs."""
    out = np.zeros((len(vals), 4), dtype=np.uint64)
    for i, v in enumerate(vals):
        for k in range(4):
            out[i, k] = (int(v) >> (64 * k)) & 0xFFFFFFFFFFFFFFFF
    return out


def from_limbs(arr) -> list:
    arr = np.asarray(arr, dtype=np.uint64).reshape(-1, 4)
    return [sum(int(arr[i, k]) << (64 * k) for k in range(4)) for i in range(arr.shape[0])]


def pack(vals, width) -> np.ndarray:
    """flat list of ints (n*width/4 per element) -> (n, width) uint64."""
    return to_limbs(vals).reshape(-1, width)


# ---- field ops -----------------------------------------------------------------------------
OPS = {"add": 0, "sub": 1, "mul": 2, "sqr": 3, "inv": 4, "neg": 5, "div": 6, "mul_xi": 7,
       "frobenius": 8, "cyclotomic_squared": 9, "conj": 10}


def _field_op(fn, width, op, a, b=None):
    a = _u64(a, width)
    out = np.empty_like(a)
    bp = None
    if b is not None:
        b = _u64(b, width)
        assert b.shape == a.shape
        bp = _p(b)
    fn(ctypes.c_int(OPS[op]), _p(a), bp, _p(out), ctypes.c_size_t(a.shape[0]))
    return out


def fp_op(op, a, b=None):
    return _field_op(lib().oracle_fp_op, 4, op, a, b)


def fp_pow(a, e):
    a, e = _u64(a, 4), _u64(e, 4)
    out = np.empty_like(a)
    lib().oracle_fp_pow(_p(a), _p(e), _p(out), ctypes.c_size_t(a.shape[0]))
    return out


def fp_sqrt(a):
    a = _u64(a, 4)
    out, ok = np.empty_like(a), np.empty(a.shape[0], dtype=np.uint8)
    lib().oracle_fp_sqrt(_p(a), _p(out), _p(ok), ctypes.c_size_t(a.shape[0]))
    return out, ok


def fp_is_square(a):
    a = _u64(a, 4)
    flags = np.empty(a.shape[0], dtype=np.uint8)
    lib().oracle_fp_is_square(_p(a), _p(flags), ctypes.c_size_t(a.shape[0]))
    return flags


def fr_op(op, a, b=None):
    return _field_op(lib().oracle_fr_op, 4, op, a, b)


def fp2_op(op, a, b=None):
    return _field_op(lib().oracle_fp2_op, 8, op, a, b)


def fp6_op(op, a, b=None):
    return _field_op(lib().oracle_fp6_op, 24, op, a, b)


def fp12_op(op, a, b=None, arg=0):
    a = _u64(a, 48)
    out = np.empty_like(a)
    bp = None
    if b is not None:
        b = _u64(b, 48)
        bp = _p(b)
    lib().oracle_fp12_op(ctypes.c_int(OPS[op]), ctypes.c_int(arg), _p(a), bp, _p(out), ctypes.c_size_t(a.shape[0]))
    return out


def fp12_sparse_mul(f, ell):
    f = _u64(f, 48)
    ell = _u64(ell, 24)
    out = np.empty_like(f)
    lib().oracle_fp12_sparse_mul(_p(f), _p(ell), _p(out), ctypes.c_size_t(f.shape[0]))
    return out


def constants(which: int) -> np.ndarray:
    buf = np.zeros(96, dtype=np.uint64)
    n = lib().oracle_constants(ctypes.c_int(which), _p(buf))
    per = 4 if which in (6, 7) else 8
    return buf[: n * per].reshape(n, per).copy()


# ---- groups ---------------------------------------------------------------------------------
def g1_scalar_mul(pts, k):
    pts, k = _u64(pts, 12), _u64(k, 4)
    out = np.empty_like(pts)
    lib().oracle_g1_scalar_mul(_p(pts), _p(k), _p(out), ctypes.c_size_t(pts.shape[0]))
    return out


def g2_scalar_mul(pts, k):
    pts, k = _u64(pts, 24), _u64(k, 4)
    out = np.empty_like(pts)
    lib().oracle_g2_scalar_mul(_p(pts), _p(k), _p(out), ctypes.c_size_t(pts.shape[0]))
    return out


def g1_add(a, b):
    a, b = _u64(a, 12), _u64(b, 12)
    out = np.empty_like(a)
    lib().oracle_g1_add(_p(a), _p(b), _p(out), ctypes.c_size_t(a.shape[0]))
    return out


def g1_double(a):
    a = _u64(a, 12)
    out = np.empty_like(a)
    lib().oracle_g1_double(_p(a), _p(out), ctypes.c_size_t(a.shape[0]))
    return out


def g2_add(a, b):
    a, b = _u64(a, 24), _u64(b, 24)
    out = np.empty_like(a)
    lib().oracle_g2_add(_p(a), _p(b), _p(out), ctypes.c_size_t(a.shape[0]))
    return out


def g2_double(a):
    a = _u64(a, 24)
    out = np.empty_like(a)
    lib().oracle_g2_double(_p(a), _p(out), ctypes.c_size_t(a.shape[0]))
    return out


def g1_to_affine(a):
    a = _u64(a, 12)
    xy = np.empty((a.shape[0], 8), dtype=np.uint64)
    inf = np.empty(a.shape[0], dtype=np.uint8)
    lib().oracle_g1_to_affine(_p(a), _p(xy), _p(inf), ctypes.c_size_t(a.shape[0]))
    return xy, inf


def g2_to_affine(a):
    a = _u64(a, 24)
    xy = np.empty((a.shape[0], 16), dtype=np.uint64)
    inf = np.empty(a.shape[0], dtype=np.uint8)
    lib().oracle_g2_to_affine(_p(a), _p(xy), _p(inf), ctypes.c_size_t(a.shape[0]))
    return xy, inf


def g2_projective_new(a):
    a = _u64(a, 24)
    st = np.empty(a.shape[0], dtype=np.uint8)
    lib().oracle_g2_projective_new(_p(a), _p(st), ctypes.c_size_t(a.shape[0]))
    return st


def g1_projective_new(a):
    a = _u64(a, 12)
    st = np.empty(a.shape[0], dtype=np.uint8)
    lib().oracle_g1_projective_new(_p(a), _p(st), ctypes.c_size_t(a.shape[0]))
    return st


def _group_binop(name, width, a, b):
    a, b = _u64(a, width), _u64(b, width)
    out = np.empty_like(a)
    getattr(lib(), name)(_p(a), _p(b), _p(out), ctypes.c_size_t(a.shape[0]))
    return out


def g1_sub(a, b): return _group_binop("oracle_g1_sub", 12, a, b)
def g2_sub(a, b): return _group_binop("oracle_g2_sub", 24, a, b)


def _group_eq(name, width, a, b):
    a, b = _u64(a, width), _u64(b, width)
    eq = np.empty(a.shape[0], dtype=np.uint8)
    getattr(lib(), name)(_p(a), _p(b), _p(eq), ctypes.c_size_t(a.shape[0]))
    return eq


def g1_ct_eq(a, b): return _group_eq("oracle_g1_ct_eq", 12, a, b)
def g2_ct_eq(a, b): return _group_eq("oracle_g2_ct_eq", 24, a, b)


def fp2_frobenius(a, e):
    a = _u64(a, 8)
    out = np.empty_like(a)
    lib().oracle_fp2_frobenius(ctypes.c_int(e), _p(a), _p(out), ctypes.c_size_t(a.shape[0]))
    return out


def fp6_frobenius(a, e):
    a = _u64(a, 24)
    out = np.empty_like(a)
    lib().oracle_fp6_frobenius(ctypes.c_int(e), _p(a), _p(out), ctypes.c_size_t(a.shape[0]))
    return out


def fp6_residue_mul(a):
    a = _u64(a, 24)
    out = np.empty_like(a)
    lib().oracle_fp6_residue_mul(_p(a), _p(out), ctypes.c_size_t(a.shape[0]))
    return out


def g2_psi(xy):
    xy = _u64(xy, 16)
    out = np.empty_like(xy)
    lib().oracle_g2_psi(_p(xy), _p(out), ctypes.c_size_t(xy.shape[0]))
    return out


# ---- pairing -----------------------------------------------------------------------------------
def g2_precompute(q_aff):
    q = _u64(q_aff, 16)
    out = np.empty((q.shape[0], 87 * 24), dtype=np.uint64)
    lib().oracle_g2_precompute(_p(q), _p(out), ctypes.c_size_t(q.shape[0]))
    return out


def miller_loop(p_aff, q_aff):
    p, q = _u64(p_aff, 8), _u64(q_aff, 16)
    out = np.empty((p.shape[0], 48), dtype=np.uint64)
    lib().oracle_miller_loop(_p(p), _p(q), _p(out), ctypes.c_size_t(p.shape[0]))
    return out


def final_exponentiation(f):
    f = _u64(f, 48)
    out = np.empty_like(f)
    lib().oracle_final_exponentiation(_p(f), _p(out), ctypes.c_size_t(f.shape[0]))
    return out


def pairing(p_proj, q_proj):
    p, q = _u64(p_proj, 12), _u64(q_proj, 24)
    out = np.empty((p.shape[0], 48), dtype=np.uint64)
    lib().oracle_pairing(_p(p), _p(q), _p(out), ctypes.c_size_t(p.shape[0]))
    return out


def bench_pairing_threads(p_proj, q_proj, threads, seconds):
    """bench.py's all-core CPU leg: `threads` POSIX threads (oracle_bench_pairing_threads, inside the C library: no Python in the loop) each
    evaluate pairing(p, q) on the first input pair until `seconds` have passed.  -> (per-thread counts, elapsed seconds)"""
    p, q = _u64(p_proj, 12), _u64(q_proj, 24)
    threads = max(1, int(threads))
    counts, sink = np.zeros(threads, dtype=np.uint64), np.zeros(threads, dtype=np.uint64)
    elapsed = ctypes.c_double(0.0)
    rc = lib().oracle_bench_pairing_threads(_p(p), _p(q), ctypes.c_int(threads), ctypes.c_double(float(seconds)), _p(counts), _p(sink), ctypes.byref(elapsed))
    if rc != 0:
        raise RuntimeError("oracle_bench_pairing_threads: a thread could not be created")
    return counts, float(elapsed.value)


def glued_pairing(p_proj, q_proj, offsets):
    p, q = _u64(p_proj, 12), _u64(q_proj, 24)
    off = np.ascontiguousarray(offsets, dtype=np.uint64)
    nj = off.shape[0] - 1
    out = np.empty((nj, 48), dtype=np.uint64)
    lib().oracle_glued_pairing(_p(p), _p(q), _p(off), _p(out), ctypes.c_size_t(nj))
    return out


def gt_pow(g, k):
    g, k = _u64(g, 48), _u64(k, 4)
    out = np.empty_like(g)
    lib().oracle_gt_pow(_p(g), _p(k), _p(out), ctypes.c_size_t(g.shape[0]))
    return out


# ---- hashing / BLS -------------------------------------------------------------------------
def keccak256(msg: bytes) -> bytes:
    out = ctypes.create_string_buffer(32)
    lib().oracle_keccak256(msg, ctypes.c_size_t(len(msg)), out)
    return out.raw


def expand_message_xmd_keccak(msg: bytes, dst: bytes, n: int) -> bytes:
    out = ctypes.create_string_buffer(n)
    ok = lib().oracle_expand_message_xmd_keccak(msg, ctypes.c_size_t(len(msg)), dst, ctypes.c_size_t(len(dst)), out, ctypes.c_size_t(n))
    assert ok
    return out.raw


def svdw_map(u):
    u = _u64(u, 4)
    out = np.empty((u.shape[0], 8), dtype=np.uint64)
    lib().oracle_svdw_map(_p(u), _p(out), ctypes.c_size_t(u.shape[0]))
    return out


def _msgs(msgs):
    off = np.zeros(len(msgs) + 1, dtype=np.uint64)
    for i, m in enumerate(msgs):
        off[i + 1] = off[i] + len(m)
    blob = b"".join(msgs) or b"\x00"
    return blob, off


def hash_to_curve(msgs, dst: bytes | None = None):
    blob, off = _msgs(msgs)
    out = np.empty((len(msgs), 12), dtype=np.uint64)
    ok = lib().oracle_hash_to_curve(blob, _p(off), dst, ctypes.c_size_t(len(dst) if dst else 0), _p(out), ctypes.c_size_t(len(msgs)))
    assert ok
    return out


def sign(sk, msgs):
    sk = _u64(sk, 4)
    blob, off = _msgs(msgs)
    out = np.empty((len(msgs), 12), dtype=np.uint64)
    ok = lib().oracle_sign(_p(sk), blob, _p(off), _p(out), ctypes.c_size_t(len(msgs)))
    assert ok
    return out


def verify(pk_proj, msgs, sig_proj):
    pk, sig = _u64(pk_proj, 24), _u64(sig_proj, 12)
    blob, off = _msgs(msgs)
    ok_flags = np.empty(len(msgs), dtype=np.uint8)
    ok = lib().oracle_verify(_p(pk), blob, _p(off), _p(sig), _p(ok_flags), ctypes.c_size_t(len(msgs)))
    assert ok
    return ok_flags
