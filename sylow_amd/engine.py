"""Batch engine: device buffers + one method per C-ABI entry point, numpy in / numpy out.

Host-side arrays are array-of-structs uint64 [n, W] (W as in include/sylow_hip.h); the engine
transposes to the struct-of-arrays [W, n] device layout.  Device-resident use (bench.py) goes
through `DeviceArray` handles directly, so the timed region contains no host traffic.
"""
from __future__ import annotations

import ctypes

import numpy as np

from . import _lib, _shapes


class DeviceArray:
    """A hipMalloc'ed buffer holding a numpy-shaped array (row-major)."""

    def __init__(self, engine: "Engine", shape, dtype):
        self.engine = engine
        self.shape = tuple(int(s) for s in shape)
        self.dtype = np.dtype(dtype)
        self.nbytes = int(np.prod(self.shape, dtype=np.int64)) * self.dtype.itemsize
        p = ctypes.c_void_p()
        _lib.check(engine.lib.sylow_hip_malloc(ctypes.byref(p), self.nbytes), "malloc")
        self.ptr = p.value
        engine._live[self.ptr] = self.nbytes              # what _shapes.check_call compares against the header's @shape lines

    def free(self):
        if self.ptr:
            self.engine._live.pop(self.ptr, None)
            self.engine.lib.sylow_hip_free(self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass

    def upload(self, arr: np.ndarray):
        arr = np.ascontiguousarray(arr, dtype=self.dtype)
        assert arr.nbytes == self.nbytes, (arr.shape, self.shape)
        _lib.check(self.engine.lib.sylow_hip_memcpy_h2d(self.ptr, arr.ctypes.data, self.nbytes, self.engine.stream), "h2d")
        _lib.check(self.engine.lib.sylow_hip_stream_sync(self.engine.stream), "sync")
        return self

    def download(self) -> np.ndarray:
        out = np.empty(self.shape, dtype=self.dtype)
        _lib.check(self.engine.lib.sylow_hip_memcpy_d2h(out.ctypes.data, self.ptr, self.nbytes, self.engine.stream), "d2h")
        _lib.check(self.engine.lib.sylow_hip_stream_sync(self.engine.stream), "sync")
        return out


def _aos(a, width):
    a = np.ascontiguousarray(a, dtype=np.uint64)
    if a.ndim == 1:
        a = a.reshape(-1, width)
    assert a.ndim == 2 and a.shape[1] == width, (a.shape, width)
    return a


class Engine:
    """One engine per process / GPU (one process per GPU, as torch.distributed launches them)."""

    def __init__(self, device: int = 0, stream: int | None = None):
        self.lib = _lib.load()
        _lib.check(self.lib.sylow_hip_init(device), "sylow_hip_init")
        self.device = device
        self.stream = stream  # raw hipStream_t as int, or None for the default stream
        self._live = {}       # base pointer -> bytes of every live DeviceArray (checked against the header's @shape lines in _call)

    # ---- buffers ---------------------------------------------------------------------------
    def empty(self, shape, dtype=np.uint64) -> DeviceArray:
        return DeviceArray(self, shape, dtype)

    def to_device_soa(self, aos: np.ndarray, width: int) -> DeviceArray:
        a = _aos(aos, width)
        return self.empty((width, a.shape[0])).upload(np.ascontiguousarray(a.T))

    def to_device(self, arr: np.ndarray, dtype=None) -> DeviceArray:
        arr = np.ascontiguousarray(arr, dtype=dtype)
        return self.empty(arr.shape, arr.dtype).upload(arr)

    def from_device_soa(self, d: DeviceArray) -> np.ndarray:
        return np.ascontiguousarray(d.download().T)

    def sync(self):
        _lib.check(self.lib.sylow_hip_stream_sync(self.stream), "sync")

    def pinned_empty(self, shape, dtype=np.uint64) -> np.ndarray:
        """A numpy array over page-locked host memory (sylow_hip_host_malloc): every copy of the host pipeline is then asynchronous.
        The memory is released when the array (and every view of it) is garbage-collected."""
        dtype = np.dtype(dtype)
        nbytes = max(1, int(np.prod(shape, dtype=np.int64)) * dtype.itemsize)
        p = ctypes.c_void_p()
        _lib.check(self.lib.sylow_hip_host_malloc(ctypes.byref(p), nbytes), "host_malloc")
        lib, addr = self.lib, p.value

        class _Owner:
            def __del__(self_inner):
                try:
                    lib.sylow_hip_host_free(addr)
                except Exception:
                    pass
        buf = (ctypes.c_uint8 * nbytes).from_address(addr)
        buf._owner = _Owner()
        return np.frombuffer(buf, dtype=dtype, count=int(np.prod(shape, dtype=np.int64))).reshape(shape)

    def xoshiro_fp_soa(self, seed: int, n: int) -> np.ndarray:
        """n values < p from the SplitMix64-seeded xoshiro256** stream, host array in SoA layout [4, n]."""
        out = np.empty((4, n), dtype=np.uint64)
        _lib.check(self.lib.sylow_hip_host_xoshiro_fp(seed & ((1 << 64) - 1), out.ctypes.data, n, n), "xoshiro")
        return out

    def _flags(self, inf, n):
        if inf is None:
            return None
        f = np.ascontiguousarray(inf, dtype=np.uint8).reshape(n)
        return self.to_device(f)

    @staticmethod
    def _ptr(d):
        return None if d is None else d.ptr

    def _call(self, name, *args):
        # launches go to the calling thread's current device: re-assert ours (other code in the process may have switched it)
        try:
            _shapes.check_call(name, args, self._live)       # every buffer at least as large as include/sylow_hip.h's @shape says
        except ValueError as e:
            raise _lib.SylowHipError(str(e)) from None
        _lib.check(self.lib.sylow_hip_set_device(self.device), "sylow_hip_set_device")
        _lib.check(getattr(self.lib, name)(*args, self.stream), name)

    def trim(self, keep_bytes: int = 0):
        """Free this device's idle scratch blocks above `keep_bytes` whose last user has completed (sylow_hip_trim)."""
        _lib.check(self.lib.sylow_hip_set_device(self.device), "sylow_hip_set_device")
        _lib.check(self.lib.sylow_hip_trim(keep_bytes), "sylow_hip_trim")

    def set_option(self, name: str, value: int = -1):
        """A route selector / threshold of the library (sylow_hip_set_option; names = _lib.OPTIONS, value < 0 = the default).  Process-wide."""
        _lib.check(self.lib.sylow_hip_set_option(_lib.OPTIONS[name], value), "sylow_hip_set_option")

    def get_option(self, name: str) -> int:
        v = ctypes.c_int64(0)
        _lib.check(self.lib.sylow_hip_get_option(_lib.OPTIONS[name], ctypes.byref(v)), "sylow_hip_get_option")
        return int(v.value)

    def wall_clock_khz(self) -> int:
        v = ctypes.c_int32(0)
        _lib.check(self.lib.sylow_hip_wall_clock_khz(ctypes.byref(v)), "sylow_hip_wall_clock_khz")
        return int(v.value)

    def clock_probe(self, acc=None):
        """Switch the live clock probe of the metric's kernels on (acc: a zeroed DeviceArray of 256 uint64) or off (None)."""
        if acc is not None and acc.nbytes < 256 * 8:
            raise ValueError("clock_probe: the accumulator holds fewer than 256 uint64 words")
        _lib.check(self.lib.sylow_hip_clock_probe(acc.ptr if acc is not None else None), "sylow_hip_clock_probe")

    @staticmethod
    def clock_probe_summary(words, khz):
        """(sustained MHz, shader-clock ticks, wavefronts, longest wavefront in ms) from the 256 accumulator words and the constant rate."""
        w = np.asarray(words, dtype=np.uint64).reshape(64, 4)
        clk, wall, waves = int(w[:, 0].sum()), int(w[:, 1].sum()), int(w[:, 2].sum())
        mhz = clk / wall * khz / 1e3 if wall else None
        return mhz, clk, waves, (int(w[:, 3].max()) / khz if khz else None)

    def set_scratch_limit(self, nbytes: int = 0):
        """Upper bound for the multi-pair routes' line tables (sylow_hip_set_scratch_limit; 0 = the default of 12 GB).  Process-wide."""
        _lib.check(self.lib.sylow_hip_set_scratch_limit(nbytes), "sylow_hip_set_scratch_limit")

    def shutdown(self):
        """Free the library's scratch blocks and generator tables on every device (it stays usable)."""
        _lib.check(self.lib.sylow_hip_shutdown(), "sylow_hip_shutdown")

    # ---- field ops (numpy AoS in/out) -----------------------------------------------------
    def _binop(self, name, width, a, b):
        a, b = _aos(a, width), _aos(b, width)
        n = a.shape[0]
        da, db = self.to_device_soa(a, width), self.to_device_soa(b, width)
        do = self.empty((width, n))
        self._call(name, da.ptr, db.ptr, do.ptr, n)
        return self.from_device_soa(do)

    def _unop(self, name, width, a, *extra):
        a = _aos(a, width)
        n = a.shape[0]
        da = self.to_device_soa(a, width)
        do = self.empty((width, n))
        self._call(name, da.ptr, *extra, do.ptr, n)
        return self.from_device_soa(do)

    def fp_add(self, a, b): return self._binop("sylow_hip_fp_add_batch", 4, a, b)
    def fp_sub(self, a, b): return self._binop("sylow_hip_fp_sub_batch", 4, a, b)
    def fp_mul(self, a, b): return self._binop("sylow_hip_fp_mul_batch", 4, a, b)
    def fp_sqr(self, a): return self._unop("sylow_hip_fp_sqr_batch", 4, a)
    def fp_neg(self, a): return self._unop("sylow_hip_fp_neg_batch", 4, a)
    def fp_inv(self, a): return self._unop("sylow_hip_fp_inv_batch", 4, a)
    def fp_pow(self, a, e): return self._binop("sylow_hip_fp_pow_batch", 4, a, e)

    def fp_sqrt(self, a):
        a = _aos(a, 4)
        n = a.shape[0]
        da, do, dk = self.to_device_soa(a, 4), self.empty((4, n)), self.empty((n,), np.uint8)
        self._call("sylow_hip_fp_sqrt_batch", da.ptr, do.ptr, dk.ptr, n)
        return self.from_device_soa(do), dk.download()

    def fp_is_square(self, a):
        a = _aos(a, 4)
        n = a.shape[0]
        da, dk = self.to_device_soa(a, 4), self.empty((n,), np.uint8)
        self._call("sylow_hip_fp_is_square_batch", da.ptr, dk.ptr, n)
        return dk.download()

    def fr_add(self, a, b): return self._binop("sylow_hip_fr_add_batch", 4, a, b)
    def fr_sub(self, a, b): return self._binop("sylow_hip_fr_sub_batch", 4, a, b)
    def fr_mul(self, a, b): return self._binop("sylow_hip_fr_mul_batch", 4, a, b)
    def fr_sqr(self, a): return self._unop("sylow_hip_fr_sqr_batch", 4, a)
    def fr_neg(self, a): return self._unop("sylow_hip_fr_neg_batch", 4, a)
    def fr_inv(self, a): return self._unop("sylow_hip_fr_inv_batch", 4, a)
    def fp2_mul(self, a, b): return self._binop("sylow_hip_fp2_mul_batch", 8, a, b)
    def fp2_sqr(self, a): return self._unop("sylow_hip_fp2_sqr_batch", 8, a)
    def fp2_inv(self, a): return self._unop("sylow_hip_fp2_inv_batch", 8, a)
    def fp6_mul(self, a, b): return self._binop("sylow_hip_fp6_mul_batch", 24, a, b)
    def fp6_inv(self, a): return self._unop("sylow_hip_fp6_inv_batch", 24, a)
    def fp2_residue_mul(self, a): return self._unop("sylow_hip_fp2_residue_mul_batch", 8, a)
    def fp6_sqr(self, a): return self._unop("sylow_hip_fp6_sqr_batch", 24, a)
    def fp6_residue_mul(self, a): return self._unop("sylow_hip_fp6_residue_mul_batch", 24, a)

    def _frobenius(self, name, width, a, e):
        a = _aos(a, width)
        n = a.shape[0]
        da, do = self.to_device_soa(a, width), self.empty((width, n))
        self._call(name, da.ptr, int(e), do.ptr, n)
        return self.from_device_soa(do)

    def fp2_frobenius(self, a, e): return self._frobenius("sylow_hip_fp2_frobenius_batch", 8, a, e)
    def fp6_frobenius(self, a, e): return self._frobenius("sylow_hip_fp6_frobenius_batch", 24, a, e)
    def fp12_mul(self, a, b): return self._binop("sylow_hip_fp12_mul_batch", 48, a, b)
    def fp12_sqr(self, a): return self._unop("sylow_hip_fp12_sqr_batch", 48, a)
    def fp12_inv(self, a): return self._unop("sylow_hip_fp12_inv_batch", 48, a)
    def fp12_frobenius(self, a, e): return self._unop("sylow_hip_fp12_frobenius_batch", 48, a, int(e))

    def f29_hook(self, op, a, b):
        a, b = _aos(a, 4), _aos(b, 4)
        n = a.shape[0]
        da, db = self.to_device_soa(a, 4), self.to_device_soa(b, 4)
        do = self.empty((4, n))
        self._call("sylow_hip_f29_hook_batch", int(op), da.ptr, db.ptr, do.ptr, n)
        return self.from_device_soa(do)

    def fp12_hook(self, op, a, b=None):
        a = _aos(a, 48)
        n = a.shape[0]
        da = self.to_device_soa(a, 48)
        db = self.to_device_soa(_aos(b, 48), 48) if b is not None else None
        do = self.empty((48, n))
        self._call("sylow_hip_fp12_hook_batch", int(op), da.ptr, self._ptr(db), do.ptr, n)
        return self.from_device_soa(do)

    def fp12_cyclotomic_sqr(self, a):
        return self._unop("sylow_hip_fp12_cyclotomic_sqr_batch", 48, a)

    def fp12_sparse_mul(self, f, ell):
        f, ell = _aos(f, 48), _aos(ell, 24)
        n = f.shape[0]
        df, dl = self.to_device_soa(f, 48), self.to_device_soa(ell, 24)
        do = self.empty((48, n))
        self._call("sylow_hip_fp12_sparse_mul_batch", df.ptr, dl.ptr, do.ptr, n)
        return self.from_device_soa(do)

    # Fp / Fr ::from_be_bytes / to_be_bytes (fp.rs:686-737, 746-778): (value mod modulus, status) -- both halves of the CtOption
    def _fe_from_bytes(self, name, blobs):
        n = len(blobs)
        assert all(len(b) == 32 for b in blobs)
        din = self.to_device(np.frombuffer(b"".join(blobs) or b"\x00", dtype=np.uint8))
        do, dst = self.empty((4, max(n, 1))), self.empty((max(n, 1),), np.uint8)
        self._call(name, din.ptr, do.ptr, dst.ptr, n)
        return self.from_device_soa(do)[:n], dst.download()[:n]

    def _fe_to_bytes(self, name, a):
        a = _aos(a, 4)
        n = a.shape[0]
        da, do = self.to_device_soa(a, 4), self.empty((n * 32,), np.uint8)
        self._call(name, da.ptr, do.ptr, n)
        raw = do.download().tobytes()
        return [raw[32 * i:32 * (i + 1)] for i in range(n)]

    def fp_from_be_bytes(self, blobs): return self._fe_from_bytes("sylow_hip_fp_from_be_bytes_batch", blobs)
    def fr_from_be_bytes(self, blobs): return self._fe_from_bytes("sylow_hip_fr_from_be_bytes_batch", blobs)
    def fp_to_be_bytes(self, a): return self._fe_to_bytes("sylow_hip_fp_to_be_bytes_batch", a)
    def fr_to_be_bytes(self, a): return self._fe_to_bytes("sylow_hip_fr_to_be_bytes_batch", a)

    # ---- groups ----------------------------------------------------------------------------
    def _scalar_mul(self, name, width, p_xy, p_inf, k):
        p_xy, k = _aos(p_xy, width), _aos(k, 4)
        n = p_xy.shape[0]
        dp, dk, di = self.to_device_soa(p_xy, width), self.to_device_soa(k, 4), self._flags(p_inf, n)
        do, doi = self.empty((width, n)), self.empty((n,), np.uint8)
        self._call(name, dp.ptr, self._ptr(di), dk.ptr, do.ptr, doi.ptr, n)
        return self.from_device_soa(do), doi.download()

    def g1_scalar_mul(self, p_xy, k, p_inf=None): return self._scalar_mul("sylow_hip_g1_scalar_mul_batch", 8, p_xy, p_inf, k)
    def g2_scalar_mul(self, p_xy, k, p_inf=None, subgroup=False):
        """k * P on the twist; subgroup=True: P is known to be in the r-torsion (4-way endomorphism split, ~1.8x faster)."""
        return self._scalar_mul("sylow_hip_g2_scalar_mul_subgroup_batch" if subgroup else "sylow_hip_g2_scalar_mul_batch", 16, p_xy, p_inf, k)

    def g2_generator_mul(self, k):
        """G2gen * k_i (the keygen shape) through the device's fixed-base table."""
        return self._generator_mul("sylow_hip_g2_generator_mul_batch", 16, k)

    def g1_generator_mul(self, k):
        """G1gen * k_i through the device's fixed-base table."""
        return self._generator_mul("sylow_hip_g1_generator_mul_batch", 8, k)

    def _generator_mul(self, name, width, k):
        k = _aos(k, 4)
        n = k.shape[0]
        dk = self.to_device_soa(k, 4)
        do, doi = self.empty((width, n)), self.empty((n,), np.uint8)
        self._call(name, dk.ptr, do.ptr, doi.ptr, n)
        return self.from_device_soa(do), doi.download()

    def g1_add(self, a_xy, b_xy, a_inf=None, b_inf=None):
        a_xy, b_xy = _aos(a_xy, 8), _aos(b_xy, 8)
        n = a_xy.shape[0]
        da, db = self.to_device_soa(a_xy, 8), self.to_device_soa(b_xy, 8)
        dai, dbi = self._flags(a_inf, n), self._flags(b_inf, n)
        do, doi = self.empty((8, n)), self.empty((n,), np.uint8)
        self._call("sylow_hip_g1_add_batch", da.ptr, self._ptr(dai), db.ptr, self._ptr(dbi), do.ptr, doi.ptr, n)
        return self.from_device_soa(do), doi.download()

    def g1_lincomb(self, p_xy, k, n_jobs, n_terms, p_inf=None):
        """sum_i k[j,i] * P[j,i]; inputs term-major: row i*n_jobs + j is term i of job j."""
        p_xy, k = _aos(p_xy, 8), _aos(k, 4)
        n = n_jobs * n_terms
        assert p_xy.shape[0] == n and k.shape[0] == n
        dp = self.to_device_soa(p_xy, 8) if n else None
        dk = self.to_device_soa(k, 4) if n else None
        di = self._flags(p_inf, n) if n else None
        do, doi = self.empty((8, n_jobs)), self.empty((n_jobs,), np.uint8)
        self._call("sylow_hip_g1_lincomb_batch", self._ptr(dp), self._ptr(di), self._ptr(dk), do.ptr, doi.ptr, n_jobs, n_terms)
        return self.from_device_soa(do), doi.download()

    def g2_add(self, a_xy, b_xy, a_inf=None, b_inf=None):
        a_xy, b_xy = _aos(a_xy, 16), _aos(b_xy, 16)
        n = a_xy.shape[0]
        da, db = self.to_device_soa(a_xy, 16), self.to_device_soa(b_xy, 16)
        dai, dbi = self._flags(a_inf, n), self._flags(b_inf, n)
        do, doi = self.empty((16, n)), self.empty((n,), np.uint8)
        self._call("sylow_hip_g2_add_batch", da.ptr, self._ptr(dai), db.ptr, self._ptr(dbi), do.ptr, doi.ptr, n)
        return self.from_device_soa(do), doi.download()

    def _group_binop(self, name, width, a_xy, b_xy, a_inf, b_inf):
        a_xy, b_xy = _aos(a_xy, width), _aos(b_xy, width)
        n = a_xy.shape[0]
        da, db = self.to_device_soa(a_xy, width), self.to_device_soa(b_xy, width)
        dai, dbi = self._flags(a_inf, n), self._flags(b_inf, n)
        do, doi = self.empty((width, n)), self.empty((n,), np.uint8)
        self._call(name, da.ptr, self._ptr(dai), db.ptr, self._ptr(dbi), do.ptr, doi.ptr, n)
        return self.from_device_soa(do), doi.download()

    def g1_sub(self, a_xy, b_xy, a_inf=None, b_inf=None): return self._group_binop("sylow_hip_g1_sub_batch", 8, a_xy, b_xy, a_inf, b_inf)
    def g2_sub(self, a_xy, b_xy, a_inf=None, b_inf=None): return self._group_binop("sylow_hip_g2_sub_batch", 16, a_xy, b_xy, a_inf, b_inf)

    def _projective_new(self, name, width, p_xyz):
        p_xyz = _aos(p_xyz, width)
        n = p_xyz.shape[0]
        dp, dst = self.to_device_soa(p_xyz, width), self.empty((n,), np.uint8)
        self._call(name, dp.ptr, dst.ptr, n)
        return dst.download()

    def g1_projective_new(self, p_xyz): return self._projective_new("sylow_hip_g1_projective_new_batch", 12, p_xyz)
    def g2_projective_new(self, p_xyz): return self._projective_new("sylow_hip_g2_projective_new_batch", 24, p_xyz)

    def _ct_eq(self, name, width, a_xyz, b_xyz):
        a_xyz, b_xyz = _aos(a_xyz, width), _aos(b_xyz, width)
        n = a_xyz.shape[0]
        da, db, deq = self.to_device_soa(a_xyz, width), self.to_device_soa(b_xyz, width), self.empty((n,), np.uint8)
        self._call(name, da.ptr, db.ptr, deq.ptr, n)
        return deq.download()

    def g1_ct_eq(self, a_xyz, b_xyz): return self._ct_eq("sylow_hip_g1_ct_eq_batch", 12, a_xyz, b_xyz)
    def g2_ct_eq(self, a_xyz, b_xyz): return self._ct_eq("sylow_hip_g2_ct_eq_batch", 24, a_xyz, b_xyz)

    def _double(self, name, width, a_xy, a_inf):
        a_xy = _aos(a_xy, width)
        n = a_xy.shape[0]
        da, dai = self.to_device_soa(a_xy, width), self._flags(a_inf, n)
        do, doi = self.empty((width, n)), self.empty((n,), np.uint8)
        self._call(name, da.ptr, self._ptr(dai), do.ptr, doi.ptr, n)
        return self.from_device_soa(do), doi.download()

    def g1_double(self, a_xy, a_inf=None): return self._double("sylow_hip_g1_double_batch", 8, a_xy, a_inf)
    def g2_double(self, a_xy, a_inf=None): return self._double("sylow_hip_g2_double_batch", 16, a_xy, a_inf)

    def gt_pow(self, gt, k):
        return self._binop_w("sylow_hip_gt_pow_batch", 48, gt, 4, k)

    def _binop_w(self, name, wa, a, wb, b):
        a, b = _aos(a, wa), _aos(b, wb)
        n = a.shape[0]
        da, db = self.to_device_soa(a, wa), self.to_device_soa(b, wb)
        do = self.empty((wa, n))
        self._call(name, da.ptr, db.ptr, do.ptr, n)
        return self.from_device_soa(do)

    def _normalize(self, name, win, wout, p):
        p = _aos(p, win)
        n = p.shape[0]
        dp = self.to_device_soa(p, win)
        do, doi = self.empty((wout, n)), self.empty((n,), np.uint8)
        self._call(name, dp.ptr, do.ptr, doi.ptr, n)
        return self.from_device_soa(do), doi.download()

    def g1_normalize(self, p_xyz): return self._normalize("sylow_hip_g1_normalize_batch", 12, 8, p_xyz)
    def g2_normalize(self, p_xyz): return self._normalize("sylow_hip_g2_normalize_batch", 24, 16, p_xyz)

    def g1_on_curve(self, p_xy, p_inf=None):
        p_xy = _aos(p_xy, 8)
        n = p_xy.shape[0]
        dp, di, dst = self.to_device_soa(p_xy, 8), self._flags(p_inf, n), self.empty((n,), np.uint8)
        self._call("sylow_hip_g1_on_curve_batch", dp.ptr, self._ptr(di), dst.ptr, n)
        return dst.download()

    def g2_psi(self, q_xy, q_inf=None):
        q_xy = _aos(q_xy, 16)
        n = q_xy.shape[0]
        dq, di = self.to_device_soa(q_xy, 16), self._flags(q_inf, n)
        do, doi, dst = self.empty((16, n)), self.empty((n,), np.uint8), self.empty((n,), np.uint8)
        self._call("sylow_hip_g2_psi_batch", dq.ptr, self._ptr(di), do.ptr, doi.ptr, dst.ptr, n)
        return self.from_device_soa(do), doi.download(), dst.download()

    def g2_subgroup_check(self, q_xy, q_inf=None):
        q_xy = _aos(q_xy, 16)
        n = q_xy.shape[0]
        dq, di = self.to_device_soa(q_xy, 16), self._flags(q_inf, n)
        ds = self.empty((n,), np.uint8)
        self._call("sylow_hip_g2_subgroup_check_batch", dq.ptr, self._ptr(di), ds.ptr, n)
        return ds.download()

    # ---- pairing ---------------------------------------------------------------------------
    def miller_loop(self, p_xy, q_xy):
        p_xy, q_xy = _aos(p_xy, 8), _aos(q_xy, 16)
        n = p_xy.shape[0]
        dp, dq = self.to_device_soa(p_xy, 8), self.to_device_soa(q_xy, 16)
        do = self.empty((48, n))
        self._call("sylow_hip_miller_loop_batch", dp.ptr, dq.ptr, do.ptr, n)
        return self.from_device_soa(do)

    def final_exp(self, f):
        return self._unop("sylow_hip_final_exp_batch", 48, f)

    def g1_sum(self, p_xy, p_inf=None):
        """sum_i P_i as one G1 point (the `+` fold of examples/verify_multiple_messages_same_signer.rs:41-60): ([1, 8] affine words, [1] flag)."""
        p_xy = _aos(p_xy, 8)
        n = p_xy.shape[0]
        dp = self.to_device_soa(p_xy, 8) if n else None
        dpi = self._flags(p_inf, n) if n else None
        do, doi = self.empty((8, 1)), self.empty((1,), np.uint8)
        self._call("sylow_hip_g1_sum_batch", self._ptr(dp), self._ptr(dpi), n, do.ptr, doi.ptr)
        return self.from_device_soa(do), doi.download()

    def pairing(self, p_xy, q_xy, p_inf=None, q_inf=None, pipelined=True, chunk=0, out=None):
        """pairing() (pairing.rs:870-893) on host arrays: [n, 8] / [n, 16] words in, [n, 48] Gt words out.  Default: the chunked,
        double-buffered host pipeline (sylow_hip_pairing_host: the copies of chunk k - 1 / k + 1 run beside the kernels of chunk k);
        `pipelined=False` is upload -> sylow_hip_pairing_batch -> download on the engine's stream (bit-identical).  `out`: a [n, 48]
        uint64 array to fill (e.g. from `pinned_empty`)."""
        p_xy, q_xy = _aos(p_xy, 8), _aos(q_xy, 16)
        n = p_xy.shape[0]
        if pipelined:
            assert q_xy.shape[0] == n
            gt = out if out is not None else np.empty((n, 48), dtype=np.uint64)
            assert gt.shape == (n, 48) and gt.dtype == np.uint64 and gt.flags.c_contiguous
            pi = None if p_inf is None else np.ascontiguousarray(p_inf, dtype=np.uint8).reshape(n)
            qi = None if q_inf is None else np.ascontiguousarray(q_inf, dtype=np.uint8).reshape(n)
            _lib.check(self.lib.sylow_hip_set_device(self.device), "sylow_hip_set_device")
            _lib.check(self.lib.sylow_hip_pairing_host(p_xy.ctypes.data, None if pi is None else pi.ctypes.data, q_xy.ctypes.data,
                                                       None if qi is None else qi.ctypes.data, gt.ctypes.data, n, chunk), "sylow_hip_pairing_host")
            return gt
        dp, dq = self.to_device_soa(p_xy, 8), self.to_device_soa(q_xy, 16)
        dpi, dqi = self._flags(p_inf, n), self._flags(q_inf, n)
        do = self.empty((48, n))
        self._call("sylow_hip_pairing_batch", dp.ptr, self._ptr(dpi), dq.ptr, self._ptr(dqi), do.ptr, n)
        return self.from_device_soa(do)

    def multi_pairing(self, p_xy, q_xy, offsets, p_inf=None, q_inf=None, skip_infinity=False, want_gt=True):
        p_xy, q_xy = _aos(p_xy, 8), _aos(q_xy, 16)
        n = p_xy.shape[0]
        off = np.ascontiguousarray(offsets, dtype=np.uint64)
        nj = off.shape[0] - 1
        dp = self.to_device_soa(p_xy, 8) if n else self.empty((8, 1))
        dq = self.to_device_soa(q_xy, 16) if n else self.empty((16, 1))
        dpi, dqi = (self._flags(p_inf, n), self._flags(q_inf, n)) if n else (None, None)
        doff = self.to_device(off)
        dgt = self.empty((48, max(nj, 1))) if want_gt else None
        dis = self.empty((max(nj, 1),), np.uint8)
        self._call("sylow_hip_multi_pairing_batch", dp.ptr, self._ptr(dpi), dq.ptr, self._ptr(dqi), doff.ptr, nj, n,
                   1 if skip_infinity else 0, self._ptr(dgt), dis.ptr)
        gt = self.from_device_soa(dgt)[:nj] if want_gt else None
        return gt, dis.download()[:nj]

    def glued_miller_loop(self, p_xy, q_xy, offsets):
        p_xy, q_xy = _aos(p_xy, 8), _aos(q_xy, 16)
        n = p_xy.shape[0]
        off = np.ascontiguousarray(offsets, dtype=np.uint64)
        nj = off.shape[0] - 1
        dp = self.to_device_soa(p_xy, 8) if n else None
        dq = self.to_device_soa(q_xy, 16) if n else None
        doff, df = self.to_device(off), self.empty((48, max(nj, 1)))
        self._call("sylow_hip_glued_miller_loop_batch", self._ptr(dp), self._ptr(dq), doff.ptr, nj, n, df.ptr)
        return self.from_device_soa(df)[:nj]

    def pairing_product(self, p_xy, q_xy, p_inf=None, q_inf=None, skip_infinity=False):
        """prod_i e(P_i, Q_i) as ONE Gt, computed in parallel over the batch; returns (gt [1, 48], is_one)."""
        p_xy, q_xy = _aos(p_xy, 8), _aos(q_xy, 16)
        n = p_xy.shape[0]
        dp = self.to_device_soa(p_xy, 8) if n else None
        dq = self.to_device_soa(q_xy, 16) if n else None
        dpi, dqi = (self._flags(p_inf, n), self._flags(q_inf, n)) if n else (None, None)
        dgt, dis = self.empty((48, 1)), self.empty((1,), np.uint8)
        self._call("sylow_hip_pairing_product_batch", self._ptr(dp), self._ptr(dpi), self._ptr(dq), self._ptr(dqi), n,
                   1 if skip_infinity else 0, dgt.ptr, dis.ptr)
        return self.from_device_soa(dgt), bool(dis.download()[0])

    def pairing_product_partial(self, p_xy, q_xy, p_inf=None, q_inf=None, skip_infinity=False):
        """Raw Miller product of one shard (no final exponentiation), [1, 48]."""
        p_xy, q_xy = _aos(p_xy, 8), _aos(q_xy, 16)
        n = p_xy.shape[0]
        dp = self.to_device_soa(p_xy, 8) if n else None
        dq = self.to_device_soa(q_xy, 16) if n else None
        dpi, dqi = (self._flags(p_inf, n), self._flags(q_inf, n)) if n else (None, None)
        df = self.empty((48, 1))
        self._call("sylow_hip_pairing_product_partial_batch", self._ptr(dp), self._ptr(dpi), self._ptr(dq), self._ptr(dqi), n,
                   1 if skip_infinity else 0, df.ptr)
        return self.from_device_soa(df)

    def fp12_product_final_exp(self, parts):
        """final_exponentiation(prod parts) for parts [k, 48]; returns (gt [1, 48], is_one)."""
        parts = _aos(parts, 48)
        k = parts.shape[0]
        dpa = self.to_device_soa(parts, 48) if k else None
        dgt, dis = self.empty((48, 1)), self.empty((1,), np.uint8)
        self._call("sylow_hip_fp12_product_final_exp", self._ptr(dpa), k, dgt.ptr, dis.ptr)
        return self.from_device_soa(dgt), bool(dis.download()[0])

    def pairing_product_all(self, p_xy, q_xy, comm=None, p_inf=None, q_inf=None, skip_infinity=False):
        """glued_pairing over the union of all ranks' pairs (comm = raw ncclComm_t as int, None = one rank)."""
        p_xy, q_xy = _aos(p_xy, 8), _aos(q_xy, 16)
        n = p_xy.shape[0]
        dp = self.to_device_soa(p_xy, 8) if n else None
        dq = self.to_device_soa(q_xy, 16) if n else None
        dpi, dqi = (self._flags(p_inf, n), self._flags(q_inf, n)) if n else (None, None)
        dgt, dis = self.empty((48, 1)), self.empty((1,), np.uint8)
        self._call("sylow_hip_pairing_product_all", self._ptr(dp), self._ptr(dpi), self._ptr(dq), self._ptr(dqi), n,
                   1 if skip_infinity else 0, comm, dgt.ptr, dis.ptr)
        return self.from_device_soa(dgt), bool(dis.download()[0])

    def all_valid(self, dflags: DeviceArray, comm=None) -> int:
        """AND of this rank's flags AND-ed over every rank of `comm` (raw ncclComm_t as int, None = one rank)."""
        out = self.empty((1,), np.int32)
        self._call("sylow_hip_all_valid", dflags.ptr, dflags.shape[0], comm, out.ptr)
        return int(out.download()[0])

    # G2PreComputed consumers (pairing.rs:590-619, 970-1022): coeffs [m, 87*24] as g2_precompute returns them
    @staticmethod
    def _check_table_idx(table_idx, n, m):
        """table_idx is consumed unchecked on the device (coeffs[table_idx[i]]): validate it while it is still a host array."""
        if table_idx is None:
            if n != m:
                raise ValueError(f"without table_idx, pair i reads table i: {n} G1 points need {n} tables, got {m}")
            return None
        ti = np.ascontiguousarray(table_idx, dtype=np.uint64).reshape(-1)
        if ti.shape[0] != n:
            raise ValueError(f"table_idx has {ti.shape[0]} entries for {n} G1 points")
        if n and (m == 0 or int(ti.max()) >= m):
            raise ValueError(f"table_idx refers to table {int(ti.max()) if n else 0}, only {m} tables were given")
        return ti

    def miller_loop_precomputed(self, coeffs, p_xy, table_idx=None):
        coeffs, p_xy = _aos(coeffs, 87 * 24), _aos(p_xy, 8)
        n, m = p_xy.shape[0], coeffs.shape[0]
        table_idx = self._check_table_idx(table_idx, n, m)
        dc, dp = self.to_device_soa(coeffs, 87 * 24), self.to_device_soa(p_xy, 8)
        dti = self.to_device(np.ascontiguousarray(table_idx, dtype=np.uint64)) if table_idx is not None else None
        do = self.empty((48, n))
        self._call("sylow_hip_miller_loop_precomputed_batch", dc.ptr, m, self._ptr(dti), dp.ptr, do.ptr, n)
        return self.from_device_soa(do)

    def glued_miller_loop_precomputed(self, coeffs, p_xy, offsets, table_idx=None):
        coeffs, p_xy = _aos(coeffs, 87 * 24), _aos(p_xy, 8)
        n, m = p_xy.shape[0], coeffs.shape[0]
        table_idx = self._check_table_idx(table_idx, n, m)
        off = np.ascontiguousarray(offsets, dtype=np.uint64)
        nj = off.shape[0] - 1
        if nj < 0 or (nj >= 0 and (np.any(off[1:] < off[:-1]) or int(off[-1]) > n)):
            raise ValueError("offsets must be non-decreasing and end at most at the number of pairs")
        dc = self.to_device_soa(coeffs, 87 * 24) if m else None
        dp = self.to_device_soa(p_xy, 8) if n else None
        dti = self.to_device(np.ascontiguousarray(table_idx, dtype=np.uint64)) if table_idx is not None else None
        doff, df = self.to_device(off), self.empty((48, max(nj, 1)))
        self._call("sylow_hip_glued_miller_loop_precomputed_batch", self._ptr(dc), m, self._ptr(dti), self._ptr(dp), doff.ptr, nj, n, df.ptr)
        return self.from_device_soa(df)[:nj]

    # ---- wire formats ----------------------------------------------------------------------
    def _to_bytes(self, name, width, nbytes, xy, inf):
        xy = _aos(xy, width)
        n = xy.shape[0]
        d, di = self.to_device_soa(xy, width), self._flags(inf, n)
        do = self.empty((n * nbytes,), np.uint8)
        self._call(name, d.ptr, self._ptr(di), do.ptr, n)
        raw = do.download().tobytes()
        return [raw[nbytes * i:nbytes * (i + 1)] for i in range(n)]

    def _from_bytes(self, name, width, nbytes, blobs):
        n = len(blobs)
        assert all(len(b) == nbytes for b in blobs)
        din = self.to_device(np.frombuffer(b"".join(blobs), dtype=np.uint8))
        dxy, dinf, dst = self.empty((width, n)), self.empty((n,), np.uint8), self.empty((n,), np.uint8)
        self._call(name, din.ptr, dxy.ptr, dinf.ptr, dst.ptr, n)
        return self.from_device_soa(dxy), dinf.download(), dst.download()

    def g1_to_be_bytes(self, xy, inf=None): return self._to_bytes("sylow_hip_g1_to_be_bytes_batch", 8, 64, xy, inf)
    def g2_to_be_bytes(self, xy, inf=None): return self._to_bytes("sylow_hip_g2_to_be_bytes_batch", 16, 128, xy, inf)
    def g1_from_be_bytes(self, blobs): return self._from_bytes("sylow_hip_g1_from_be_bytes_batch", 8, 64, blobs)
    def g2_from_be_bytes(self, blobs): return self._from_bytes("sylow_hip_g2_from_be_bytes_batch", 16, 128, blobs)

    # ---- the value-typed calls on the reference's wire format (pipeline.hip) ----------------
    def pairing_from_bytes(self, p_blobs, q_blobs, chunk=0):
        """pairing() on G1Affine / G2Affine::to_be_bytes blobs (64 / 128 bytes each): decoding + validation on the device inside the host
        pipeline.  Returns (gt [n, 48], status_p [n], status_q [n]); an element with a non-zero status entered as the identity (Gt = 1)."""
        n = len(p_blobs)
        assert len(q_blobs) == n and all(len(b) == 64 for b in p_blobs) and all(len(b) == 128 for b in q_blobs)
        pb, qb = np.frombuffer(b"".join(p_blobs) or b"\0", dtype=np.uint8), np.frombuffer(b"".join(q_blobs) or b"\0", dtype=np.uint8)
        gt, sp, sq = np.empty((n, 48), dtype=np.uint64), np.empty((n,), dtype=np.uint8), np.empty((n,), dtype=np.uint8)
        _lib.check(self.lib.sylow_hip_set_device(self.device), "sylow_hip_set_device")
        _lib.check(self.lib.sylow_hip_pairing_host_bytes(pb.ctypes.data, qb.ctypes.data, gt.ctypes.data, sp.ctypes.data, sq.ctypes.data, n, chunk), "sylow_hip_pairing_host_bytes")
        return gt, sp, sq

    def bls_verify_from_bytes(self, pk_blobs, msgs, sig_blobs, chunk=0):
        """verify() on wire-format keys (128 bytes) and signatures (64 bytes).  Returns (ok [n], status_pk [n], status_sig [n])."""
        n = len(msgs)
        assert len(pk_blobs) == n and len(sig_blobs) == n and all(len(b) == 128 for b in pk_blobs) and all(len(b) == 64 for b in sig_blobs)
        kb, sb = np.frombuffer(b"".join(pk_blobs) or b"\0", dtype=np.uint8), np.frombuffer(b"".join(sig_blobs) or b"\0", dtype=np.uint8)
        blob = np.frombuffer(b"".join(msgs) or b"\0", dtype=np.uint8)
        off = np.zeros(n + 1, dtype=np.uint64)
        if n:
            off[1:] = np.cumsum([len(m) for m in msgs])
        ok, sk, ss = np.empty((n,), dtype=np.uint8), np.empty((n,), dtype=np.uint8), np.empty((n,), dtype=np.uint8)
        _lib.check(self.lib.sylow_hip_set_device(self.device), "sylow_hip_set_device")
        _lib.check(self.lib.sylow_hip_bls_verify_host_bytes(kb.ctypes.data, blob.ctypes.data, off.ctypes.data, sb.ctypes.data, ok.ctypes.data,
                                                            sk.ctypes.data, ss.ctypes.data, n, chunk), "sylow_hip_bls_verify_host_bytes")
        return ok, sk, ss

    # ---- hashing / BLS ---------------------------------------------------------------------
    def _msgs(self, msgs):
        off = np.zeros(len(msgs) + 1, dtype=np.uint64)
        for i, m in enumerate(msgs):
            off[i + 1] = off[i] + len(m)
        blob = np.frombuffer(b"".join(msgs) or b"\x00", dtype=np.uint8)
        return self.to_device(blob), self.to_device(off)

    def fext_op(self, op, a, b=None):
        """Component-wise FieldExtension operators on Fp2 / Fp6 / Fp12 batches: op in add / sub / neg / scale (b = one Fp per element)."""
        a = np.ascontiguousarray(a, dtype=np.uint64)
        n, width = a.shape
        degree = width // 4
        da, do = self.to_device_soa(a, width), self.empty((width, n))
        if op == "neg":
            self._call("sylow_hip_fext_neg_batch", da.ptr, do.ptr, degree, n)
        else:
            db = self.to_device_soa(_aos(b, 4 if op == "scale" else width), 4 if op == "scale" else width)
            self._call("sylow_hip_fext_%s_batch" % op, da.ptr, db.ptr, do.ptr, degree, n)
        return self.from_device_soa(do)

    def svdw_map(self, u):
        """SvdW map of field elements u [n, 4] -> (xy [n, 8], status [n])."""
        u = _aos(u, 4)
        n = u.shape[0]
        du = self.to_device_soa(u, 4)
        do, ds = self.empty((8, n)), self.empty((n,), np.uint8)
        self._call("sylow_hip_svdw_map_batch", du.ptr, do.ptr, ds.ptr, n)
        return self.from_device_soa(do), ds.download()

    def fp_compute_naf(self, k):
        """Fp::compute_naf on raw 256-bit values k [n, 4] -> (np [n, 4], nm [n, 4])."""
        k = _aos(k, 4)
        n = k.shape[0]
        dk = self.to_device_soa(k, 4)
        dp, dm = self.empty((4, n)), self.empty((4, n))
        self._call("sylow_hip_fp_compute_naf_batch", dk.ptr, dp.ptr, dm.ptr, n)
        return self.from_device_soa(dp), self.from_device_soa(dm)

    def hash_to_field(self, msgs, dst: bytes | None = None):
        n = len(msgs)
        dm, doff = self._msgs(msgs)
        do = self.empty((8, n))
        self._call("sylow_hip_hash_to_field_batch", dm.ptr, doff.ptr, dst, len(dst) if dst else 0, do.ptr, n)
        return self.from_device_soa(do)

    def hash_to_g1(self, msgs, dst: bytes | None = None):
        n = len(msgs)
        dm, doff = self._msgs(msgs)
        do, doi = self.empty((8, n)), self.empty((n,), np.uint8)
        self._call("sylow_hip_hash_to_g1_batch", dm.ptr, doff.ptr, dst, len(dst) if dst else 0, do.ptr, doi.ptr, n)
        return self.from_device_soa(do), doi.download()

    def bls_sign(self, sk, msgs):
        sk = _aos(sk, 4)
        n = len(msgs)
        dm, doff = self._msgs(msgs)
        dsk = self.to_device_soa(sk, 4)
        do, doi = self.empty((8, n)), self.empty((n,), np.uint8)
        self._call("sylow_hip_bls_sign_batch", dsk.ptr, dm.ptr, doff.ptr, do.ptr, doi.ptr, n)
        return self.from_device_soa(do), doi.download()

    def bls_verify(self, pk_xy, msgs, sig_xy, pk_inf=None, sig_inf=None, fused=False, two_pairings=False, pipelined=True, chunk=0):
        """verify (lib.rs:223-236).  Default and `fused`: one final exponentiation per element; `two_pairings`: the literal form.
        The default form runs through the chunked host pipeline (sylow_hip_bls_verify_host) unless `pipelined=False`.
        `msgs`: a list of bytes, or a (blob uint8 array, offsets uint64 [n + 1]) pair."""
        pk_xy, sig_xy = _aos(pk_xy, 16), _aos(sig_xy, 8)
        if isinstance(msgs, tuple):
            blob, off = np.ascontiguousarray(msgs[0], dtype=np.uint8), np.ascontiguousarray(msgs[1], dtype=np.uint64)
            n = off.shape[0] - 1
        else:
            n = len(msgs)
            blob, off = None, None
        if pipelined and not two_pairings and not fused:
            if blob is None:
                blob = np.frombuffer(b"".join(msgs) or b"\0", dtype=np.uint8)
                off = np.zeros(n + 1, dtype=np.uint64)
                if n:
                    off[1:] = np.cumsum([len(m) for m in msgs])
            ok = np.empty((n,), dtype=np.uint8)
            ki = None if pk_inf is None else np.ascontiguousarray(pk_inf, dtype=np.uint8).reshape(n)
            si = None if sig_inf is None else np.ascontiguousarray(sig_inf, dtype=np.uint8).reshape(n)
            _lib.check(self.lib.sylow_hip_set_device(self.device), "sylow_hip_set_device")
            _lib.check(self.lib.sylow_hip_bls_verify_host(pk_xy.ctypes.data, None if ki is None else ki.ctypes.data, blob.ctypes.data, off.ctypes.data,
                                                          sig_xy.ctypes.data, None if si is None else si.ctypes.data, ok.ctypes.data, n, chunk), "sylow_hip_bls_verify_host")
            return ok
        if blob is not None:
            msgs = [bytes(blob[int(off[i]):int(off[i + 1])]) for i in range(n)]
        dm, doff = self._msgs(msgs)
        dpk, dsig = self.to_device_soa(pk_xy, 16), self.to_device_soa(sig_xy, 8)
        dpi, dsi = self._flags(pk_inf, n), self._flags(sig_inf, n)
        dok = self.empty((n,), np.uint8)
        name = "sylow_hip_bls_verify_two_pairings_batch" if two_pairings else "sylow_hip_bls_verify_fused_batch" if fused else "sylow_hip_bls_verify_batch"
        self._call(name, dpk.ptr, self._ptr(dpi), dm.ptr, doff.ptr, dsig.ptr, self._ptr(dsi), dok.ptr, n)
        return dok.download()

    def bls_aggregate_verify(self, pk_xy, msgs, sig_xy, pk_inf=None, sig_inf=None, comm=None):
        """One boolean for the whole batch: prod_i e(sig_i, G2gen) e(-H(msg_i), pk_i) == 1 (one key row = the same signer).
        Returns (gt [48] words, is_one)."""
        pk_xy, sig_xy = _aos(pk_xy, 16), _aos(sig_xy, 8)
        n, n_pk = len(msgs), pk_xy.shape[0]
        assert sig_xy.shape[0] == n and (n_pk in (1, n) or n == 0)
        dm, doff = self._msgs(msgs)
        dpk = self.to_device_soa(pk_xy, 16) if n_pk else None
        dsig = self.to_device_soa(sig_xy, 8) if n else None
        dpi, dsi = (self._flags(pk_inf, n_pk) if n_pk else None), (self._flags(sig_inf, n) if n else None)
        dgt, done = self.empty((48, 1)), self.empty((1,), np.uint8)
        self._call("sylow_hip_bls_aggregate_verify_batch", self._ptr(dpk), self._ptr(dpi), n_pk, dm.ptr, doff.ptr, self._ptr(dsig), self._ptr(dsi), n, comm, dgt.ptr, done.ptr)
        return self.from_device_soa(dgt)[0], int(done.download()[0])

    def bls_batch_verify_weighted(self, pk_xy, msgs, sig_xy, weights, pk_inf=None, sig_inf=None, comm=None):
        """The small-exponent batch test prod_i [e(sig_i, G2gen) e(-H(m_i), pk_i)]^(w_i) == identity (sound one-boolean batch
        verification); weights [n, 4] Fp values drawn by the caller after the signatures are fixed.  Returns (Gt words, bool)."""
        pk_xy, sig_xy, weights = _aos(pk_xy, 16), _aos(sig_xy, 8), _aos(weights, 4)
        n, n_pk = sig_xy.shape[0], pk_xy.shape[0]
        assert len(msgs) == n and weights.shape[0] == n and n_pk in (1, n)
        dpk, dsig, dw = self.to_device_soa(pk_xy, 16), self.to_device_soa(sig_xy, 8), self.to_device_soa(weights, 4)
        dm, doff = self._msgs(msgs)
        dpi, dsi = self._flags(pk_inf, n_pk), self._flags(sig_inf, n)
        dgt, dis = self.empty((48, 1)), self.empty((1,), np.uint8)
        self._call("sylow_hip_bls_batch_verify_weighted", dpk.ptr, self._ptr(dpi), n_pk, dm.ptr, doff.ptr, dsig.ptr, self._ptr(dsi), dw.ptr, n, comm, dgt.ptr, dis.ptr)
        return self.from_device_soa(dgt), bool(dis.download()[0])

    def bls_verify_same_signer(self, pk_xy, msgs, sig_xy, pk_inf=None, sig_inf=None):
        pk_xy, sig_xy = _aos(pk_xy, 16), _aos(sig_xy, 8)
        assert pk_xy.shape[0] == 1
        n = len(msgs)
        dm, doff = self._msgs(msgs)
        dpk, dsig = self.to_device_soa(pk_xy, 16), self.to_device_soa(sig_xy, 8)
        dpi, dsi = self._flags(pk_inf, 1), self._flags(sig_inf, n)
        dok = self.empty((n,), np.uint8)
        self._call("sylow_hip_bls_verify_same_signer_batch", dpk.ptr, self._ptr(dpi), dm.ptr, doff.ptr, dsig.ptr, self._ptr(dsi), dok.ptr, n)
        return dok.download()

    def g2_line_table(self, pk_xy) -> DeviceArray:
        """Device-resident line table of ONE key (opaque int32 words): build once, reuse across bls_verify_line_table calls."""
        pk_xy = _aos(pk_xy, 16)
        assert pk_xy.shape[0] == 1
        dpk = self.to_device_soa(pk_xy, 16)
        table = self.empty((int(self.lib.sylow_hip_g2_line_table_words()),), np.int32)
        self._call("sylow_hip_g2_line_table", dpk.ptr, 1, 0, table.ptr)
        self.sync()                      # dpk is released when this frame returns
        return table

    def bls_verify_line_table(self, table: DeviceArray, msgs, sig_xy, pk_inf=None, sig_inf=None):
        sig_xy = _aos(sig_xy, 8)
        n = len(msgs)
        dm, doff = self._msgs(msgs)
        dsig = self.to_device_soa(sig_xy, 8)
        dpi, dsi = self._flags(pk_inf, 1), self._flags(sig_inf, n)
        dok = self.empty((n,), np.uint8)
        self._call("sylow_hip_bls_verify_line_table_batch", table.ptr, self._ptr(dpi), dm.ptr, doff.ptr, dsig.ptr, self._ptr(dsi), dok.ptr, n)
        return dok.download()

    def g2_precompute(self, q_xy):
        q_xy = _aos(q_xy, 16)
        n = q_xy.shape[0]
        dq = self.to_device_soa(q_xy, 16)
        dc = self.empty((87 * 24, n))
        self._call("sylow_hip_g2_precompute_batch", dq.ptr, dc.ptr, n)
        return self.from_device_soa(dc)

    def flags_all(self, dflags: DeviceArray) -> int:
        out = self.empty((1,), np.int32)
        self._call("sylow_hip_flags_all", dflags.ptr, dflags.shape[0], out.ptr)
        return int(out.download()[0])
