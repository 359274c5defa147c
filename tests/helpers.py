"""Shared test helpers: deterministic inputs (SplitMix64-seeded xoshiro256**)."""
import os

import numpy as np

P = 0x30644E72E131A029B85045B68181585D97816A916871CA8D3C208C16D87CFD47
M64 = (1 << 64) - 1
SEED = int(os.environ.get("SYLOW_TEST_SEED", "0x53594C4F57"), 0)  # "SYLOW"; another seed re-draws every PRNG-built input of the suite


class Xoshiro:
    def __init__(self, seed):
        s = []
        x = seed & M64
        for _ in range(4):  # SplitMix64
            x = (x + 0x9E3779B97F4A7C15) & M64
            z = x
            z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & M64
            z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & M64
            s.append(z ^ (z >> 31))
        self.s = s

    def next(self):
        s = self.s
        r = (((s[1] * 5) & M64) << 7 | ((s[1] * 5) & M64) >> 57) & M64
        r = (r * 9) & M64
        t = (s[1] << 17) & M64
        s[2] ^= s[0]; s[3] ^= s[1]; s[1] ^= s[2]; s[0] ^= s[3]
        s[2] ^= t
        s[3] = ((s[3] << 45) | (s[3] >> 19)) & M64
        return r

    def fp(self):
        """uniform in [0, p) by rejection of 256-bit draws"""
        while True:
            v = self.next() | (self.next() << 64) | (self.next() << 128) | (self.next() << 192)
            v &= (1 << 254) - 1
            if v < P:
                return v

    def u256(self):
        return self.next() | (self.next() << 64) | (self.next() << 128) | (self.next() << 192)


def limbs(vals):
    out = np.zeros((len(vals), 4), dtype=np.uint64)
    for i, v in enumerate(vals):
        for k in range(4):
            out[i, k] = (int(v) >> (64 * k)) & M64
    return out


def pack(vals, width):
    return limbs(vals).reshape(-1, width)


def ints(arr):
    arr = np.asarray(arr, dtype=np.uint64).reshape(-1, 4)
    return [sum(int(arr[i, k]) << (64 * k) for k in range(4)) for i in range(arr.shape[0])]


def rand_fp_array(rng, n, width_fp):
    """n elements of width_fp Fp each -> (n, 4*width_fp) uint64"""
    return limbs([rng.fp() for _ in range(n * width_fp)]).reshape(n, 4 * width_fp)


def fast_rand_fp_array(seed, n, width_fp):
    """numpy-generated values < 2^253 (< p): for large batches where python-int loops are too slow"""
    g = np.random.default_rng(seed)
    a = g.integers(0, 1 << 63, size=(n, width_fp, 4), dtype=np.uint64) * np.uint64(2) + g.integers(0, 2, size=(n, width_fp, 4), dtype=np.uint64)
    a[:, :, 3] &= np.uint64((1 << 61) - 1)
    return a.reshape(n, 4 * width_fp)


def _R():
    from oracle import pyref
    return pyref


def fp2_sqrt(a):
    """sqrt in Fp2 = Fp[u]/(u^2+1), p = 3 mod 4 (complex method); None if a is not a square"""
    a0, a1 = a
    if a1 == 0:
        s = _R().fp_sqrt(a0)
        if s is not None:
            return (s, 0)
        s = _R().fp_sqrt((-a0) % P)
        return (0, s)
    n = _R().fp_sqrt((a0 * a0 + a1 * a1) % P)
    if n is None:
        return None
    for nn in (n, (-n) % P):
        h = (a0 + nn) * _R().fp_inv(2) % P
        x0 = _R().fp_sqrt(h)
        if x0 is not None and x0 != 0:
            x1 = a1 * _R().fp_inv(2 * x0 % P) % P
            if _R().fp2_square((x0, x1)) == (a0 % P, a1 % P):
                return (x0, x1)
    return None
