#!/usr/bin/env python3
"""Exhaustive check of the quotient estimate used by fp_mul9_addsub (sylow_amd/csrc/bn254_fp.hpp):
for t in [0, 10p), q^ = ((t >> 250) * 677) >> 13 equals floor(t/p) or floor(t/p) - 1."""
import random
P = 0x30644E72E131A029B85045B68181585D97816A916871CA8D3C208C16D87CFD47
qhat = lambda t: ((t >> 250) * 677) >> 13
cands = set()
for q in range(11):
    for d in range(-3, 4):
        cands.add(q * P + d)
for h in range(125):
    for d in (-1, 0, 1):
        cands.add((h << 250) + d)
random.seed(5)
cands |= {random.randrange(10 * P) for _ in range(500000)}
for t in cands:
    if 0 <= t < 10 * P:
        assert 0 <= t // P - qhat(t) <= 1, hex(t)
# between consecutive boundaries both floor(t/p) and q^ are constant, so the boundary set is exhaustive
print("ok: q^ in {q, q-1} on [0, 10p)")
