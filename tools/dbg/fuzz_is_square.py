"""fp_is_square / fp_sqrt on structured and random inputs against Python's pow (the variable-length Jacobi cascade and the pinned
window-table fetch of the square-root chain)"""
import os, sys, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, sylow_amd
P = 21888242871839275222246405745257275088696311157297823662689037894645226208583
eng = sylow_amd.Engine(0)
rnd = random.Random(12345)
vals = [0, 1, 2, 3, 4, 5, 7, 8, P - 1, P - 2, P - 3, (P - 1) // 2, (P + 1) // 2, (P + 1) // 4]
for b in range(1, 254):
    vals += [1 << b, (1 << b) - 1, (1 << b) + 1, P - (1 << b) if (1 << b) < P else 1]
for limbs in range(1, 9):
    for _ in range(2000):
        vals.append(rnd.getrandbits(32 * limbs) % P)
for _ in range(40000):
    vals.append(rnd.randrange(P))
# squares of small and random numbers, non-residues times squares
for _ in range(5000):
    x = rnd.randrange(P); vals.append(x * x % P)
n = len(vals)
a = np.array([[(v >> (64 * i)) & (2**64 - 1) for i in range(4)] for v in vals], dtype=np.uint64)
got = eng.fp_is_square(a).astype(bool)
exp = np.array([pow(v, (P - 1) // 2, P) in (0, 1) for v in vals])
bad = np.nonzero(got != exp)[0]
print("is_square: n =", n, "mismatches:", len(bad), [hex(vals[i]) for i in bad[:5]])
r, ok = eng.fp_sqrt(a)
ok = ok.astype(bool)
assert np.array_equal(ok, exp), "sqrt flag"
for i in range(0, n, 7):
    if exp[i]:
        y = sum(int(r[i, j]) << (64 * j) for j in range(4))
        assert y * y % P == vals[i], hex(vals[i])
print("sqrt ok")
sys.exit(1 if len(bad) else 0)
