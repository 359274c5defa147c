"""Model of the Jacobi symbol by sign-free divsteps ("posdivsteps": the swap step keeps (g + f) / 2 instead of (g - f) / 2, so f and g stay
non-negative and the symbol can be tracked from the low bits of f and g alone -- the variant libsecp256k1's jacobi32 uses), for BN254's p:
checks the symbol against Euler's criterion and prints how many steps random inputs need (average, and the slowest of every 64: a
wavefront runs until its last lane is done).  Result (docs/REJECTED.md): ~748 steps on average, ~806 per wavefront = 27 batches of 30,
against ~512 plain divsteps (18 batches) for the inversion -- no faster than the shipped binary algorithm, so it was not built."""
import math
import random
import statistics

P = 0x30644E72E131A029B85045B68181585D97816A916871CA8D3C208C16D87CFD47


def jacobi_posdiv(a, eta=-1):
    f, g, jac, steps = P, a, 0, 0
    if g == 0:
        return 0, 0
    while f != 1:
        if g & 1:
            if eta < 0:
                eta = -eta
                jac ^= ((f & g) >> 1) & 1          # reciprocity: both 3 mod 4
                f, g = g, f
            g += f
        eta -= 1
        g >>= 1
        jac ^= ((f >> 1) ^ (f >> 2)) & 1           # (2 / f) = -1 iff f = 3, 5 mod 8
        steps += 1
        if steps > 5000:
            return None, steps
    return (-1 if jac else 1), steps


if __name__ == "__main__":
    random.seed(1)
    cnt = []
    for _ in range(6400):
        a = random.randrange(1, P)
        j, s = jacobi_posdiv(a)
        e = pow(a, (P - 1) // 2, P)
        assert j == (-1 if e == P - 1 else e)
        cnt.append(s)
    waves = [max(cnt[i:i + 64]) for i in range(0, len(cnt) - 63, 64)]
    print("steps: avg %.1f, min %d, max %d; slowest of 64: avg %.1f, max %d; batches of 30 per wavefront: %.2f"
          % (statistics.mean(cnt), min(cnt), max(cnt), statistics.mean(waves), max(waves), statistics.mean(math.ceil(x / 30) for x in waves)))
