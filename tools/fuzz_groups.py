"""Seeded differential fuzz of the group-law and pairing entry points against the oracle (the routines that changed most in round 4: lazy
linear layer, isomorphic twist, the {17, 35} chain, the one-kernel pairing on isomorphic curves).  Every round draws fresh points, scalars
with planted edge values, identity flags, twist points inside and outside the r-torsion, and compares affine results / status codes / Gt
values with oracle/ bit for bit.   python3 tools/fuzz_groups.py [seconds] [seed]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from helpers import P, Xoshiro, fp2_sqrt, limbs, pack
from oracle import coracle as C, pyref as R
import sylow_amd

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed = int(sys.argv[2], 0) if len(sys.argv) > 2 else 1
rng = Xoshiro(seed)
eng = sylow_amd.Engine(0)
G1 = [1, 2]
G2 = list(R.G2_GEN_AFF[0]) + list(R.G2_GEN_AFF[1])
ONE4 = np.array([[1, 0, 0, 0]], dtype=np.uint64)
r = R.R_ORDER
EDGE = [0, 1, 2, 3, r - 1, r, r + 1, P - 1, (1 << 253) - 1, 1 << 127, (1 << 128) - 1, 17, 35, 4965661367192848881, 4965661367192848880]

# A FLAGGED affine point is the identity whatever its coordinate words hold: what From<Affine> for Projective gives the reference is the
# canonical (0 : 1 : 0) (group.rs:271-277), and that is what the oracle is fed (round 6: the group-law kernels load flagged points the same way)
def g1_proj(xy, inf):
    z = np.repeat(ONE4, xy.shape[0], 0) * (1 - inf.astype(np.uint64))[:, None]
    out = np.concatenate([xy, z], axis=1)
    out[inf.astype(bool)] = np.array([0, 0, 0, 0, 1, 0, 0, 0, 0, 0, 0, 0], dtype=np.uint64)
    return out
def g2_proj(xy, inf):
    z = np.concatenate([np.repeat(ONE4, xy.shape[0], 0), np.zeros((xy.shape[0], 4), np.uint64)], axis=1) * (1 - inf.astype(np.uint64))[:, None]
    out = np.concatenate([xy, z], axis=1)
    ident = np.zeros(24, dtype=np.uint64); ident[8] = 1
    out[inf.astype(bool)] = ident
    return out
def scalars(n):
    ks = [rng.fp() for _ in range(n)]
    for j in range(min(n, 6)):
        ks[rng.next() % n] = EDGE[rng.next() % len(EDGE)] % P
    return ks
def twist_points(m):
    out = []
    while len(out) < m:
        x = (rng.fp(), rng.fp())
        y = fp2_sqrt(R.fp2_add(R.fp2_mul(R.fp2_square(x), x), R.TWIST_B))
        if y is not None: out.append(list(x) + list(y))
    return pack([v for q in out for v in q], 16)

t0 = time.time(); rounds = checks = 0
while time.time() - t0 < budget:
    n = 8 + rng.next() % 120
    b1, _ = eng.g1_scalar_mul(np.repeat(pack(G1, 8), n, 0), limbs([rng.fp() or 1 for _ in range(n)]))
    b2, _ = eng.g2_scalar_mul(np.repeat(pack(G2, 16), n, 0), limbs([rng.fp() or 1 for _ in range(n)]))
    i1 = np.array([rng.next() % 23 == 0 for _ in range(n)], np.uint8); i2 = np.array([rng.next() % 29 == 0 for _ in range(n)], np.uint8)
    k = limbs(scalars(n))
    # G1 / G2 scalar multiplication (generic and r-torsion split), generator tables
    gx, gi = eng.g1_scalar_mul(b1, k, p_inf=i1); ex, ei = C.g1_to_affine(C.g1_scalar_mul(g1_proj(b1, i1), k))
    assert np.array_equal(gi, ei) and np.array_equal(gx, ex), ("g1 mul", seed, rounds)
    gx, gi = eng.g2_scalar_mul(b2, k, p_inf=i2); ex, ei = C.g2_to_affine(C.g2_scalar_mul(g2_proj(b2, i2), k))
    assert np.array_equal(gi, ei) and np.array_equal(gx, ex), ("g2 mul", seed, rounds)
    sx, si = eng.g2_scalar_mul(b2, k, p_inf=i2, subgroup=True)
    assert np.array_equal(si, ei) and np.array_equal(sx, ex), ("g2 mul split", seed, rounds)
    gx, gi = eng.g2_generator_mul(k); ex, ei = C.g2_to_affine(C.g2_scalar_mul(g2_proj(np.repeat(pack(G2, 16), n, 0), np.zeros(n, np.uint8)), k))
    assert np.array_equal(gi, ei) and np.array_equal(gx, ex), ("g2 gen", seed, rounds)
    gx, gi = eng.g1_generator_mul(k); ex, ei = C.g1_to_affine(C.g1_scalar_mul(g1_proj(np.repeat(pack(G1, 8), n, 0), np.zeros(n, np.uint8)), k))
    assert np.array_equal(gi, ei) and np.array_equal(gx, ex), ("g1 gen", seed, rounds)
    # additions with planted P + P, P - P, identities
    c2 = np.roll(b2, 1, axis=0).copy(); c2[0] = b2[0]
    if n > 1: c2[1] = b2[1]; c2[1, 8:] = limbs([(P - v) % P for v in [int(x) for x in C.from_limbs(b2[1:2, 8:12])] + [int(x) for x in C.from_limbs(b2[1:2, 12:16])]]).reshape(-1)
    gx, gi = eng.g2_add(b2, c2, i2, np.roll(i2, 3)); ex, ei = C.g2_to_affine(C.g2_add(g2_proj(b2, i2), g2_proj(c2, np.roll(i2, 3))))
    assert np.array_equal(gi, ei) and np.array_equal(gx, ex), ("g2 add", seed, rounds)
    # subgroup check: r-torsion points, arbitrary twist points, perturbed points
    m = 6
    tw = twist_points(m)
    pts = np.concatenate([b2[:m], tw, b2[:2] ^ np.uint64(1)])
    st = eng.g2_subgroup_check(pts)
    exp = C.g2_projective_new(g2_proj(pts[:2 * m], np.zeros(2 * m, np.uint8)))
    assert np.array_equal(st[:2 * m], exp) and st[:m].tolist() == [0] * m and st[m:2 * m].tolist() == [2] * m and st[2 * m:].tolist() == [1, 1], ("subgroup", seed, rounds)
    # pairing: the one-wavefront route and (every fourth round) the one-kernel route on isomorphic curves, identities included
    one = np.zeros((n, 4), np.uint64); one[:, 0] = 1
    big = rounds % 4 == 0
    if big:
        reps = 1100 // n + 1
        pp, qq, ii1, ii2 = np.tile(b1, (reps, 1)), np.tile(np.roll(b2, 1, axis=0), (reps, 1)), np.tile(i1, reps), np.tile(i2, reps)
    else:
        pp, qq, ii1, ii2 = b1, b2, i1, i2
    gt = eng.pairing(pp, qq, p_inf=ii1, q_inf=ii2, pipelined=False)
    idx = np.arange(0, pp.shape[0], max(1, pp.shape[0] // 6))[:6]
    e = C.final_exponentiation(C.miller_loop(pp[idx], qq[idx]))
    ident = np.zeros(48, np.uint64); ident[0] = 1
    e[(ii1[idx] | ii2[idx]).astype(bool)] = ident
    assert np.array_equal(gt[idx], e), ("pairing", seed, rounds, big)
    rounds += 1; checks += 9
print("fuzz_groups ok: seed %s, %d rounds, %d oracle comparisons, %.0f s" % (sys.argv[2] if len(sys.argv) > 2 else "1", rounds, checks, time.time() - t0))
