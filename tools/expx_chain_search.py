"""Search for a cheaper f -> f^x chain (x = the BN parameter) for a UNITARY f: cyclotomic squaring = 6 product leaves, product = 18, conjugate
free.  Family searched: signed sliding windows over a table of odd powers T (1 in T), recoding by dynamic programming over the bits of x
(minimum number of non-zero digits from +-T, any zero runs), plus the cheapest way found to build T (additions / subtractions / doublings of
entries).  Prints the best (multiplications, squarings) per table size.   python3 tools/expx_chain_search.py"""
import itertools, sys
X = 4965661367192848881
S_COST, M_COST = 6, 18

def recode(x, T):
    """min #nonzero signed digits d_i in +-T with sum d_i 2^i = x; returns (count, digits LSB first)"""
    from functools import lru_cache
    Tset = sorted(T)
    sys.setrecursionlimit(10000)
    best = {}
    def go(v):
        # v: remaining value (can be negative), returns (count, list)
        if v == 0: return (0, [])
        if v in best: return best[v]
        if v % 2 == 0:
            c, d = go(v // 2)
            r = (c, [0] + d)
        else:
            r = None
            for t in Tset:
                for sgn in (1, -1):
                    d0 = sgn * t
                    w = v - d0
                    if w % 2: continue
                    if abs(w) >= abs(v) * 2 and abs(v) > 64: continue
                    if abs(w // 2) > abs(v) and abs(v) > 2 * max(Tset): continue
                    if abs(v) <= max(Tset) and w != 0 and abs(w//2) >= abs(v): continue
                    c, d = go(w // 2)
                    if r is None or c + 1 < r[0]:
                        r = (c + 1, [d0] + d)
        best[v] = r
        return r
    return go(x)

def build_cost(T):
    """(muls, sqrs) to get all of T from 1: BFS over sets, allowing a+b, a-b, 2a of known values (values up to 2*max)"""
    target = set(T) - {1}
    if not target: return (0, 0, [])
    lim = 2 * max(T) + 2
    start = frozenset([1])
    frontier = {start: (0, 0, [])}
    bestc = None
    for depth in range(1, 9):
        nxt = {}
        for known, (m, s, path) in frontier.items():
            ks = sorted(known)
            cands = []
            for a in ks:
                if 2 * a <= lim and 2 * a not in known: cands.append((2 * a, 0, 1, "%d=2*%d" % (2 * a, a)))
            for a in ks:
                for b in ks:
                    if a + b <= lim and a + b not in known and a <= b: cands.append((a + b, 1, 0, "%d=%d+%d" % (a + b, a, b)))
                    if a > b and a - b not in known: cands.append((a - b, 1, 0, "%d=%d-%d" % (a - b, a, b)))
            for v, dm, ds, desc in cands:
                nk = frozenset(known | {v})
                c = (m + dm, s + ds, path + [desc])
                cost = c[0] * M_COST + c[1] * S_COST
                if nk not in nxt or cost < nxt[nk][0] * M_COST + nxt[nk][1] * S_COST: nxt[nk] = c
                if target <= nk:
                    if bestc is None or cost < bestc[0] * M_COST + bestc[1] * S_COST: bestc = c
        if bestc is not None: return bestc
        # prune: keep states that contain progress
        frontier = dict(sorted(nxt.items(), key=lambda kv: (-len(target & kv[0]), kv[1][0] * M_COST + kv[1][1] * S_COST))[:4000])
    return (99, 99, [])

def main():
    odds = list(range(3, 64, 2))
    results = []
    for size in range(0, 4):
        for extra in itertools.combinations(odds, size):
            T = (1,) + extra
            cnt, digs = recode(X, T)
            top = len(digs) - 1
            while digs[top] == 0: top -= 1
            # chain: start from digs[top] (a table entry), then `top` squarings, one mul per further nonzero digit
            muls = cnt - 1
            sq = top
            results.append((muls, sq, T, digs))
    out = []
    for muls, sq, T, digs in results:
        if muls > 14: continue
        bm, bs, path = build_cost(T)
        out.append((muls + bm, sq + bs, T, path, digs))
    out.sort(key=lambda r: r[0] * M_COST + r[1] * S_COST)
    for r in out[:12]:
        print("M=%d S=%d cost=%d  T=%s  build=%s" % (r[0], r[1], r[0] * M_COST + r[1] * S_COST, r[2], r[3]))
        print("   digits MSB first:", [d for d in r[4][::-1]])
    print("current: width-4 signed windows: M=16 S=63 cost=%d" % (16 * M_COST + 63 * S_COST))

main()
