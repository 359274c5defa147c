"""Search for a cheaper f -> f^x chain (x = the BN parameter) for a UNITARY f: cyclotomic squaring = 6 product leaves, product = 18, conjugate
free.  Family searched: signed sliding windows over a table of odd powers T (1 in T), recoding by dynamic programming over the bits of x
(minimum number of non-zero digits from +-T, any zero runs), plus the cheapest way found to build T (additions / subtractions / doublings of
entries).  Prints the best (multiplications, squarings) per table size.   python3 tools/expx_chain_search.py"""
import itertools, sys
X = 4965661367192848881
S_COST, M_COST = (float(sys.argv[3]), float(sys.argv[4])) if len(sys.argv) > 4 else (6, 18)      # group law on lane pairs: 7.5 12.4

def recode(x, T):
    """min #nonzero signed digits d_i in +-T with sum d_i 2^i = x: exact search over the remaining value v -> (v - d) / 2 (memoised; a value
    being expanded counts as unreachable, which only removes cycles).  Returns (count, digits LSB first)."""
    sys.setrecursionlimit(100000)
    Tset = sorted(T)
    INF = (10 ** 9, None)
    memo = {}
    def go(v):
        if v == 0: return (0, [])
        if v in memo: return memo[v]
        memo[v] = INF                                   # in progress
        if v % 2 == 0:
            c, d = go(v // 2)
            r = (c, [0] + d) if d is not None else INF
        else:
            r = INF
            for t in Tset:
                for d0 in (t, -t):
                    w = v - d0
                    if abs(w) > 2 * abs(v) + 2: continue          # never useful: the value would grow
                    c, d = go(w // 2)
                    if d is not None and c + 1 < r[0]: r = (c + 1, [d0] + d)
        memo[v] = r
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
    top_odd = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    max_size = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    odds = list(range(3, top_odd, 2))
    results = []
    for size in range(0, max_size + 1):
        for extra in itertools.combinations(odds, size):
            T = (1,) + extra
            cnt, digs = recode(X, T)
            if digs is None: continue
            top = len(digs) - 1
            while digs[top] == 0: top -= 1
            # chain: start from digs[top] (a table entry), then `top` squarings, one mul per further nonzero digit
            muls = cnt - 1
            sq = top
            results.append((muls, sq, T, digs))
    out = []
    for muls, sq, T, digs in results:
        if muls > 17: continue
        bm, bs, path = build_cost(T)
        out.append((muls + bm, sq + bs, T, path, digs))
    out.sort(key=lambda r: r[0] * M_COST + r[1] * S_COST)
    for r in out[:12]:
        print("M=%d S=%d cost=%.1f  T=%s  build=%s" % (r[0], r[1], r[0] * M_COST + r[1] * S_COST, r[2], r[3]))
        print("   digits MSB first:", [d for d in r[4][::-1]])
    print("rounds 1-3: width-4 signed windows: M=16 S=63 cost=%.1f" % (16 * M_COST + 63 * S_COST))

main()
