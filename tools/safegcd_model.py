"""Bit-level model of the 30-bit-limb safegcd modular inversion (Bernstein-Yang divsteps, 20 x 30 steps, half-delta start)
used by fp_inv in sylow_amd/csrc/bn254_f29.hpp; validates the transition-matrix / update formulas and prints the constants."""
import random
P = 0x30644E72E131A029B85045B68181585D97816A916871CA8D3C208C16D87CFD47
M30 = (1 << 30) - 1
def i32(x): x &= 0xffffffff; return x - (1 << 32) if x >> 31 else x
def u32(x): return x & 0xffffffff
def to30(x): return [(x >> (30 * i)) & M30 for i in range(9)]
def val(v): return sum(v[i] << (30 * i) for i in range(9))
PL = to30(P); PINV30 = pow(P, -1, 1 << 30)

def divsteps_30(zeta, f0, g0):
    u, v, q, r = 1, 0, 0, 1
    f, g = u32(f0), u32(g0)
    for _ in range(30):
        c1 = u32(zeta >> 31)                 # all ones if zeta < 0
        c2 = u32(-(g & 1))
        x = u32((f ^ c1) - c1); y = u32((u ^ c1) - c1); z = u32((v ^ c1) - c1)
        g = u32(g + (x & c2)); q = u32(q + (y & c2)); r = u32(r + (z & c2))
        c1 &= c2
        zeta = i32((u32(zeta) ^ c1) - 1)
        f = u32(f + (g & c1)); u = u32(u + (q & c1)); v = u32(v + (r & c1))
        g >>= 1; u = u32(u << 1); v = u32(v << 1)
    return zeta, (i32(u), i32(v), i32(q), i32(r))

def update_fg(f, g, t):
    u, v, q, r = t
    cf = u * f[0] + v * g[0]; cg = q * f[0] + r * g[0]
    assert cf & M30 == 0 and cg & M30 == 0
    cf >>= 30; cg >>= 30
    nf, ng = [0] * 9, [0] * 9
    for i in range(1, 9):
        cf += u * f[i] + v * g[i]; cg += q * f[i] + r * g[i]
        nf[i - 1] = cf & M30; cf >>= 30
        ng[i - 1] = cg & M30; cg >>= 30
    nf[8] = cf; ng[8] = cg
    assert -2**31 <= cf < 2**31 and -2**31 <= cg < 2**31
    return nf, ng

def update_de(d, e, t):
    u, v, q, r = t
    sd = -1 if d[8] < 0 else 0; se = -1 if e[8] < 0 else 0
    md = (u & sd) + (v & se); me = (q & sd) + (r & se)
    cd = u * d[0] + v * e[0]; ce = q * d[0] + r * e[0]
    md -= (PINV30 * (cd & 0xffffffff) + md) & M30
    me -= (PINV30 * (ce & 0xffffffff) + me) & M30
    cd += PL[0] * md; ce += PL[0] * me
    assert cd & M30 == 0 and ce & M30 == 0
    cd >>= 30; ce >>= 30
    nd, ne = [0] * 9, [0] * 9
    for i in range(1, 9):
        cd += u * d[i] + v * e[i] + PL[i] * md
        ce += q * d[i] + r * e[i] + PL[i] * me
        nd[i - 1] = cd & M30; cd >>= 30
        ne[i - 1] = ce & M30; ce >>= 30
    nd[8] = cd; ne[8] = ce
    assert -2**31 <= cd < 2**31 and -2**31 <= ce < 2**31
    assert -2 * P < val(nd) < P and -2 * P < val(ne) < P
    return nd, ne

def normalize(d, sign):
    r = list(d)
    if r[8] < 0: r = [r[i] + PL[i] for i in range(9)]
    if sign < 0: r = [-x for x in r]
    for i in range(8):
        r[i + 1] += r[i] >> 30; r[i] &= M30
    if r[8] < 0:
        r = [r[i] + PL[i] for i in range(9)]
        for i in range(8):
            r[i + 1] += r[i] >> 30; r[i] &= M30
    return r

def modinv(x):
    f, g, d, e = list(PL), to30(x), [0] * 9, [1] + [0] * 8
    zeta = -1
    for _ in range(20):
        zeta, t = divsteps_30(zeta, f[0], g[0])
        d, e = update_de(d, e, t)
        f, g = update_fg(f, g, t)
    assert val(g) == 0
    assert x == 0 or abs(val(f)) == 1, val(f)
    r = normalize(d, f[8])
    v = val(r)
    assert 0 <= v < P
    return v

if __name__ == "__main__":
    random.seed(1)
    cases = [0, 1, 2, P - 1, P - 2, (P - 1) // 2, (1 << 253), 3] + [random.randrange(P) for _ in range(300)]
    for x in cases:
        inv = modinv(x)
        assert (x == 0 and inv == 0) or x * inv % P == 1, hex(x)
    print("safegcd model ok on", len(cases), "cases")
    print("P30 =", ", ".join("0x%08x" % l for l in PL))
    print("PINV30 = 0x%08x" % PINV30)
    R = 1 << 256
    print("R3 =", ", ".join("0x%08xu" % ((pow(R, 3, P) >> (32 * i)) & 0xffffffff) for i in range(8)))
