// plk_group.hip -- the G2 side on lane pairs: group law, scalar multiplication, subgroup check, endomorphism, the G2 wire format,
// EIP-197 pair decoding, and Gt * Fr.
#include "plk_common.hpp"

namespace plk {
// ------------------------------------------------------------------ G2 group law on lane pairs -------------------------------
// The complete RCB'15 formulas of bn254_pairing.hpp (proj_double / proj_add, generic over the coordinate field like
// group.rs) instantiated over the lane-pair Fp2 on the carry-free core: a projective G2 point is 27 VGPRs per lane.
// Class invariant and value bounds as for OpsF29 (bn254_pairing.hpp): coordinates N-class, additions carry-normalised,
// products |V| < 2 VaVb/169 + 1; with inputs |V| <= 7 proj_double returns |V| <= 2.3, proj_add (inputs <= 2.3) <= 2.3.
struct OpsW2 {
  typedef W2 F;
  static BN_DEV F add(const F& a, const F& b) { return w2_norm(w2_add(a, b)); }
  static BN_DEV F sub(const F& a, const F& b) { return w2_norm(w2_sub(a, b)); }
  static BN_DEV F neg(const F& a) { return w2_norm(w2_neg(a)); }
  static BN_DEV F mul(const F& a, const F& b) { return w2_mul_ilp(a, b); }       // two-accumulator leaf: 5 % faster here (bn254_f29.hpp)
  static BN_DEV F sqr(const F& a) { return w2_sqr(a); }                          // coordinates are N-class: non-negative limbs
  static BN_DEV F zero() { return W2{OpsF29::zero()}; }
  static BN_DEV F one() { return W2{sel9(lane_odd(), OpsF29::one(), OpsF29::zero())}; }
  static BN_DEV bool is_zero(const F& a) { return s2_is_zero(w2_to_s2(a)); }
  static BN_DEV F select(const F& a, const F& b, bool c) { return w2_select(a, b, c); }
  // 3 b' as R-class lane-pair digits (3 b' 2^261 mod p, balanced), one select per limb where it is used -- see w2_twist_b()
  static BN_DEV F mul_b3(const F& a) {
    const F29 k0{{0x10dfc87a, 0x0066b592, 0x16ad88c7, 0x02c15844, 0x158f2f8c, 0x09ad0e4d, 0x137cf714, 0x14872cdb, -1389659}};
    const F29 k1{{0x01e5cc12, 0x15bda508, 0x18588eb7, 0x00f3e938, 0x18f2b0b4, 0x03bffebe, 0x13752c37, 0x0ec49a37, 0x0017dd10}};
    return w2_mul_ilp(a, W2{sel9(lane_odd(), k0, k1)});
  }
  // the lazy linear layer of proj_add_lazy / proj_double_lazy, each lane on its own coordinate
  static BN_DEV F ladd(const F& a, const F& b) { return w2_add(a, b); }
  static BN_DEV F lsub(const F& a, const F& b) { return w2_sub(a, b); }
  static BN_DEV F norm(const F& a) { return w2_norm(a); }
  static BN_DEV F norm_x8(const F& a) { return W2{f29_norm_x8(a.c)}; }
  static BN_DEV F norm_sub3(const F& a, const F& b) { return W2{f29_norm_sub3(a.c, b.c)}; }
  static BN_DEV F mul_b3_lazy(const F& a) { return mul_b3(w2_norm(a)); }   // a product leaf: its operand must be N-class
};
typedef Proj<W2> G2Q;
BN_NOINLINE void g2q_double(G2Q& r, const G2Q& p) { r = proj_double_lazy<OpsW2>(p); }
BN_NOINLINE void g2q_add(G2Q& r, const G2Q& p, const G2Q& q) { r = proj_add_lazy<OpsW2>(p, q); }
// ---- the isomorphic twist of the multi-step routines --------------------------------------------------------------------------
// In the complete formulas every addition multiplies twice and every doubling once by 3 b', and b' = 3 / (9 + u) is a generic Fp2 element: a
// full product leaf each time (13 % of the leaves of a scalar multiplication).  The map phi(x, y) = (s^2 x, s^3 y) with s in Fp,
// s^6 = 82 / 3, is a group isomorphism from E': y^2 = x^3 + b' onto E'': y^2 = x^3 + b' s^6 = x^3 + (9 - u), where 3 b'' = 27 - 3 u and
// the multiplication is ONE two-term reduce pass (27 own -+ 3 partner's coordinate: 60 instructions against 345).  Scalar multiplications, the
// fixed-base table and the subgroup relation are computed on E'' -- phi on the way in (two Fp scalings), phi^-1 on the way out -- and are
// the same group elements / the same boolean: phi commutes with the group law, and with psi because s is in Fp (conj s = s, so
// phi psi phi^-1 = psi coordinate for coordinate).  Single additions and doublings stay on E' (OpsW2): the map would cost more than it saves.
// tests/test_device_constants.py checks s^6 = 82 / 3 and the four scaling constants; parity of every routine is against the oracle on E'.
struct OpsW2I : OpsW2 {
  static BN_DEV F mul_b3(const F& a) { return w2_mul_27m3u(a); }      // bn254_pair29.hpp: (27 a0 + 3 a1) + (27 a1 - 3 a0) u, R-class
  static BN_DEV F mul_b3_lazy(const F& a) { return mul_b3(a); }         // 64-bit terms: limbs up to 2^30 in magnitude need no carry pass first
};
BN_DEV G2Q g2q_to_iso(const G2Q& p) { return G2Q{w2_scale(p.x, f29_iso_s2()), w2_scale(p.y, f29_iso_s3()), p.z}; }      // (X : Y : Z) -> (s^2 X : s^3 Y : Z)
BN_DEV G2Q g2q_from_iso(const G2Q& p) {                                 // (X : Y : Z) -> (s^-2 X : s^-3 Y : Z)
  const F29 s2i{{0x0a58afee, 0x062df742, 0x0d946d23, 0x0efd68f7, 0x04fb80f1, 0x185fd309, 0x1b649eb6, 0x008a56c1, -689288}};
  const F29 s3i{{0x01ede983, 0x09ec02f0, 0x0d7df454, 0x15cd1e1c, 0x11f38b74, 0x1a89d98e, 0x1671744d, 0x0f15cb65, 0x0006a386}};
  return G2Q{w2_scale(p.x, s2i), w2_scale(p.y, s3i), p.z};
}
BN_NOINLINE void g2qi_double(G2Q& r, const G2Q& p) { r = proj_double_lazy<OpsW2I>(p); }
BN_NOINLINE void g2qi_add(G2Q& r, const G2Q& p, const G2Q& q) { r = proj_add_lazy<OpsW2I>(p, q); }
// `region`: this lane's G1_TABLE_BYTES_PER_LANE bytes of a leased global block for the window table (NULL: the stack frame) -- a lane of
// the pair holds its own coordinate of every entry, 27 words like a G1 point
// points in and out on E'' (g2q_to_iso / g2q_from_iso at the callers)
BN_NOINLINE void g2q_scalar_mul(G2Q& out, const G2Q& p, const u32 (&k)[8], int nwin = 64, void* region = nullptr) {
  auto dbl = [](const G2Q& a) { G2Q r; g2qi_double(r, a); return r; };
  auto add = [](const G2Q& a, const G2Q& b) { G2Q r; g2qi_add(r, a, b); return r; };
  auto dbl_loop = [](const G2Q& a) { return proj_double_lazy<OpsW2I>(a); };
  auto add_loop = [](const G2Q& a, const G2Q& b) { return proj_add_lazy<OpsW2I>(a, b); };
  if (region) {
    ProjTableGlobal<G2Q> tab{(ProjTableGlobal<G2Q>::gptr)region};
    out = scalar_mul_window<OpsW2I>(p, k, dbl, add, nwin, dbl_loop, add_loop, tab);
  } else {
    out = scalar_mul_window<OpsW2I>(p, k, dbl, add, nwin, dbl_loop, add_loop);
  }
}
// k * Q for Q in the r-torsion (G2 proper): the 4-dimensional GLS split of bn254_pairing.hpp (gls4_decompose) -- four 64-bit
// sub-scalars against Q, psi Q, psi^2 Q, psi^3 Q on one shared window schedule: 17 windows of 4 doublings and 4 complete additions
// (68 + 68 group operations instead of 256 + 64), ONE table of 1Q..8Q; the psi^i image of a table entry is taken when it is used:
//   psi (X:Y:Z) = (e0 conj X : e1 conj Y : conj Z),  psi^2 = (beta X : -Y : Z) with beta in Fp,  psi^3 = (e3 conj X : -e1 conj Y : conj Z)
// (conj is a field automorphism, so the maps act on projective coordinates; the minus signs fold into the digit's sign).
// Only valid on the r-torsion: elsewhere psi is not multiplication by lam -- callers with arbitrary twist points use g2q_scalar_mul.
// Points in and out on E'' (psi has the same coordinate form there).
template <class TAB>
BN_DEV void g2q_scalar_mul_gls_t(G2Q& out, const G2Q& p, const u32 (&k)[8], TAB& tab) {
  u32 m[4][2];
  bool neg[4];
  gls4_decompose(m, neg, k);
  signed char dig[4][17];
#pragma unroll 1
  for (int i = 0; i < 4; ++i) gls4_digits(dig[i], m[i]);
  {
    G2Q t1 = p, t2, t3, t4, t;
    // an identity handed over as (x : y : 0) becomes the canonical (0 : 1 : 0) (see g1_scalar_mul)
    const bool pinf = OpsW2::is_zero(t1.z);
    t1.x = OpsW2::select(t1.x, OpsW2::zero(), pinf);
    t1.y = OpsW2::select(t1.y, OpsW2::one(), pinf);
    t1.z = OpsW2::select(t1.z, OpsW2::zero(), pinf);
    tab.put(0, proj_zero<OpsW2>());
    tab.put(1, t1);
    g2qi_double(t2, t1); tab.put(2, t2);
    g2qi_add(t3, t2, t1); tab.put(3, t3);
    g2qi_double(t4, t2); tab.put(4, t4);
    g2qi_add(t, t4, t1); tab.put(5, t);
    g2qi_double(t, t3); tab.put(6, t);
    g2qi_add(t, t, t1); tab.put(7, t);
    g2qi_double(t, t4); tab.put(8, t);
  }
  // R-class lane-pair digits of the constants (value 2^261 mod p, balanced): e0 = xi^((p-1)/3), e1 = xi^((p-1)/2), e3 = e0 conj(e0 conj e0), beta = e0 conj e0
  const F29 e0a{{0x0c289449, 0x05f0a422, 0x0f85cd6c, 0x144ada8b, 0x053ef805, 0x01e2f615, 0x0b280ae6, 0x0c277edf, -774555}};
  const F29 e0b{{0x11142ef1, 0x0b31acc7, 0x1d5818bc, 0x180afc17, 0x1a63177e, 0x15765b3b, 0x118f742e, 0x063a509a, 0x00135e4e}};
  const F29 e1a{{0x02a20931, 0x026f9a50, 0x16a46c6e, 0x158851e6, 0x0cbb39a7, 0x14064374, 0x1e4cd58d, 0x139448ce, -1252754}};
  const F29 e1b{{0x19a647d5, 0x19fdefab, 0x1d925d1a, 0x0d1f6c5f, 0x08ac6cc5, 0x1fa5621a, 0x134f06fe, 0x09a72816, 0x0015871d}};
  const F29 e3a{{0x136caecd, 0x19c70818, 0x1dae30d1, 0x028eb786, 0x0bee8f49, 0x1a51d4be, 0x135c7d00, 0x11fdec39, 0x000cad5f}};
  const F29 e3b{{0x0e67888f, 0x1909bbf8, 0x1437ee3c, 0x018c6b33, 0x1225801f, 0x14b58183, 0x0624ca44, 0x108b53c5, -654776}};
  const F29 beta{{0x18ccb791, 0x175b1c3a, 0x0b83d6e2, 0x0e8ed071, 0x1282bee2, 0x04220e84, 0x1fe4017f, 0x15084d4a, 0x00169119}};
  const bool odd = lane_odd();
  G2Q res = proj_zero<OpsW2>();
#pragma unroll 1
  for (int w = 16; w >= 0; --w) {
    if (w != 16) {
#pragma unroll 1
      for (int j = 0; j < 4; ++j) res = proj_double_lazy<OpsW2I>(res);      // inlined in the loop: no point travels through the stack frame
    }
#pragma unroll 1
    for (int i = 0; i < 4; ++i) {
      const int d = dig[i][w], mag = d < 0 ? -d : d;
      G2Q q = tab.get(mag);
      bool flip = (d < 0) != neg[i];
      if (i == 2) {
        q.x = w2_scale(q.x, beta);
        flip = !flip;
      } else if (i != 0) {
        q.x = w2_mul(W2{sel9(odd, i == 1 ? e0a : e3a, i == 1 ? e0b : e3b)}, w2_conj(q.x));
        q.y = w2_mul(W2{sel9(odd, e1a, e1b)}, w2_conj(q.y));
        q.z = w2_conj(q.z);
        if (i == 3) flip = !flip;
      }
      q.y = OpsW2::select(q.y, OpsW2::neg(q.y), flip);
      res = proj_add_lazy<OpsW2I>(res, q);
    }
  }
  out = res;
}
BN_NOINLINE void g2q_scalar_mul_gls(G2Q& out, const G2Q& p, const u32 (&k)[8], void* region = nullptr) {
  if (region) {
    ProjTableGlobal<G2Q> tab{(ProjTableGlobal<G2Q>::gptr)region};
    g2q_scalar_mul_gls_t(out, p, k, tab);
  } else {
    ProjTableLocal<G2Q> tab;
    g2q_scalar_mul_gls_t(out, p, k, tab);
  }
}
// group.rs:475-495 (through the saturated core: one Fp2 inversion)
BN_DEV void g2q_to_affine(S2& x, S2& y, bool& inf, const G2Q& p) {
  const S2 zi = s2_inv(w2_to_s2(p.z));
  inf = s2_is_zero(zi);
  x = s2_select(s2_mul(w2_to_s2(p.x), zi), s2_zero(), inf);
  y = s2_select(s2_mul(w2_to_s2(p.y), zi), s2_one(), inf);
}
BN_DEV bool g2q_on_curve_affine(const S2& x, const S2& y) {      // g2.rs:279-297
  return s2_eq(s2_sub(s2_sqr(y), s2_mul(s2_sqr(x), x)), s2_const(C_TWIST_B));
}
// g2.rs:488-513: (x+1)Q + psi(xQ) + psi^2(xQ) == psi^3(2xQ) for Q on the twist, affine
BN_NOINLINE bool g2q_in_subgroup_proj(const G2Q& q_in) {
  const G2Q q = g2q_to_iso(q_in);                  // the relation is checked on E'': same boolean (phi is a group isomorphism commuting with psi)
  // x Q by the signed-digit chain of the CONSTANT x over the table {17 Q, 35 Q} (the chain of exp_by_neg_z29, bn254_pair29.hpp: the same three
  // masks, tests/test_wnaf_constants.py): 62 doublings + 13 additions in all, every branch wave-uniform, no table in memory.  (The width-3 NAF
  // of the first half of the round: 63 + 18; the general window schedule: 68 + 24 and a nine-entry table in the scratch frame.)
  G2Q q17, q35;
  {
    G2Q t = proj_double_lazy<OpsW2I>(q);
#pragma unroll 1
    for (int j = 0; j < 3; ++j) t = proj_double_lazy<OpsW2I>(t);            // 16 Q
    q17 = proj_add_lazy<OpsW2I>(t, q);
    t = proj_double_lazy<OpsW2I>(q17);
    q35 = proj_add_lazy<OpsW2I>(t, q);
  }
  G2Q a = q35;                                                               // top digit (bit 57) is +35
  constexpr u64 NZ = BN_X_C_NZ, NEG = BN_X_C_NEG, IS17 = BN_X_C_17;
#pragma unroll 1
  for (int i = 56; i >= 0; --i) {
    a = proj_double_lazy<OpsW2I>(a);
    if ((NZ >> i) & 1) {
      G2Q t = ((IS17 >> i) & 1) ? q17 : q35;
      if ((NEG >> i) & 1) t.y = OpsW2::neg(t.y);
      a = proj_add_lazy<OpsW2I>(a, t);
    }
  }
  // psi on projective coordinates: conj is a field automorphism, so psi(X:Y:Z) = (eps0 conj X : eps1 conj Y : conj Z)
  const W2 e0 = w2_const(C_EPS_EXP0), e1 = w2_const(C_EPS_EXP1);
  auto psi = [&](G2Q& r, const G2Q& p) {
    r.x = w2_mul(e0, w2_conj(p.x));
    r.y = w2_mul(e1, w2_conj(p.y));
    r.z = w2_conj(p.z);
  };
  G2Q b, c, l, r;
  psi(b, a);
  g2qi_add(a, a, q);
  psi(c, b);
  g2qi_add(l, c, b);
  g2qi_add(l, l, a);
  psi(r, c);
  g2qi_double(r, r);
  const G2Q nl = proj_neg<OpsW2>(l);
  g2qi_add(r, r, nl);
  return OpsW2::is_zero(r.z);
}
BN_DEV bool g2q_in_subgroup(const S2& x, const S2& y) { return g2q_in_subgroup_proj(G2Q{w2_from_s2(x), w2_from_s2(y), OpsW2::one()}); }
BN_DEV G2Q load_g2q(const u64* xy, const uint8_t* inf, size_t n, size_t i, int odd) {
  // a FLAGGED point is the identity whatever its coordinate words hold: the canonical (0 : 1 : 0), as load_g1_flagged (g1.hip)
  const bool z = inf && inf[i];
  const W2 zero = OpsW2::zero(), one = OpsW2::one();
  return G2Q{z ? zero : w2_from_s2(load_s2(xy, n, i, 0, odd)), z ? one : w2_from_s2(load_s2(xy, n, i, 8, odd)), z ? zero : one};
}
BN_DEV void store_g2q_affine(u64* oxy, uint8_t* oinf, size_t n, size_t i, int odd, const G2Q& r) {      // r on E'
  S2 x, y; bool rinf;
  g2q_to_affine(x, y, rinf, r);
  store_s2(oxy, n, i, 0, odd, x); store_s2(oxy, n, i, 8, odd, y);
  if (!odd) oinf[i] = rinf ? 1 : 0;
}
// ------------------------------------------------------------------ k * G2gen with a fixed-base table ---------------------------
// KeyPair::generate's public half (lib.rs:131-137: G2Projective::generator() * secret_key) for a batch of keys.  The base never
// changes, so k mod r is cut into 32 signed 8-bit digits and the product is 32 complete additions of table entries
// T[w][j] = j 256^w G (j = 1..128, affine ON THE ISOMORPHIC TWIST E'', R-class lane-pair digits) -- no doublings.  The table (32 x 128 x 36 words = 590 KB,
// resident in L2 / MALL) is built once per device by 4096 lane pairs, each with the generic window product.
constexpr int COMB_WIN = 32, COMB_ENT = 128;
constexpr size_t COMB_WORDS = (size_t)COMB_WIN * COMB_ENT * 36;
__global__ void HEAVY_BOUNDS k_g2_comb_table(i32* table) {
  const size_t t = TID, e = pair_index(t);
  const int odd = pair_role(t);
  if (e >= (size_t)COMB_WIN * COMB_ENT) return;
  const int w = (int)(e / COMB_ENT), j = (int)(e % COMB_ENT) + 1;
  u32 k[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
  for (int q = 0; q < 8; ++q) if (q == (w >> 2)) k[q] = (u32)j << (8 * (w & 3));
  const G2Q g = g2q_to_iso(G2Q{w2_from_s2(s2_g2gen_x()), w2_from_s2(s2_g2gen_y()), OpsW2::one()});
  G2Q r;
  g2q_scalar_mul(r, g, k);
  S2 x, y; bool inf;
  g2q_to_affine(x, y, inf, r);                      // never the identity: j 256^w < r.  The table holds affine points of E''
  const W2 wx = w2_from_s2(x), wy = w2_from_s2(y);
  i32* dst = table + e * 36;
#pragma unroll
  for (int q = 0; q < 9; ++q) { dst[odd * 9 + q] = wx.c.v[q]; dst[18 + odd * 9 + q] = wy.c.v[q]; }
}
__global__ void HEAVY_BOUNDS k_g2_generator_mul(const u64* ks, const i32* __restrict__ table, u64* oxy, uint8_t* oinf, size_t n) {
  const size_t t = TID, i = pair_index(t);
  const int odd = pair_role(t);
  if (i >= n) return;
  u32 k[8];
  load_scalar(k, ks, n, i);
  cond_sub_const(k, 0xf0000001u, 0x43e1f593u, 0x79b97091u, 0x2833e848u, 0x8181585du, 0xb85045b6u, 0xe131a029u, 0x30644e72u);   // k mod r (k < p < 2r)
  G2Q res = proj_zero<OpsW2>();
  int carry = 0;
#pragma unroll 1
  for (int w = 0; w < COMB_WIN; ++w) {
    u32 byte = 0;
#pragma unroll
    for (int q = 0; q < 8; ++q) if (q == (w >> 2)) byte = (k[q] >> (8 * (w & 3))) & 255u;
    int d = (int)byte + carry;
    carry = d >= 128;
    d -= carry << 8;                                  // d in [-128, 127]; k < 2^254 leaves no carry out of the last window
    const int mag = d < 0 ? -d : d;
    const i32* src = table + ((size_t)w * COMB_ENT + (size_t)(mag ? mag - 1 : 0)) * 36;
    W2 ex, ey;
#pragma unroll
    for (int q = 0; q < 9; ++q) { ex.c.v[q] = src[odd * 9 + q]; ey.c.v[q] = src[18 + odd * 9 + q]; }
    const bool nz = mag != 0;
    G2Q q2;                                           // digit 0 adds the identity (0 : 1 : 0): the formulas are complete
    q2.x = OpsW2::select(OpsW2::zero(), ex, nz);
    q2.y = OpsW2::select(OpsW2::one(), OpsW2::select(ey, OpsW2::neg(ey), d < 0), nz);
    q2.z = OpsW2::select(OpsW2::zero(), OpsW2::one(), nz);
    res = proj_add_lazy<OpsW2I>(res, q2);         // inlined in the loop: no point travels through the stack frame
  }
  store_g2q_affine(oxy, oinf, n, i, odd, g2q_from_iso(res));
}

// tables: NULL, or 2 n * G1_TABLE_BYTES_PER_LANE bytes (thread t's window table contiguous)
__global__ void HEAVY_BOUNDS k_g2_scalar_mul(const u64* pxy, const uint8_t* pinf, const u64* ks, u64* oxy, uint8_t* oinf, size_t n, uint8_t* tables) {
  const size_t t = TID, i = pair_index(t);
  const int odd = pair_role(t);
  if (i >= n) return;
  u32 k[8];
  load_scalar(k, ks, n, i);
  G2Q r;
  g2q_scalar_mul(r, g2q_to_iso(load_g2q(pxy, pinf, n, i, odd)), k, 64, tables ? tables + t * G1_TABLE_BYTES_PER_LANE : nullptr);
  store_g2q_affine(oxy, oinf, n, i, odd, g2q_from_iso(r));
}
// the same for inputs in the r-torsion (G2Projective values of the reference are: G2Projective::new checks, g2.rs:460-525)
__global__ void HEAVY_BOUNDS k_g2_scalar_mul_gls(const u64* pxy, const uint8_t* pinf, const u64* ks, u64* oxy, uint8_t* oinf, size_t n, uint8_t* tables) {
  const size_t t = TID, i = pair_index(t);
  const int odd = pair_role(t);
  if (i >= n) return;
  u32 k[8];
  load_scalar(k, ks, n, i);
  G2Q r;
  g2q_scalar_mul_gls(r, g2q_to_iso(load_g2q(pxy, pinf, n, i, odd)), k, tables ? tables + t * G1_TABLE_BYTES_PER_LANE : nullptr);
  store_g2q_affine(oxy, oinf, n, i, odd, g2q_from_iso(r));
}
__global__ void HEAVY_BOUNDS k_g2_add(const u64* axy, const uint8_t* ainf, const u64* bxy, const uint8_t* binf, u64* oxy, uint8_t* oinf, size_t n) {
  const size_t t = TID, i = pair_index(t);
  const int odd = pair_role(t);
  if (i >= n) return;
  G2Q r;
  g2q_add(r, load_g2q(axy, ainf, n, i, odd), load_g2q(bxy, binf, n, i, odd));
  store_g2q_affine(oxy, oinf, n, i, odd, r);
}
// Sub for projective points (group.rs:614-624): self + (-other)
__global__ void HEAVY_BOUNDS k_g2_sub(const u64* axy, const uint8_t* ainf, const u64* bxy, const uint8_t* binf, u64* oxy, uint8_t* oinf, size_t n) {
  const size_t t = TID, i = pair_index(t);
  const int odd = pair_role(t);
  if (i >= n) return;
  G2Q r;
  g2q_add(r, load_g2q(axy, ainf, n, i, odd), proj_neg<OpsW2>(load_g2q(bxy, binf, n, i, odd)));
  store_g2q_affine(oxy, oinf, n, i, odd, r);
}
BN_DEV G2Q load_g2q_proj(const u64* pxyz, size_t n, size_t i, int odd) {
  return G2Q{w2_from_s2(load_s2(pxyz, n, i, 0, odd)), w2_from_s2(load_s2(pxyz, n, i, 8, odd)), w2_from_s2(load_s2(pxyz, n, i, 16, odd))};
}
// G2Projective::new([x, y, z]) (g2.rs:460-525): Y^2 Z == X^3 + b' Z^3 or Z == 0, then the subgroup relation on the projective
// point itself.  Off the curve the reference's endomorphism() panics inside the subgroup test (g2.rs:151): NOT_ON_CURVE here.
__global__ void HEAVY_BOUNDS k_g2_projective_new(const u64* pxyz, uint8_t* status, size_t n) {
  const size_t t = TID, i = pair_index(t);
  const int odd = pair_role(t);
  if (i >= n) return;
  const S2 x = load_s2(pxyz, n, i, 0, odd), y = load_s2(pxyz, n, i, 8, odd), z = load_s2(pxyz, n, i, 16, odd);
  const S2 lhs = s2_mul(s2_sqr(y), z);
  const S2 rhs = s2_add(s2_mul(s2_sqr(x), x), s2_mul(s2_mul(s2_sqr(z), z), s2_const(C_TWIST_B)));
  const bool zz = s2_is_zero(z);
  uint8_t st = SYLOW_HIP_ST_OK;
  if (!(s2_eq(lhs, rhs) || zz)) st = SYLOW_HIP_ST_NOT_ON_CURVE;
  else if (zz) {
    // Z == 0 is waved through the curve test whatever X and Y are (g2.rs:469), and the torsion test then runs the complete formulas on
    // (X, Y, 0).  With X == 0 or Y == 0 every intermediate keeps Z == 0 and the verdict is Ok; otherwise x Q comes out as (XY, Y^2, 0)
    // (the last digit of the NAF of x is +1), (x + 1) Q = (XY, Y^2, 0) + (X, Y, 0) has Z = 3 XY X (2 X Y^2) != 0 and the relation
    // fails: NotInSubgroup -- the outcome of the reference's arithmetic, stated directly (the oracle replays it step by step)
    if (!s2_is_zero(x) && !s2_is_zero(y)) st = SYLOW_HIP_ST_NOT_IN_SUBGROUP;
  }
  else if (!g2q_in_subgroup_proj(load_g2q_proj(pxyz, n, i, odd))) st = SYLOW_HIP_ST_NOT_IN_SUBGROUP;
  if (!odd) status[i] = st;
}
// ConstantTimeEq for projective points (group.rs:426-447)
__global__ void HEAVY_BOUNDS k_g2_ct_eq(const u64* a, const u64* b, uint8_t* eq, size_t n) {
  const size_t t = TID, i = pair_index(t);
  const int odd = pair_role(t);
  if (i >= n) return;
  const S2 ax = load_s2(a, n, i, 0, odd), ay = load_s2(a, n, i, 8, odd), az = load_s2(a, n, i, 16, odd);
  const S2 bx = load_s2(b, n, i, 0, odd), by = load_s2(b, n, i, 8, odd), bz = load_s2(b, n, i, 16, odd);
  const bool iz = s2_is_zero(az), yz = s2_is_zero(bz);
  const bool same = s2_eq(s2_mul(ax, bz), s2_mul(bx, az)) && s2_eq(s2_mul(ay, bz), s2_mul(by, az));
  if (!odd) eq[i] = ((iz && yz) || (!iz && !yz && same)) ? 1 : 0;
}
__global__ void HEAVY_BOUNDS k_g2_double(const u64* axy, const uint8_t* ainf, u64* oxy, uint8_t* oinf, size_t n) {
  const size_t t = TID, i = pair_index(t);
  const int odd = pair_role(t);
  if (i >= n) return;
  G2Q r;
  g2q_double(r, load_g2q(axy, ainf, n, i, odd));
  store_g2q_affine(oxy, oinf, n, i, odd, r);
}
__global__ void HEAVY_BOUNDS k_g2_normalize(const u64* pxyz, u64* oxy, uint8_t* oinf, size_t n) {
  const size_t t = TID, i = pair_index(t);
  const int odd = pair_role(t);
  if (i >= n) return;
  const G2Q p{w2_from_s2(load_s2(pxyz, n, i, 0, odd)), w2_from_s2(load_s2(pxyz, n, i, 8, odd)), w2_from_s2(load_s2(pxyz, n, i, 16, odd))};
  store_g2q_affine(oxy, oinf, n, i, odd, p);
}
// G2Affine::endomorphism (g2.rs:140-152): psi(x, y) = (eps0 conj x, eps1 conj y), psi(identity) = identity; status reports the
// on-curve re-check the reference performs on the result (it panics there; here NOT_ON_CURVE)
__global__ void __launch_bounds__(BLOCK) k_g2_psi(const u64* qxy, const uint8_t* qinf, u64* oxy, uint8_t* oinf, uint8_t* status, size_t n) {
  const size_t t = TID, i = pair_index(t);
  const int odd = pair_role(t);
  if (i >= n) return;
  const bool inf = qinf && qinf[i];
  S2 x = load_s2(qxy, n, i, 0, odd), y = load_s2(qxy, n, i, 8, odd), px, py;
  g2_psi_affine(px, py, x, y);
  const bool on = inf || g2q_on_curve_affine(px, py);
  store_s2(oxy, n, i, 0, odd, inf ? x : px);
  store_s2(oxy, n, i, 8, odd, inf ? y : py);
  if (!odd) { oinf[i] = inf ? 1 : 0; if (status) status[i] = on ? SYLOW_HIP_ST_OK : SYLOW_HIP_ST_NOT_ON_CURVE; }
}
// g2.rs:460-525 on an affine input
__global__ void HEAVY_BOUNDS k_g2_subgroup_check(const u64* qxy, const uint8_t* qinf, uint8_t* status, size_t n) {
  const size_t t = TID, i = pair_index(t);
  const int odd = pair_role(t);
  if (i >= n) return;
  uint8_t st = SYLOW_HIP_ST_OK;
  if (!(qinf && qinf[i])) {                       // Z == 0 passes both tests (g2.rs:469,510)
    const S2 x = load_s2(qxy, n, i, 0, odd), y = load_s2(qxy, n, i, 8, odd);
    if (!g2q_on_curve_affine(x, y)) st = SYLOW_HIP_ST_NOT_ON_CURVE;
    else if (!g2q_in_subgroup(x, y)) st = SYLOW_HIP_ST_NOT_IN_SUBGROUP;
  }
  if (!odd) status[i] = st;
}

// ------------------------------------------------------------------ Gt * Fr ---------------------------------------------------
// Mul<&Fr> for &Gt (gt.rs:161-187): the reference walks the 256 signed digits of fp.rs:653-662 MSB first, squaring every step and
// multiplying by g (digit +1) or conj(g) (digit -1).  Fp12 is commutative, so its result is exactly g^(K+) * conj(g)^(K-) with
// K+ / K- the integers formed by the +1 / -1 digits -- for ANY input, unitary or not.  That value is computed here with fixed
// 4-digit windows: a window of a non-adjacent form holds at most two non-zero digits, 21 patterns in all, so the table is
// {1, g^1, g^2, g^4, g^5, g^8, g^9, g^10}, their conjugates, and six mixed entries g^i conj(g)^j (3 squarings + 6 products, the
// rest are conjugations); then 63 x (4 squarings + 1 product), every lane multiplying at every window (no divergence).  256 squarings
// + 63 + 6 products instead of a wave-uniform product at nearly every one of the 256 steps.
// Only one of each conjugate pair of entries is stored (the scratch pool holds two waves per SIMD only up to 4 KB per lane, and
// 21 Fp12 values are 4.5 KB): slot 0 = 1, slots 1..7 = g^{1,2,4,5,8,9,10}, slots 8..10 = g^8 conj(g)^2, g^8 conj(g), g^4 conj(g);
// a pattern with the roles of g and conj(g) swapped reads the same slot and conjugates it (conj is a ring automorphism).
BN_DEV int gt_window_slot(u32 wp, u32 wm, bool& conj) {
  const u64 nib = 0x0000076500430210ull;                           // nibble v = slot of g^v: 1->1 2->2 4->3 5->4 8->5 9->6 10->7
  const u32 sp = (u32)((nib >> (4 * wp)) & 15u), sm = (u32)((nib >> (4 * wm)) & 15u);
  conj = wp < wm;                                                    // the larger part picks the orientation (never equal unless both 0)
  const u32 hi = conj ? wm : wp, lo = conj ? wp : wm;
  int slot = (int)(conj ? sm : sp);
  if (lo != 0) slot = (hi == 8 && lo == 2) ? 8 : (hi == 8 && lo == 1) ? 9 : 10;                       // (4, 1)
  return slot;
}
__global__ void HEAVY_BOUNDS k_gt_pow(const u64* g, const u64* ks, u64* out, size_t n) {
  const size_t t = TID, i = pair_index(t);
  const int odd = pair_role(t);
  const bool active = i < n;
  const size_t ii = active ? i : 0;          // out-of-range lanes recompute element 0 (uniform control flow), store nothing
  W12 tab[11];
  {
    S12 sa, so = s12_one();
    load_s12(sa, g, n, ii, odd);
    w12_from_s12(tab[0], so);
    w12_from_s12(tab[1], sa);
  }
  tab[2] = w12_sqr(tab[1]);                  // g^2
  tab[3] = w12_sqr(tab[2]);                  // g^4
  w12_mul_nl(tab[4], tab[3], tab[1]);        // g^5
  tab[5] = w12_sqr(tab[3]);                  // g^8
  w12_mul_nl(tab[6], tab[5], tab[1]);        // g^9
  w12_mul_nl(tab[7], tab[5], tab[2]);        // g^10
  {
    const W12 c1 = w12_conj(tab[1]), c2 = w12_conj(tab[2]);
    w12_mul_nl(tab[8], tab[5], c2);          // g^8 conj(g)^2
    w12_mul_nl(tab[9], tab[5], c1);          // g^8 conj(g)
    w12_mul_nl(tab[10], tab[3], c1);         // g^4 conj(g)
  }
  // digits of fp.rs:653-662 on the raw 256-bit scalar
  u32 k[8], xh[8], x3[8], np[8], nm[8];
  {
    const Fp kp = load_plain(ks, n, ii, 0);
#pragma unroll
    for (int j = 0; j < 8; ++j) k[j] = kp.v[j];
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) xh[j] = (k[j] >> 1) | (j < 7 ? (k[j + 1] << 31) : 0);
  u64 c = 0;
#pragma unroll
  for (int j = 0; j < 8; ++j) { c += (u64)k[j] + xh[j]; x3[j] = (u32)c; c >>= 32; }
#pragma unroll
  for (int j = 0; j < 8; ++j) { const u32 cc = xh[j] ^ x3[j]; np[j] = x3[j] & cc; nm[j] = xh[j] & cc; }
  auto entry = [&](int w) {                  // the table value of window w: g^(K+ window) conj(g)^(K- window)
    u32 wp = 0, wm = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) if (j == (w >> 3)) { wp = (np[j] >> (4 * (w & 7))) & 15u; wm = (nm[j] >> (4 * (w & 7))) & 15u; }
    bool cj;
    const W12 m = tab[gt_window_slot(wp, wm, cj)];
    const W12 mc = w12_conj(m);
    W12 r = m;
    r.c1.c0 = w2_select(m.c1.c0, mc.c1.c0, cj); r.c1.c1 = w2_select(m.c1.c1, mc.c1.c1, cj); r.c1.c2 = w2_select(m.c1.c2, mc.c1.c2, cj);
    return r;
  };
  // Every Gt the reference can hand out (a pairing value, products and powers of them) lies in the cyclotomic subgroup
  // g^(p^4 - p^2 + 1) = 1, where the Granger-Scott squaring (6 Fp2 products) IS the square (12): checked per element as
  // frob^4(g) g == frob^2(g) (~35 Fp2 products against the ~4 000 of the power), and taken when it holds for the whole wavefront --
  // the products and conjugates the loop forms stay in the subgroup.  Other inputs keep the generic squaring: same value either way.
  bool cyc;
  {
    W12 f2, f4, lhs;
    w12_frobenius_nl<2>(f2, tab[1]);
    w12_frobenius_nl<2>(f4, f2);
    w12_mul_nl(lhs, f4, tab[1]);
    S12 a, b;
    w12_to_s12(a, lhs);
    w12_to_s12(b, f2);
    const bool eq = s2_eq(a.c0.c0, b.c0.c0) && s2_eq(a.c0.c1, b.c0.c1) && s2_eq(a.c0.c2, b.c0.c2) &&
                    s2_eq(a.c1.c0, b.c1.c0) && s2_eq(a.c1.c1, b.c1.c1) && s2_eq(a.c1.c2, b.c1.c2);
    const bool zero = s2_is_zero(b.c0.c0) && s2_is_zero(b.c0.c1) && s2_is_zero(b.c0.c2) && s2_is_zero(b.c1.c0) && s2_is_zero(b.c1.c1) && s2_is_zero(b.c1.c2);
    cyc = wave_max((eq && !zero) ? 0 : 1) == 0;
  }
  W12 res = entry(63);
#pragma unroll 1
  for (int w = 62; w >= 0; --w) {
    if (cyc) {
#pragma unroll 1
      for (int q = 0; q < 4; ++q) w12_cyclotomic_sqr_nl(res, res);
    } else {
#pragma unroll 1
      for (int q = 0; q < 4; ++q) res = w12_sqr(res);
    }
    const W12 m = entry(w);
    w12_mul_nl(res, res, m);
  }
  S12 sr;
  w12_to_s12(sr, res);
  if (active) store_s12(out, n, i, odd, sr);
}
// k_gt_pow for single calls and small batches: ONE element per wavefront (all 32 lane pairs hold it), the squarings and products of the
// same window schedule spread over the wavefront (bn254_pair29.hpp: w12_mul_wide, w12_cyclotomic_sqr_wide), the eleven table entries
// parked in LDS.  Same table, same digits, same sequence of field operations: the same canonical value.
struct alignas(16) GtShelf { i32 v[66][2][12]; };
typedef __attribute__((address_space(3))) GtShelf* GtShelfPtr;
BN_DEV void gt_shelve(GtShelfPtr sh, int k, const W12& v, int odd, bool writer) {
  const W2* const c[6] = {&v.c0.c0, &v.c0.c1, &v.c0.c2, &v.c1.c0, &v.c1.c1, &v.c1.c2};
  if (writer) {
#pragma unroll
    for (int t = 0; t < 6; ++t)
#pragma unroll
      for (int i = 0; i < 9; ++i) sh->v[6 * k + t][odd][i] = c[t]->c.v[i];
  }
  __syncthreads();
}
BN_DEV W12 gt_fetch(GtShelfPtr sh, int k, int odd) {
  W12 r;
  W2* const c[6] = {&r.c0.c0, &r.c0.c1, &r.c0.c2, &r.c1.c0, &r.c1.c1, &r.c1.c2};
#pragma unroll
  for (int t = 0; t < 6; ++t)
#pragma unroll
    for (int i = 0; i < 9; ++i) c[t]->c.v[i] = sh->v[6 * k + t][odd][i];
  return r;
}
__global__ void HEAVY_BOUNDS k_gt_pow_wide(const u64* g, const u64* ks, u64* out, size_t n) {
  __shared__ WideLds lds;
  __shared__ GtShelf shelf;
  const WideLdsPtr x = (WideLdsPtr)&lds;
  const GtShelfPtr sh = (GtShelfPtr)&shelf;
  const size_t i = blockIdx.x;
  const int lane = (int)(threadIdx.x & 63u), odd = pair_role((u32)lane);
  const bool writer = pair_index((u32)lane) == 0;
  auto sqr = [&](const W12& a) { return w12_mul_wide<1, WK_SQUARE>(a, a, x); };
  auto mul = [&](const W12& a, const W12& b) { return w12_mul_wide<1, WK_DENSE>(a, b, x); };
  W12 g1;
  {
    S12 sa, so = s12_one();
    load_s12(sa, g, n, i, odd);
    W12 one;
    w12_from_s12(one, so);
    w12_from_s12(g1, sa);
    gt_shelve(sh, 0, one, odd, writer);
    gt_shelve(sh, 1, g1, odd, writer);
  }
  {   // slots 2 .. 10: g^2, g^4, g^5, g^8, g^9, g^10, g^8 conj(g)^2, g^8 conj(g), g^4 conj(g)   (k_gt_pow's table)
    const W12 g2 = sqr(g1);
    gt_shelve(sh, 2, g2, odd, writer);
    const W12 g4 = sqr(g2);
    gt_shelve(sh, 3, g4, odd, writer);
    gt_shelve(sh, 4, mul(g4, g1), odd, writer);
    const W12 g8 = sqr(g4);
    gt_shelve(sh, 5, g8, odd, writer);
    gt_shelve(sh, 6, mul(g8, g1), odd, writer);
    gt_shelve(sh, 7, mul(g8, g2), odd, writer);
    gt_shelve(sh, 8, mul(g8, w12_conj(g2)), odd, writer);
    gt_shelve(sh, 9, mul(g8, w12_conj(g1)), odd, writer);
    gt_shelve(sh, 10, mul(g4, w12_conj(g1)), odd, writer);
  }
  // digits of fp.rs:653-662 on the raw 256-bit scalar (as k_gt_pow)
  u32 k[8], xh[8], x3[8], np[8], nm[8];
  {
    const Fp kp = load_plain(ks, n, i, 0);
#pragma unroll
    for (int j = 0; j < 8; ++j) k[j] = kp.v[j];
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) xh[j] = (k[j] >> 1) | (j < 7 ? (k[j + 1] << 31) : 0);
  u64 c = 0;
#pragma unroll
  for (int j = 0; j < 8; ++j) { c += (u64)k[j] + xh[j]; x3[j] = (u32)c; c >>= 32; }
#pragma unroll
  for (int j = 0; j < 8; ++j) { const u32 cc = xh[j] ^ x3[j]; np[j] = x3[j] & cc; nm[j] = xh[j] & cc; }
  auto entry = [&](int w) {
    u32 wp = 0, wm = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) if (j == (w >> 3)) { wp = (np[j] >> (4 * (w & 7))) & 15u; wm = (nm[j] >> (4 * (w & 7))) & 15u; }
    bool cj;
    const int slot = __builtin_amdgcn_readfirstlane(gt_window_slot(wp, wm, cj));      // the whole wavefront holds one scalar
    const W12 m = gt_fetch(sh, slot, odd);
    const W12 mc = w12_conj(m);
    W12 r = m;
    r.c1.c0 = w2_select(m.c1.c0, mc.c1.c0, cj); r.c1.c1 = w2_select(m.c1.c1, mc.c1.c1, cj); r.c1.c2 = w2_select(m.c1.c2, mc.c1.c2, cj);
    return r;
  };
  // cyclotomic inputs take the Granger-Scott squaring (see k_gt_pow): frob^4(g) g == frob^2(g)
  bool cyc;
  {
    const W12 f2 = w12_frobenius_wide<2, 1>(g1, x);
    const W12 f4 = w12_frobenius_wide<2, 1>(f2, x);
    const W12 lhs = mul(f4, g1);
    S12 a, b;
    w12_to_s12(a, lhs);
    w12_to_s12(b, f2);
    const bool eq = s2_eq(a.c0.c0, b.c0.c0) && s2_eq(a.c0.c1, b.c0.c1) && s2_eq(a.c0.c2, b.c0.c2) &&
                    s2_eq(a.c1.c0, b.c1.c0) && s2_eq(a.c1.c1, b.c1.c1) && s2_eq(a.c1.c2, b.c1.c2);
    const bool zero = s2_is_zero(b.c0.c0) && s2_is_zero(b.c0.c1) && s2_is_zero(b.c0.c2) && s2_is_zero(b.c1.c0) && s2_is_zero(b.c1.c1) && s2_is_zero(b.c1.c2);
    cyc = wave_max((eq && !zero) ? 0 : 1) == 0;
  }
  W12 res = entry(63);
#pragma unroll 1
  for (int w = 62; w >= 0; --w) {
    if (cyc) {
#pragma unroll 1
      for (int q = 0; q < 4; ++q) res = w12_cyclotomic_sqr_wide<1>(res, x);
    } else {
#pragma unroll 1
      for (int q = 0; q < 4; ++q) res = sqr(res);
    }
    res = mul(res, entry(w));
  }
  S12 sr;
  w12_to_s12(sr, res);
  if (writer) store_s12(out, n, i, odd, sr);
}
}  // namespace plk

// one LANE PAIR per 192-byte pair: decode + validate into the SoA arrays the multi-pairing kernel consumes.  Both lanes decode
// the six field elements; the G2 checks (twist equation, subgroup) run on the lane-pair Fp2 (g2q_* above).
__global__ void HEAVY_BOUNDS k_evm_decode_pairs(const uint8_t* in, size_t n_pairs, u64* pxy, uint8_t* pinf, u64* qxy, uint8_t* qinf, uint8_t* pst) {
  const size_t t = TID, i = bn254::pl::pair_index(t);
  const bool odd = bn254::pl::pair_role(t) != 0;
  if (i >= n_pairs) return;
  const uint8_t* b = in + 192 * i;
  Fp f[6];
  bool ok = true;
#pragma unroll
  for (int k = 0; k < 6; ++k) ok = read_be_fp(f[k], b + 32 * k) && ok;
  uint8_t st = SYLOW_HIP_ST_OK;
  bool ainf = true, binf = true;
  if (!ok) {
    st = SYLOW_HIP_ST_DECODE_ERROR;
  } else {
    ainf = fp_is_zero(f[0]) && fp_is_zero(f[1]);
    if (!ainf && !g1_on_curve_affine(fp_to_mont(f[0]), fp_to_mont(f[1]))) st = SYLOW_HIP_ST_NOT_ON_CURVE;
    binf = fp_is_zero(f[2]) && fp_is_zero(f[3]) && fp_is_zero(f[4]) && fp_is_zero(f[5]);
    if (!st && !binf) {
      // (bax, bay), (bbx, bby): x = f[3] + f[2] u, y = f[5] + f[4] u; this lane's coordinate
      const pl::S2 x{fp_to_mont(pl::sel(odd, f[3], f[2]))}, y{fp_to_mont(pl::sel(odd, f[5], f[4]))};
      if (!plk::g2q_on_curve_affine(x, y)) st = SYLOW_HIP_ST_NOT_ON_CURVE;
      else if (!plk::g2q_in_subgroup(x, y)) st = SYLOW_HIP_ST_NOT_IN_SUBGROUP;
    }
  }
  if (odd) return;
  bool dead = st != SYLOW_HIP_ST_OK;
  Fp zero = fp_zero(), one = fp_from_limbs(1, 0, 0, 0, 0, 0, 0, 0);
  // identities (and invalid pairs, whose job is rejected anyway) are stored in the canonical (0, 1) encoding
  store_plain(pxy, n_pairs, i, 0, (ainf || dead) ? zero : f[0]);
  store_plain(pxy, n_pairs, i, 4, (ainf || dead) ? one : f[1]);
  store_plain(qxy, n_pairs, i, 0, (binf || dead) ? zero : f[3]);
  store_plain(qxy, n_pairs, i, 4, (binf || dead) ? zero : f[2]);
  store_plain(qxy, n_pairs, i, 8, (binf || dead) ? one : f[5]);
  store_plain(qxy, n_pairs, i, 12, (binf || dead) ? zero : f[4]);
  pinf[i] = (ainf || dead) ? 1 : 0;
  qinf[i] = (binf || dead) ? 1 : 0;
  pst[i] = st;
}

__global__ void __launch_bounds__(BLOCK) k_g2_to_bytes(const u64* xy, const uint8_t* inf, uint8_t* out, size_t n) {
  size_t i = TID;
  if (i >= n) return;
  bool z = inf && inf[i];
  const Fp zero = fp_zero(), one = fp_from_limbs(1, 0, 0, 0, 0, 0, 0, 0);
  Fp xc0 = z ? zero : fp_reduce_plain(load_plain(xy, n, i, 0)), xc1 = z ? zero : fp_reduce_plain(load_plain(xy, n, i, 4));
  Fp yc0 = z ? one : fp_reduce_plain(load_plain(xy, n, i, 8)), yc1 = z ? zero : fp_reduce_plain(load_plain(xy, n, i, 12));
  uint8_t* o = out + 128 * i;
  write_be_fp(o, xc1); write_be_fp(o + 32, xc0); write_be_fp(o + 64, yc1); write_be_fp(o + 96, yc0);
  if (z) o[0] |= 0x80;
}
// one LANE PAIR per 128-byte encoding (both lanes decode, the curve / subgroup checks run on the lane-pair Fp2)
__global__ void HEAVY_BOUNDS k_g2_from_bytes(const uint8_t* in, u64* xy, uint8_t* inf, uint8_t* status, size_t n) {
  const size_t t = TID, i = bn254::pl::pair_index(t);
  const bool odd = bn254::pl::pair_role(t) != 0;
  if (i >= n) return;
  const uint8_t* b = in + 128 * i;
  const bool flag = (b[0] >> 7) & 1;
  Fp xc1, xc0, yc1, yc0;
  bool ok = read_be_fp(xc1, b, true);
  ok = read_be_fp(xc0, b + 32) && ok;
  ok = read_be_fp(yc1, b + 64) && ok;
  ok = read_be_fp(yc0, b + 96) && ok;
  const Fp zero = fp_zero(), one = fp_from_limbs(1, 0, 0, 0, 0, 0, 0, 0);
  uint8_t st = SYLOW_HIP_ST_OK;
  bool is01 = fp_is_zero(xc0) && fp_is_zero(xc1) && fp_eq(yc0, one) && fp_is_zero(yc1);
  if (!ok) st = SYLOW_HIP_ST_DECODE_ERROR;
  else if (flag) st = is01 ? SYLOW_HIP_ST_OK : SYLOW_HIP_ST_DECODE_ERROR;
  else {
    const pl::S2 x{fp_to_mont(pl::sel(odd, xc0, xc1))}, y{fp_to_mont(pl::sel(odd, yc0, yc1))};
    if (!plk::g2q_on_curve_affine(x, y)) st = SYLOW_HIP_ST_NOT_ON_CURVE;
    else if (!plk::g2q_in_subgroup(x, y)) st = SYLOW_HIP_ST_NOT_IN_SUBGROUP;
  }
  if (odd) return;
  bool z = flag || st != SYLOW_HIP_ST_OK;
  store_plain(xy, n, i, 0, z ? zero : xc0); store_plain(xy, n, i, 4, z ? zero : xc1);
  store_plain(xy, n, i, 8, z ? one : yc0); store_plain(xy, n, i, 12, z ? zero : yc1);
  inf[i] = z ? 1 : 0;
  status[i] = st;
}

namespace plkh {
size_t g2_comb_bytes() { return plk::COMB_WORDS * sizeof(bn254::i32); }
int32_t build_g2_comb(bn254::i32* table, void* stream) {
  plk::k_g2_comb_table<<<GRID(2 * (size_t)plk::COMB_WIN * plk::COMB_ENT)>>>(table); LAUNCHED();
}
int32_t evm_decode_pairs(const uint8_t* in, size_t n_pairs, uint64_t* pxy, uint8_t* pinf, uint64_t* qxy, uint8_t* qinf, uint8_t* pst, void* stream) {
  k_evm_decode_pairs<<<dim3((unsigned)((2 * n_pairs + BLOCK - 1) / BLOCK)), dim3(BLOCK), 0, (hipStream_t)stream>>>(in, n_pairs, pxy, pinf, qxy, qinf, pst);
  LAUNCHED();
}
}  // namespace plkh

// window tables of `threads` lanes in a leased global block, one contiguous KB per lane (bn254_pairing.hpp: ProjTableGlobal);
// a failed lease keeps them in the stack frame (NULL)
static uint8_t* plkh_window_tables(host::Lease& ws, size_t threads, void* stream) {
  if (ws.acquire(threads * G1_TABLE_BYTES_PER_LANE, (hipStream_t)stream) == SYLOW_HIP_OK) return (uint8_t*)ws.p;
  (void)hipGetLastError();
  return nullptr;
}

extern "C" {
int32_t sylow_hip_g2_scalar_mul_batch(const uint64_t* p_xy, const uint8_t* p_inf, const uint64_t* k, uint64_t* out_xy, uint8_t* out_inf, size_t n, void* stream) {
  ARGCHK(p_xy && k && out_xy && out_inf); if (!n) return SYLOW_HIP_OK;
  host::Lease ws;
  uint8_t* tables = plkh_window_tables(ws, 2 * n, stream);
  plk::k_g2_scalar_mul<<<GRID(2 * n)>>>(p_xy, p_inf, k, out_xy, out_inf, n, tables);
  const hipError_t e = hipGetLastError();
  const int32_t rc = ws.release();
  return e != hipSuccess ? host::fail(e, "kernel launch") : rc;
}
int32_t sylow_hip_g2_scalar_mul_subgroup_batch(const uint64_t* p_xy, const uint8_t* p_inf, const uint64_t* k, uint64_t* out_xy, uint8_t* out_inf, size_t n, void* stream) {
  ARGCHK(p_xy && k && out_xy && out_inf); if (!n) return SYLOW_HIP_OK;
  host::Lease ws;
  uint8_t* tables = plkh_window_tables(ws, 2 * n, stream);
  plk::k_g2_scalar_mul_gls<<<GRID(2 * n)>>>(p_xy, p_inf, k, out_xy, out_inf, n, tables);
  const hipError_t e = hipGetLastError();
  const int32_t rc = ws.release();
  return e != hipSuccess ? host::fail(e, "kernel launch") : rc;
}
int32_t sylow_hip_g2_generator_mul_batch(const uint64_t* k, uint64_t* out_xy, uint8_t* out_inf, size_t n, void* stream) {
  ARGCHK(k && out_xy && out_inf); if (!n) return SYLOW_HIP_OK;
  const bn254::i32* table = nullptr;
  int32_t rc = host::g2_gen_comb(&table, (hipStream_t)stream);
  if (rc != SYLOW_HIP_OK) return rc;
  plk::k_g2_generator_mul<<<GRID(2 * n)>>>(k, table, out_xy, out_inf, n); LAUNCHED();
}
int32_t sylow_hip_g2_normalize_batch(const uint64_t* p_xyz, uint64_t* out_xy, uint8_t* out_inf, size_t n, void* stream) {
  ARGCHK(p_xyz && out_xy && out_inf); if (!n) return SYLOW_HIP_OK;
  plk::k_g2_normalize<<<GRID(2 * n)>>>(p_xyz, out_xy, out_inf, n); LAUNCHED();
}
int32_t sylow_hip_g2_psi_batch(const uint64_t* q_xy, const uint8_t* q_inf, uint64_t* out_xy, uint8_t* out_inf, uint8_t* status, size_t n, void* stream) {
  ARGCHK(q_xy && out_xy && out_inf); if (!n) return SYLOW_HIP_OK;
  plk::k_g2_psi<<<GRID(2 * n)>>>(q_xy, q_inf, out_xy, out_inf, status, n); LAUNCHED();
}
int32_t sylow_hip_g2_subgroup_check_batch(const uint64_t* q_xy, const uint8_t* q_inf, uint8_t* status, size_t n, void* stream) {
  ARGCHK(q_xy && status); if (!n) return SYLOW_HIP_OK;
  plk::k_g2_subgroup_check<<<GRID(2 * n)>>>(q_xy, q_inf, status, n); LAUNCHED();
}
int32_t sylow_hip_g2_to_be_bytes_batch(const uint64_t* p_xy, const uint8_t* p_inf, uint8_t* out, size_t n, void* stream) {
  ARGCHK(p_xy && out); if (!n) return SYLOW_HIP_OK; k_g2_to_bytes<<<GRID(n)>>>(p_xy, p_inf, out, n); LAUNCHED();
}
int32_t sylow_hip_g2_from_be_bytes_batch(const uint8_t* in, uint64_t* out_xy, uint8_t* out_inf, uint8_t* status, size_t n, void* stream) {
  ARGCHK(in && out_xy && out_inf && status); if (!n) return SYLOW_HIP_OK; k_g2_from_bytes<<<GRID(2 * n)>>>(in, out_xy, out_inf, status, n); LAUNCHED();
}
int32_t sylow_hip_gt_pow_batch(const uint64_t* gt, const uint64_t* k, uint64_t* out, size_t n, void* stream) {
  ARGCHK(gt && k && out); if (!n) return SYLOW_HIP_OK;
  // single calls and small batches: one wavefront per element (one power 3.1 -> ~0.8 ms)
  if (plkh::wide_batch_max() != 0 && n <= 2048) { plk::k_gt_pow_wide<<<dim3((unsigned)n), dim3(64), 0, (hipStream_t)stream>>>(gt, k, out, n); LAUNCHED(); }
  plk::k_gt_pow<<<GRID(2 * n)>>>(gt, k, out, n); LAUNCHED();
}
int32_t sylow_hip_g2_add_batch(const uint64_t* a_xy, const uint8_t* a_inf, const uint64_t* b_xy, const uint8_t* b_inf, uint64_t* out_xy, uint8_t* out_inf, size_t n, void* stream) {
  ARGCHK(a_xy && b_xy && out_xy && out_inf); if (!n) return SYLOW_HIP_OK;
  plk::k_g2_add<<<GRID(2 * n)>>>(a_xy, a_inf, b_xy, b_inf, out_xy, out_inf, n); LAUNCHED();
}
int32_t sylow_hip_g2_sub_batch(const uint64_t* a_xy, const uint8_t* a_inf, const uint64_t* b_xy, const uint8_t* b_inf, uint64_t* out_xy, uint8_t* out_inf, size_t n, void* stream) {
  ARGCHK(a_xy && b_xy && out_xy && out_inf); if (!n) return SYLOW_HIP_OK;
  plk::k_g2_sub<<<GRID(2 * n)>>>(a_xy, a_inf, b_xy, b_inf, out_xy, out_inf, n); LAUNCHED();
}
int32_t sylow_hip_g2_projective_new_batch(const uint64_t* p_xyz, uint8_t* status, size_t n, void* stream) {
  ARGCHK(p_xyz && status); if (!n) return SYLOW_HIP_OK; plk::k_g2_projective_new<<<GRID(2 * n)>>>(p_xyz, status, n); LAUNCHED();
}
int32_t sylow_hip_g2_ct_eq_batch(const uint64_t* a_xyz, const uint64_t* b_xyz, uint8_t* eq, size_t n, void* stream) {
  ARGCHK(a_xyz && b_xyz && eq); if (!n) return SYLOW_HIP_OK; plk::k_g2_ct_eq<<<GRID(2 * n)>>>(a_xyz, b_xyz, eq, n); LAUNCHED();
}
int32_t sylow_hip_g2_double_batch(const uint64_t* a_xy, const uint8_t* a_inf, uint64_t* out_xy, uint8_t* out_inf, size_t n, void* stream) {
  ARGCHK(a_xy && out_xy && out_inf); if (!n) return SYLOW_HIP_OK;
  plk::k_g2_double<<<GRID(2 * n)>>>(a_xy, a_inf, out_xy, out_inf, n); LAUNCHED();
}
}  // extern "C"
