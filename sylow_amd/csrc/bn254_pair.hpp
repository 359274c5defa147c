// bn254_pair.hpp -- the pairing on LANE PAIRS: two lanes of a wavefront carry one element of every Fp2-based type, one its c0
// coordinate and its partner the c1 coordinate ("even" / "odd" lane below; the geometry is defined under "lane-pair geometry").
//
// Why: with one pairing per lane an Fp12 is 96 VGPRs, so Fp12 products cannot keep their operands, result and
// temporaries inside the 256-register budget of a 2-waves-per-SIMD kernel; the one-pairing-per-lane kernels of round 1 therefore moved
// ~0.9 KB of scratch per Fp12 operation through L2/HBM (0.93 TB per 2^20 pairings, profiles/r01_pairing_v5), which is
// what bounds it.  Splitting every Fp2 across two lanes halves the per-lane state (Fp12 = 48 VGPRs, the whole Miller
// state ~150): the tower lives in registers, and the only memory traffic left is the algorithmic input / output.
//
// The price is one DPP exchange (row_half_mirror: lane l <-> lane 7 - l) per Fp2 product: (a0 + a1 u)(b0 + b1 u) = (a0 b0 - a1 b1) + (a1 b0 + a0 b1) u,
// so each lane needs its partner's coordinates of both operands (16 v_mov_dpp) and computes ONE fused two-product
// Montgomery pass (fp_dot2_inline) -- the same arithmetic a one-element-per-lane kernel spends per coordinate.  Additions,
// subtractions, halvings and multiplications by Fp scalars are lane-local.
//
// Formulas, digit schedule and line functions are the reference's (each routine cites the lines it replays), so raw Miller values
// and Gt results are bit-identical to the oracle.
#pragma once
#include "bn254_tower.hpp"

namespace bn254 {
namespace pl {

struct S2 { Fp c; };              // this lane's coordinate of an Fp2 element
struct S6 { S2 c0, c1, c2; };
struct S12 { S6 c0, c1; };

// ---- lane-pair geometry --------------------------------------------------------------------------------------------------
// Within every group of 8 lanes, lanes 0..3 carry the c0 coordinates of four elements and lanes 7..4 the c1 coordinates of the same
// four (lane l and lane 7 - l are partners: DPP row_half_mirror).  The role of a lane is therefore a property of its DPP BANK (four
// contiguous lanes): banks 0 and 2 of a row hold c0, banks 1 and 3 hold c1 -- which is what lets a single DPP instruction act on one
// role only (bank_mask 0x5 = the c0 lanes, 0xA = the c1 lanes; with adjacent-lane pairs a role is a lane PARITY, which no DPP mask
// can select: the operand exchange of the product leaf then needs 36 instructions instead of 27, bn254_pair29.hpp).
// Thread t of a launch handles coordinate pair_role(t) of element pair_index(t); a wavefront carries 32 elements; element order in
// memory is unchanged (each limb plane is still read as 32-byte runs per 4 lanes, 256 contiguous bytes per wavefront).
template <class T> BN_DEV int pair_role(T t) { return (int)((t >> 2) & 1); }
template <class T> BN_DEV T pair_index(T t) { return (T)(((t >> 3) << 2) | ((t & 3) ^ (((t >> 2) & 1) ? 3 : 0))); }
BN_DEV bool lane_odd() { return pair_role(__lane_id()) != 0; }          // "odd" = this lane holds the c1 coordinate
// value held by the partner lane (row_half_mirror: lane l <-> lane 7 - l of its group of 8)
BN_DEV u32 swap_u32(u32 x) { return (u32)__builtin_amdgcn_mov_dpp((int)x, 0x141, 0xF, 0xF, true); }
BN_DEV Fp xchg(const Fp& a) {
  Fp r;
#pragma unroll
  for (int i = 0; i < 8; ++i) r.v[i] = swap_u32(a.v[i]);
  return r;
}
BN_DEV Fp sel(bool odd, const Fp& if_even, const Fp& if_odd) {
  Fp r;
#pragma unroll
  for (int i = 0; i < 8; ++i) r.v[i] = odd ? if_odd.v[i] : if_even.v[i];
  return r;
}

// 9a + w mod p for a canonical, w in [0, p]: the multiply-by-9 pass of fp_mul9_addsub with the sign already applied
BN_DEV Fp fp_mul9_plus(const Fp& a, const Fp& w) {
  const u32 p[8] = {BN_P0, BN_P1, BN_P2, BN_P3, BN_P4, BN_P5, BN_P6, BN_P7};
  u32 t[9];
  u64 acc = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    acc += (u64)a.v[i] * 9u + w.v[i];
    t[i] = (u32)acc;
    acc >>= 32;
  }
  t[8] = (u32)acc;
  u32 h = (t[8] << 6) | (t[7] >> 26);
  u32 q = (h * 677u) >> 13;                 // quotient estimate, tools/check_mulxi_quotient.py
  u32 r[8];
  u64 bor = 0, mp = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    mp += (u64)q * p[i];
    u64 d = (u64)t[i] - (u32)mp - bor;
    r[i] = (u32)d;
    bor = (d >> 32) & 1;
    mp >>= 32;
  }
  return fp_cond_sub_p(r, 0);
}

// ------------------------------------------------------------------ leaves (16 ABI argument registers) ----------
// even lane: a0 b0 + (p - a1) b1;  odd lane: a1 b0 + a0 b1.  The odd lane sends its coordinate already negated.
BN_NOINLINE Fp s2_mul_leaf(Fp a, Fp b) {
  const bool odd = lane_odd();
  const Fp ra = xchg(sel(odd, a, fp_neg_lazy(a)));
  const Fp rb = xchg(b);
  return fp_dot2_inline(a, sel(odd, b, rb), ra, sel(odd, rb, b));
}
// even lane: (a0 + a1)(a0 - a1);  odd lane: a0 * 2 a1   (fp2.rs:164-171); lazily reduced operands < 2p
BN_NOINLINE Fp s2_sqr_leaf(Fp a) {
  const bool odd = lane_odd();
  const Fp o = xchg(a);
  const Fp s = fp_add_lazy(a, o);
  const Fp d = fp_add_lazy(a, fp_neg_lazy(o));
  const Fp t = fp_add_lazy(a, a);
  return fp_mul_inline(sel(odd, s, o), sel(odd, d, t));
}

// ------------------------------------------------------------------ S2 ------------------------------------------
BN_DEV S2 s2_zero() { return S2{fp_zero()}; }
BN_DEV S2 s2_one() { return S2{sel(lane_odd(), fp_one(), fp_zero())}; }
BN_DEV S2 s2_const(const uint32_t (&c)[2][8]) { return S2{sel(lane_odd(), fp_const(c[0]), fp_const(c[1]))}; }
BN_DEV S2 s2_add(const S2& a, const S2& b) { return S2{fp_add(a.c, b.c)}; }
BN_DEV S2 s2_sub(const S2& a, const S2& b) { return S2{fp_sub(a.c, b.c)}; }
BN_DEV S2 s2_neg(const S2& a) { return S2{fp_neg(a.c)}; }
BN_DEV S2 s2_dbl(const S2& a) { return S2{fp_dbl(a.c)}; }
BN_DEV S2 s2_conj(const S2& a) { return S2{sel(lane_odd(), a.c, fp_neg(a.c))}; }
BN_DEV S2 s2_mul(const S2& a, const S2& b) { return S2{s2_mul_leaf(a.c, b.c)}; }
BN_DEV S2 s2_sqr(const S2& a) { return S2{s2_sqr_leaf(a.c)}; }
// x (9 + u): even 9 a0 - a1, odd 9 a1 + a0
BN_DEV S2 s2_mul_xi(const S2& a) {
  const Fp o = xchg(a.c);
  return S2{fp_mul9_plus(a.c, sel(lane_odd(), fp_neg_lazy(o), o))};
}
BN_DEV bool s2_is_zero(const S2& a) {
  u32 z = fp_is_zero(a.c) ? 1u : 0u;
  return (z & swap_u32(z)) != 0;
}
BN_DEV bool s2_eq(const S2& a, const S2& b) {
  u32 z = fp_eq(a.c, b.c) ? 1u : 0u;
  return (z & swap_u32(z)) != 0;
}
BN_DEV S2 s2_select(const S2& a, const S2& b, bool c) { return S2{fp_select(a.c, b.c, c)}; }
// conj / (a0^2 + a1^2), inv(0) = 0 (fp2.rs:355-360); both lanes run the same Fp inversion
BN_DEV S2 s2_inv(const S2& a) {
  const Fp t = fp_mul(a.c, a.c);
  const Fp i = fp_inv(fp_add(t, xchg(t)));
  const Fp r = fp_mul(a.c, i);
  return S2{sel(lane_odd(), r, fp_neg(r))};
}

// ------------------------------------------------------------------ S6 ------------------------------------------
BN_DEV S6 s6_zero() { return S6{s2_zero(), s2_zero(), s2_zero()}; }
BN_DEV S6 s6_one() { return S6{s2_one(), s2_zero(), s2_zero()}; }
BN_DEV S6 s6_add(const S6& a, const S6& b) { return S6{s2_add(a.c0, b.c0), s2_add(a.c1, b.c1), s2_add(a.c2, b.c2)}; }
BN_DEV S6 s6_sub(const S6& a, const S6& b) { return S6{s2_sub(a.c0, b.c0), s2_sub(a.c1, b.c1), s2_sub(a.c2, b.c2)}; }
BN_DEV S6 s6_neg(const S6& a) { return S6{s2_neg(a.c0), s2_neg(a.c1), s2_neg(a.c2)}; }
BN_DEV S6 s6_dbl(const S6& a) { return S6{s2_dbl(a.c0), s2_dbl(a.c1), s2_dbl(a.c2)}; }
BN_DEV S6 s6_mul_v(const S6& a) { return S6{s2_mul_xi(a.c2), a.c0, a.c1}; }
// Karatsuba over v: 6 products (value of fp6.rs:283-367)
BN_DEV S6 s6_mul(const S6& a, const S6& b) {
  S2 v0 = s2_mul(a.c0, b.c0);
  S2 v1 = s2_mul(a.c1, b.c1);
  S2 v2 = s2_mul(a.c2, b.c2);
  S2 t0 = s2_mul(s2_add(a.c1, a.c2), s2_add(b.c1, b.c2));
  S2 t1 = s2_mul(s2_add(a.c0, a.c1), s2_add(b.c0, b.c1));
  S2 t2 = s2_mul(s2_add(a.c0, a.c2), s2_add(b.c0, b.c2));
  S6 r;
  r.c0 = s2_add(v0, s2_mul_xi(s2_sub(s2_sub(t0, v1), v2)));
  r.c1 = s2_add(s2_sub(s2_sub(t1, v0), v1), s2_mul_xi(v2));
  r.c2 = s2_add(s2_sub(s2_sub(t2, v0), v2), v1);
  return r;
}
// CH-SQR2 (fp6.rs:219-236)
BN_DEV S6 s6_sqr(const S6& a) {
  S2 s0 = s2_sqr(a.c0);
  S2 s1 = s2_dbl(s2_mul(a.c0, a.c1));
  S2 s2 = s2_sqr(s2_add(s2_sub(a.c0, a.c1), a.c2));
  S2 s3 = s2_dbl(s2_mul(a.c1, a.c2));
  S2 s4 = s2_sqr(a.c2);
  S6 r;
  r.c0 = s2_add(s0, s2_mul_xi(s3));
  r.c1 = s2_add(s1, s2_mul_xi(s4));
  r.c2 = s2_sub(s2_sub(s2_add(s2_add(s1, s2), s3), s0), s4);
  return r;
}
// fp6.rs:415-423
BN_DEV S6 s6_inv(const S6& a) {
  S2 t0 = s2_sub(s2_sqr(a.c0), s2_mul(a.c1, s2_mul_xi(a.c2)));
  S2 t1 = s2_sub(s2_mul_xi(s2_sqr(a.c2)), s2_mul(a.c0, a.c1));
  S2 t2 = s2_sub(s2_sqr(a.c1), s2_mul(a.c0, a.c2));
  S2 d = s2_add(s2_mul_xi(s2_add(s2_mul(a.c2, t1), s2_mul(a.c1, t2))), s2_mul(a.c0, t0));
  S2 di = s2_inv(d);
  return S6{s2_mul(di, t0), s2_mul(di, t1), s2_mul(di, t2)};
}

// ------------------------------------------------------------------ S12 -----------------------------------------
BN_DEV S12 s12_one() { return S12{s6_one(), s6_zero()}; }
BN_DEV S12 s12_conj(const S12& a) { return S12{a.c0, s6_neg(a.c1)}; }
// fp12.rs:229-238
BN_DEV S12 s12_mul(const S12& a, const S12& b) {
  S6 t0 = s6_mul(a.c0, b.c0);
  S6 t1 = s6_mul(a.c1, b.c1);
  S6 t2 = s6_mul(s6_add(a.c0, a.c1), s6_add(b.c0, b.c1));
  S12 r;
  r.c1 = s6_sub(s6_sub(t2, t0), t1);
  r.c0 = s6_add(s6_mul_v(t1), t0);
  return r;
}
// fp12.rs:536-550
BN_DEV S12 s12_sqr(const S12& a) {
  S6 c0 = s6_sub(a.c0, a.c1);
  S6 c3 = s6_sub(a.c0, s6_mul_v(a.c1));
  S6 c2 = s6_mul(a.c0, a.c1);
  S6 t = s6_add(s6_mul(c0, c3), c2);
  S12 r;
  r.c1 = s6_dbl(c2);
  r.c0 = s6_add(t, s6_mul_v(c2));
  return r;
}
// fp12.rs:281-286
BN_DEV S12 s12_inv(const S12& a) {
  S6 d = s6_sub(s6_sqr(a.c0), s6_mul_v(s6_sqr(a.c1)));
  S6 t = s6_inv(d);
  return S12{s6_mul(a.c0, t), s6_neg(s6_mul(a.c1, t))};
}
BN_DEV bool s12_is_one(const S12& a) {
  bool z = s2_eq(a.c0.c0, s2_one());
  z = z && s2_is_zero(a.c0.c1) && s2_is_zero(a.c0.c2) && s2_is_zero(a.c1.c0) && s2_is_zero(a.c1.c1) && s2_is_zero(a.c1.c2);
  return z;
}

// ------------------------------------------------------------------ Miller loop ---------------------------------
struct G2S { S2 x, y, z; };
// g2.rs:140-152
BN_DEV void g2_psi_affine(S2& xo, S2& yo, const S2& x, const S2& y) {
  xo = s2_mul(s2_const(C_EPS_EXP0), s2_conj(x));
  yo = s2_mul(s2_const(C_EPS_EXP1), s2_conj(y));
}

// ------------------------------------------------------------------ final exponentiation ------------------------
// pairing.rs:274-284
BN_DEV void fp4_square(S2& c0, S2& c1, const S2& a, const S2& b) {
  S2 t0 = s2_sqr(a);
  S2 t1 = s2_sqr(b);
  c0 = s2_add(s2_mul_xi(t1), t0);
  c1 = s2_sub(s2_sub(s2_sqr(s2_add(a, b)), t0), t1);
}
// pairing.rs:309-350
BN_DEV S12 cyclotomic_sqr(const S12& f) {
  S2 z0 = f.c0.c0, z4 = f.c0.c1, z3 = f.c0.c2, z2 = f.c1.c0, z1 = f.c1.c1, z5 = f.c1.c2;
  S2 t0, t1, t2, t3;
  fp4_square(t0, t1, z0, z1);
  z0 = s2_sub(t0, z0); z0 = s2_add(s2_dbl(z0), t0);
  z1 = s2_add(t1, z1); z1 = s2_add(s2_dbl(z1), t1);
  fp4_square(t0, t1, z2, z3);
  fp4_square(t2, t3, z4, z5);
  z4 = s2_sub(t0, z4); z4 = s2_add(s2_dbl(z4), t0);
  z5 = s2_add(t1, z5); z5 = s2_add(s2_dbl(z5), t1);
  t0 = s2_mul_xi(t3);
  z2 = s2_add(t0, z2); z2 = s2_add(s2_dbl(z2), t0);
  z3 = s2_sub(t2, z3); z3 = s2_add(s2_dbl(z3), t2);
  S12 r;
  r.c0.c0 = z0; r.c0.c1 = z4; r.c0.c2 = z3;
  r.c1.c0 = z2; r.c1.c1 = z1; r.c1.c2 = z5;
  return r;
}
// pairing.rs:366-392 with the width-3 signed-digit form of x (see exp_by_neg_z_sat in bn254_pairing.hpp)
BN_NOINLINE void exp_by_neg_z(S12& r, const S12& f) {
  const S12 f3 = s12_mul(cyclotomic_sqr(f), f);
  S12 res = f;
  const u64 nz = 0x4908924444891211ull, ng = 0x0108000400880210ull, th = 0x0108804404880200ull;
#pragma unroll 1
  for (int i = 61; i >= 0; --i) {
    res = cyclotomic_sqr(res);
    if ((nz >> i) & 1) {
      S12 m = ((th >> i) & 1) ? f3 : f;
      if ((ng >> i) & 1) m = s12_conj(m);
      res = s12_mul(res, m);
    }
  }
  r = s12_conj(res);
}

}  // namespace pl
}  // namespace bn254
