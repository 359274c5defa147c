// G1 / G2 group law and the optimal-ate pairing for BN254, one element per lane.
// Batched replacement for sylow's src/groups/group.rs (complete RCB'15 formulas),
// src/groups/g2.rs (psi, subgroup check) and src/pairing.rs (doubling/addition steps, Miller
// loop, final exponentiation).
//
// Bit-exactness contract:
//  * Gt after final exponentiation and affine-normalised points are unique -> any correct
//    algorithm matches the reference bit for bit.
//  * The RAW Miller value is not unique (lines are scaled by subfield factors), so the line
//    formulas and the signed-digit schedule below replay pairing.rs:756-818 and :26-30 exactly;
//    the line coefficients are consumed on the fly instead of being materialised as the
//    reference's [Ell; 87] table (16.8 KB per point, pairing.rs:556).
#pragma once
#include "bn254_f29.hpp"

namespace bn254 {

// ---------------------------------------------------------------- generic projective group ----
// Homogeneous projective (X:Y:Z), identity (0:1:0)  (group.rs:303-316)
template <class F> struct Proj { F x, y, z; };
typedef Proj<Fp> G1P;
typedef Proj<Fp2> G2P;

// field adaptors so the RCB formulas are written once (group.rs is generic over F the same way)
struct OpsFp {
  typedef Fp F;
  static BN_DEV F add(const F& a, const F& b) { return fp_add(a, b); }
  static BN_DEV F sub(const F& a, const F& b) { return fp_sub(a, b); }
  static BN_DEV F neg(const F& a) { return fp_neg(a); }
  static BN_DEV F mul(const F& a, const F& b) { return fp_mul(a, b); }
  static BN_DEV F sqr(const F& a) { return fp_mul(a, a); }
  static BN_DEV F zero() { return fp_zero(); }
  static BN_DEV F one() { return fp_one(); }
  static BN_DEV bool is_zero(const F& a) { return fp_is_zero(a); }
  static BN_DEV F select(const F& a, const F& b, bool c) { return fp_select(a, b, c); }
  // 3*b = 9: x9 by doublings
  static BN_DEV F mul_b3(const F& a) { return fp_add(fp_dbl(fp_dbl(fp_dbl(a))), a); }
};
struct OpsFp2 {
  typedef Fp2 F;
  static BN_DEV F add(const F& a, const F& b) { return fp2_add(a, b); }
  static BN_DEV F sub(const F& a, const F& b) { return fp2_sub(a, b); }
  static BN_DEV F neg(const F& a) { return fp2_neg(a); }
  static BN_DEV F mul(const F& a, const F& b) { return fp2_mul(a, b); }
  static BN_DEV F sqr(const F& a) { return fp2_mul(a, a); }
  static BN_DEV F zero() { return fp2_zero(); }
  static BN_DEV F one() { return fp2_one(); }
  static BN_DEV bool is_zero(const F& a) { return fp2_is_zero(a); }
  static BN_DEV F select(const F& a, const F& b, bool c) { return fp2_select(a, b, c); }
  static BN_DEV F mul_b3(const F& a) { return fp2_mul(a, fp2_const(C_TWIST_B3)); }
};

template <class O> BN_DEV Proj<typename O::F> proj_zero() { return Proj<typename O::F>{O::zero(), O::one(), O::zero()}; }

// group.rs:339-386, RCB'15 algorithm 9 (a = 0)
template <class O>
BN_DEV Proj<typename O::F> proj_double(const Proj<typename O::F>& p) {
  typedef typename O::F F;
  F t0 = O::sqr(p.y);
  F z3 = O::add(t0, t0);
  z3 = O::add(z3, z3);
  z3 = O::add(z3, z3);
  F t1 = O::mul(p.y, p.z);
  F t2 = O::sqr(p.z);
  t2 = O::mul_b3(t2);
  F x3 = O::mul(t2, z3);
  F y3 = O::add(t0, t2);
  z3 = O::mul(t1, z3);
  t1 = O::add(t2, t2);
  t2 = O::add(t1, t2);
  t0 = O::sub(t0, t2);
  y3 = O::mul(t0, y3);
  y3 = O::add(x3, y3);
  t1 = O::mul(p.x, p.y);
  x3 = O::mul(t0, t1);
  x3 = O::add(x3, x3);
  // select zero -> zero (group.rs:377-385); the formulas already map (0:1:0) to (0:y:0), the
  // select pins the canonical (0:1:0)
  bool z = O::is_zero(p.z);
  Proj<F> r;
  r.x = O::select(x3, O::zero(), z);
  r.y = O::select(y3, O::one(), z);
  r.z = O::select(z3, O::zero(), z);
  return r;
}
// group.rs:528-599, RCB'15 algorithm 7 (a = 0), complete
template <class O>
BN_DEV Proj<typename O::F> proj_add(const Proj<typename O::F>& p, const Proj<typename O::F>& q) {
  typedef typename O::F F;
  F t0 = O::mul(p.x, q.x);
  F t1 = O::mul(p.y, q.y);
  F t2 = O::mul(p.z, q.z);
  F t3 = O::mul(O::add(p.x, p.y), O::add(q.x, q.y));
  t3 = O::sub(t3, O::add(t0, t1));
  F t4 = O::mul(O::add(p.y, p.z), O::add(q.y, q.z));
  t4 = O::sub(t4, O::add(t1, t2));
  F y3 = O::sub(O::mul(O::add(p.x, p.z), O::add(q.x, q.z)), O::add(t0, t2));
  F x3 = O::add(t0, t0);
  t0 = O::add(x3, t0);
  t2 = O::mul_b3(t2);
  F z3 = O::add(t1, t2);
  t1 = O::sub(t1, t2);
  y3 = O::mul_b3(y3);
  x3 = O::mul(t4, y3);
  t2 = O::mul(t3, t1);
  x3 = O::sub(t2, x3);
  y3 = O::mul(y3, t0);
  t1 = O::mul(t1, z3);
  y3 = O::add(t1, y3);
  t0 = O::mul(t0, t3);
  z3 = O::mul(z3, t4);
  z3 = O::add(z3, t0);
  return Proj<F>{x3, y3, z3};
}
// The same two formulas for the carry-free policies (OpsF29, OpsW2), linear layer trimmed.  Coordinates are N-class (limbs in [0, 2^29)), so a
// difference of two of them is a legal product operand as it stands (|limbs| < 2^29): the three cross terms of the addition come from
// (a_i - a_j)(b_i - b_j) = t_i + t_j - (a_i b_j + a_j b_i) with no carry pass on the operands, sums of products get ONE carry pass, and values that
// only feed products stay un-normalised where the leaf's bound (|limbs| products summing under 2.5 * 2^58 per column) allows.  Same group element
// as proj_add / proj_double limb for limb after the canonical reduction (tests/test_gpu_group.py, test_gpu_small_rows.py compare affine results).
// Policy extras: lsub / ladd (no carry pass), norm, norm_x8 (8 a), norm_sub3 (a - 3 b), mul_b3_lazy (operand limbs up to 2^30 in magnitude).
template <class O>
BN_DEV Proj<typename O::F> proj_double_lazy(const Proj<typename O::F>& p) {
  typedef typename O::F F;
  F t0 = O::sqr(p.y);
  F z3 = O::norm_x8(t0);
  F t1 = O::mul(p.y, p.z);
  F t2 = O::mul_b3(O::sqr(p.z));
  F x3 = O::mul(t2, z3);
  F y3 = O::norm(O::ladd(t0, t2));
  z3 = O::mul(t1, z3);
  t0 = O::norm_sub3(t0, t2);
  y3 = O::mul(t0, y3);
  y3 = O::norm(O::ladd(x3, y3));
  t1 = O::mul(p.x, p.y);
  x3 = O::mul(t0, t1);
  x3 = O::norm(O::ladd(x3, x3));
  bool z = O::is_zero(p.z);
  Proj<F> r;
  r.x = O::select(x3, O::zero(), z);
  r.y = O::select(y3, O::one(), z);
  r.z = O::select(z3, O::zero(), z);
  return r;
}
template <class O>
BN_DEV Proj<typename O::F> proj_add_lazy(const Proj<typename O::F>& p, const Proj<typename O::F>& q) {
  typedef typename O::F F;
  F t0 = O::mul(p.x, q.x);
  F t1 = O::mul(p.y, q.y);
  F t2 = O::mul(p.z, q.z);
  F t3 = O::norm(O::lsub(O::ladd(t0, t1), O::mul(O::lsub(p.x, p.y), O::lsub(q.x, q.y))));      // x1 y2 + x2 y1
  F t4 = O::norm(O::lsub(O::ladd(t1, t2), O::mul(O::lsub(p.y, p.z), O::lsub(q.y, q.z))));      // y1 z2 + y2 z1
  F y3 = O::lsub(O::ladd(t0, t2), O::mul(O::lsub(p.x, p.z), O::lsub(q.x, q.z)));               // x1 z2 + x2 z1, limbs in (-2^29, 2^30)
  t0 = O::norm(O::ladd(O::ladd(t0, t0), t0));
  t2 = O::mul_b3(t2);
  F z3 = O::norm(O::ladd(t1, t2));
  t1 = O::lsub(t1, t2);                                                                        // product operand only
  y3 = O::mul_b3_lazy(y3);
  F x3 = O::norm(O::lsub(O::mul(t3, t1), O::mul(t4, y3)));
  y3 = O::norm(O::ladd(O::mul(t1, z3), O::mul(y3, t0)));
  z3 = O::norm(O::ladd(O::mul(z3, t4), O::mul(t0, t3)));
  return Proj<F>{x3, y3, z3};
}
template <class O> BN_DEV Proj<typename O::F> proj_neg(const Proj<typename O::F>& p) {
  return Proj<typename O::F>{p.x, O::neg(p.y), p.z};
}

BN_NOINLINE G1P g1_double(G1P p) { return proj_double<OpsFp>(p); }
BN_NOINLINE G1P g1_add(G1P p, G1P q) { return proj_add<OpsFp>(p, q); }

// ---- where a window table of 0P..8P lives (see G1TableGlobal below for why) ----
typedef unsigned int g1tab_u32x4 __attribute__((ext_vector_type(4)));
BN_DEV const F29& f29_of(const F29& a) { return a; }       // coordinate -> its 9 digits (the lane-pair W2 has its own overload)
BN_DEV F29& f29_of(F29& a) { return a; }
template <class PT>
struct ProjTableLocal {
  PT T[9];
  BN_DEV void put(int m, const PT& v) { T[m] = v; }
  BN_DEV PT get(int m) const { return T[m]; }
};
template <class PT>
struct ProjTableGlobal {
  typedef __attribute__((address_space(1))) g1tab_u32x4* gptr;
  gptr base;                                           // this lane's 1 KB region
  BN_DEV void put(int m, const PT& v) {
    const F29 &x = f29_of(v.x), &y = f29_of(v.y), &z = f29_of(v.z);
    const i32 w[28] = {x.v[0], x.v[1], x.v[2], x.v[3], x.v[4], x.v[5], x.v[6], x.v[7], x.v[8],
                       y.v[0], y.v[1], y.v[2], y.v[3], y.v[4], y.v[5], y.v[6], y.v[7], y.v[8],
                       z.v[0], z.v[1], z.v[2], z.v[3], z.v[4], z.v[5], z.v[6], z.v[7], z.v[8], 0};
#pragma unroll
    for (int c = 0; c < 7; ++c) base[7 * m + c] = g1tab_u32x4{(u32)w[4 * c], (u32)w[4 * c + 1], (u32)w[4 * c + 2], (u32)w[4 * c + 3]};
  }
  BN_DEV PT get(int m) const {
    u32 w[28];
#pragma unroll
    for (int c = 0; c < 7; ++c) { const g1tab_u32x4 q = base[7 * m + c]; w[4 * c] = q.x; w[4 * c + 1] = q.y; w[4 * c + 2] = q.z; w[4 * c + 3] = q.w; }
    PT r;
#pragma unroll
    for (int i = 0; i < 9; ++i) { f29_of(r.x).v[i] = (i32)w[i]; f29_of(r.y).v[i] = (i32)w[9 + i]; f29_of(r.z).v[i] = (i32)w[18 + i]; }
    return r;
  }
};
// k*P for the batch kernels: signed fixed-window (w = 4) double-and-add with a WAVE-UNIFORM schedule.
// The reference walks the 256 NAF digits of k (fp.rs:653-662) with a data-dependent add (group.rs:653-664); on a
// 64-wide wavefront that makes every step pay for an addition (some lane always has a non-zero
// digit).  Here every lane does 4 doublings + one complete addition of +-T[|d|] per window
// (T[0] = identity: the RCB formulas are complete, so adding it is exact), 64 windows, table of
// 1P..8P in the lane's scratch frame.  k*P as a group element is the same; only affine-normalised
// results cross the boundary.  k is the Fp VALUE (< p, not reduced mod r: N4).  `nwin` < 64 walks only the low
// 4 nwin bits (callers with a short fixed scalar, e.g. the 63-bit BN parameter of the subgroup check: 17 windows incl. the carry).
// `dbl` / `add` build the table (out-of-line group operations keep that straight-line part small); `dbl_loop` / `add_loop` run in
// the window loop (inlined there, no point travels through the stack frame: measured -16 % on the lane-pair G2 product).
template <class O, class DBL, class ADD, class DBLL, class ADDL, class TAB>
BN_DEV Proj<typename O::F> scalar_mul_window(const Proj<typename O::F>& p, const u32 (&k)[8], DBL dbl, ADD add, int nwin, DBLL dbl_loop, ADDL add_loop, TAB& tab) {
  typedef Proj<typename O::F> Pt;
  // signed recoding: k = sum d_i 16^i, d_i in [-8, 7]; k < 2^254 so the top digit cannot overflow
  signed char dig[64];
  int carry = 0;
#pragma unroll 1
  for (int i = 0; i < 64; ++i) {
    int d = (int)((k[i >> 3] >> (4 * (i & 7))) & 15) + carry;
    carry = d >= 8;
    dig[i] = (signed char)(d - (carry << 4));
  }
  {
    tab.put(0, proj_zero<O>());
    tab.put(1, p);
    const Pt t2 = dbl(p);
    tab.put(2, t2);
    const Pt t3 = add(t2, p);
    tab.put(3, t3);
    const Pt t4 = dbl(t2);
    tab.put(4, t4);
    tab.put(5, add(t4, p));
    const Pt t6 = dbl(t3);
    tab.put(6, t6);
    tab.put(7, add(t6, p));
    tab.put(8, dbl(t4));
  }
  Pt res = proj_zero<O>();
#pragma unroll 1
  for (int i = nwin - 1; i >= 0; --i) {
    if (i != nwin - 1) {
#pragma unroll 1
      for (int j = 0; j < 4; ++j) res = dbl_loop(res);
    }
    int d = dig[i];
    int m = d < 0 ? -d : d;
    Pt q = tab.get(m);
    q.y = O::select(q.y, O::neg(q.y), d < 0);
    res = add_loop(res, q);
  }
  return res;
}
template <class O, class DBL, class ADD, class DBLL, class ADDL>
BN_DEV Proj<typename O::F> scalar_mul_window(const Proj<typename O::F>& p, const u32 (&k)[8], DBL dbl, ADD add, int nwin, DBLL dbl_loop, ADDL add_loop) {
  ProjTableLocal<Proj<typename O::F>> tab;
  return scalar_mul_window<O>(p, k, dbl, add, nwin, dbl_loop, add_loop, tab);
}
template <class O, class DBL, class ADD>
BN_DEV Proj<typename O::F> scalar_mul_window(const Proj<typename O::F>& p, const u32 (&k)[8], DBL dbl, ADD add, int nwin = 64) {
  return scalar_mul_window<O>(p, k, dbl, add, nwin, dbl, add);
}
// G1 group law on the carry-free core (bn254_f29.hpp): the same complete formulas over F29 coordinates.  Class invariant:
// every coordinate is N-class (limbs in [0, 2^29), top limb signed) -- additions are carry-normalised (34 instructions
// against 25 for a saturated modular add) and products are one v_mad_i64_i32 per partial product (233 instructions against
// 439 issue slots).  Value bounds (|V| = |value| / p): products return |V| < VaVb/169 + 1, the x9 of mul_b3 ends in a
// reduce pass (|V| < 0.51), so with inputs |V| <= 7 proj_double returns |V| <= 2.1 and proj_add (inputs <= 2.1) returns
// |V| <= 2.1 -- far inside the |V| <= 40 limit of the core.  Zero tests go through the canonical form.
struct OpsF29 {
  typedef F29 F;
  static BN_DEV F add(const F& a, const F& b) { return f29_norm(f29_add(a, b)); }
  static BN_DEV F sub(const F& a, const F& b) { return f29_norm(f29_sub(a, b)); }
  static BN_DEV F neg(const F& a) { return f29_norm(f29_neg(a)); }
  static BN_DEV F mul(const F& a, const F& b) { return f29_mul_leaf(W_ARGS(a), W_ARGS(b)); }
  static BN_DEV F sqr(const F& a) { return f29_sqr_leaf(W_ARGS(a)); }                 // coordinates are N-class (|limbs| < 2^29): inside f29_sqr's column bound
  static BN_DEV F zero() { return F29{{0, 0, 0, 0, 0, 0, 0, 0, 0}}; }
  // 2^261 mod p, the Montgomery one of the core
  static BN_DEV F one() { return F29{{0x157ccc21, 0x141c2758, 0x185230d3, 0x014c0419, 0x0aa36fb9, 0x1d4240ce, 0x11d54c07, 0x052ac7a8, 0x000dc836}}; }
  static BN_DEV bool is_zero(const F& a) { return fp_is_zero(f29_to_fp(a)); }
  static BN_DEV F select(const F& a, const F& b, bool c) {
    F r;
#pragma unroll
    for (int i = 0; i < 9; ++i) r.v[i] = c ? b.v[i] : a.v[i];
    return r;
  }
  static BN_DEV F mul_b3(const F& a) { return f29_reduce_from([&](int i) { return (i64)a.v[i] * 9; }); }   // 3 b = 9
  // the lazy linear layer of proj_add_lazy / proj_double_lazy
  static BN_DEV F ladd(const F& a, const F& b) { return f29_add(a, b); }
  static BN_DEV F lsub(const F& a, const F& b) { return f29_sub(a, b); }
  static BN_DEV F norm(const F& a) { return f29_norm(a); }
  static BN_DEV F norm_x8(const F& a) { return f29_norm_x8(a); }
  static BN_DEV F norm_sub3(const F& a, const F& b) { return f29_norm_sub3(a, b); }
  static BN_DEV F mul_b3_lazy(const F& a) { return mul_b3(a); }            // a reduce pass over 64-bit terms: any 32-bit limbs
};
// OpsF29 with the product / squaring INLINED at their call sites (no 18 + 9 argument / result moves, no call): for loops small enough to stay in
// the instruction cache with 22 inlined leaves (the G1 window loop: 8 per doubling, 14 per addition).  BN_G1_INL selects it (A/B: 0).
struct OpsF29I : OpsF29 {
  static BN_DEV F mul(const F& a, const F& b) { return f29_mul(a, b); }
  static BN_DEV F sqr(const F& a) { return f29_sqr(a); }
};
#ifndef BN_G1_INL
#define BN_G1_INL 1
#endif
typedef Proj<F29> G1W;
BN_DEV F29 f29_from_fp_reduced(const Fp& a) {
  const F29 t = f29_from_fp(a);
  return f29_reduce_from([&](int i) { return (i64)t.v[i]; });
}
// ---- GLV: k P = k1 P + k2 phi(P) with phi(x, y) = (beta x, y) = lambda P on G1 (beta^3 = 1 in Fp, lambda^3 = 1 mod r) ----
// Every point of E(Fp) has order r (cofactor 1), so k acts through k mod r, and k mod r = k1 + k2 lambda with |k1|, |k2| < 2^128:
// (k1, k2) = (k, 0) - c1 (a1, b1) - c2 (a2, b2) for the reduced lattice basis below and c_i = floor(k g_i / 2^256).
// Halves the doublings of the window schedule (128 + 66 additions instead of 256 + 64).  tools/glv_model.py derives the
// constants from the curve and replays this arithmetic limb-exactly on 20 000 scalars.
//   a1 = 0x89d3256894d213e3, b1 = 0x6f4d8248eeb859fd0be4e1541221250b, a2 = 0x6f4d8248eeb859fc8211bbeb7d4f1128, b2 = -a1
// low `NR` limbs of a (NA limbs) times b (NB limbs)
template <int NA, int NB, int NR>
BN_DEV void mp_mul_lo(u32 (&out)[NR], const u32 (&a)[NA], const u32 (&b)[NB]) {
  u64 acc = 0, carry = 0;
#pragma unroll
  for (int k = 0; k < NR; ++k) {
    acc = carry; carry = 0;
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      const int j = k - i;
      if (j < 0 || j >= NB) continue;
      const u64 pr = (u64)a[i] * b[j];
      acc += pr & 0xffffffffu;
      carry += pr >> 32;
    }
    out[k] = (u32)acc;
    carry += acc >> 32;
  }
}
// |k1|, |k2| (4 limbs each) and their signs for k < p
BN_DEV void glv_decompose(u32 (&m1)[4], bool& n1, u32 (&m2)[4], bool& n2, const u32 (&kin)[8]) {
  u32 k[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) k[i] = kin[i];
  cond_sub_const(k, 0xf0000001u, 0x43e1f593u, 0x79b97091u, 0x2833e848u, 0x8181585du, 0xb85045b6u, 0xe131a029u, 0x30644e72u);   // k mod r (k < p < 2r)
  const u32 g1[3] = {0xc7e0b3d7u, 0xd91d232eu, 0x00000002u};                                  // floor(2^256 |b2| / r)
  const u32 g2[5] = {0x00ff6565u, 0x5398fd03u, 0xa773d2d2u, 0x4ccef014u, 0x00000002u};        // floor(2^256 b1 / r)
  const u32 a1[2] = {0x94d213e3u, 0x89d32568u};
  const u32 b1[4] = {0x1221250bu, 0x0be4e154u, 0xeeb859fdu, 0x6f4d8248u};
  const u32 a2[4] = {0x7d4f1128u, 0x8211bbebu, 0xeeb859fcu, 0x6f4d8248u};
  u32 t1[11], t2[13];
  mp_mul_lo<8, 3, 11>(t1, k, g1);
  mp_mul_lo<8, 5, 13>(t2, k, g2);
  const u32 c1[3] = {t1[8], t1[9], t1[10]};
  const u32 c2[5] = {t2[8], t2[9], t2[10], t2[11], t2[12]};
  u32 c1a1[5], c2a2[5], c2b2[5], c1b1[5];
  mp_mul_lo<3, 2, 5>(c1a1, c1, a1);
  mp_mul_lo<5, 4, 5>(c2a2, c2, a2);
  mp_mul_lo<5, 2, 5>(c2b2, c2, a1);          // |b2| = a1
  mp_mul_lo<3, 4, 5>(c1b1, c1, b1);
  // k1 = k - c1 a1 - c2 a2, k2 = c2 |b2| - c1 b1, both modulo 2^160 (|k_i| < 2^128)
  u32 v1[5], v2[5];
  {
    i64 c = 0;
#pragma unroll
    for (int i = 0; i < 5; ++i) { c += (i64)k[i] - c1a1[i] - c2a2[i]; v1[i] = (u32)c; c >>= 32; }
    c = 0;
#pragma unroll
    for (int i = 0; i < 5; ++i) { c += (i64)c2b2[i] - c1b1[i]; v2[i] = (u32)c; c >>= 32; }
  }
  auto mag = [](u32 (&m)[4], bool& neg, const u32 (&v)[5]) {
    neg = (v[4] >> 31) != 0;
    const u32 s = neg ? 0xffffffffu : 0u;
    u64 c = s & 1u;
#pragma unroll
    for (int i = 0; i < 4; ++i) { c += (u64)(v[i] ^ s); m[i] = (u32)c; c >>= 32; }
  };
  mag(m1, n1, v1);
  mag(m2, n2, v2);
}
// ---- 4-dimensional GLS on G2: psi (untwist-Frobenius-twist) is multiplication by lam = p mod r = 6 x^2 on the r-torsion, and
// lam^4 - lam^2 + 1 = 0 mod r, so k = k0 + k1 lam + k2 lam^2 + k3 lam^3 (mod r) with |k_i| < 2^64 (tools/gls4_model.py: LLL basis, the
// rounding constants, the bound, and a limb-exact replay of this routine on 30 000 scalars).  With the basis rows
//   (2x+1, 0, 2x, 1), (2x, x+1, -x, x), (x+1, x, x, -2x), (2x+1, -x, -(x+1), -x)
// and c_j = round(k g_j / 2^320) (g_j = round(2^320 cofactor_j / r), all positive), everything modulo 2^96:
//   k0 = k - (2 a0 + c0) - 2 a1 - (a2 + c2) - (2 a3 + c3)      a_j = c_j x
//   k1 =   - (a1 + c1) - a2 + a3
//   k2 =   - 2 a0 + a1 - a2 + (a3 + c3)
//   k3 =   - c0 - a1 + 2 a2 + a3
// Out: magnitudes (two limbs) and signs.
BN_DEV void gls4_decompose(u32 (&m)[4][2], bool (&neg)[4], const u32 (&kin)[8]) {
  u32 k[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) k[i] = kin[i];
  cond_sub_const(k, 0xf0000001u, 0x43e1f593u, 0x79b97091u, 0x2833e848u, 0x8181585du, 0xb85045b6u, 0xe131a029u, 0x30644e72u);   // k mod r (k < p < 2r)
  const u32 g0[8] = {0xe558c73cu, 0x353ccca0u, 0x32e42728u, 0x2dff2915u, 0xa3e5577fu, 0x55b4ca7bu, 0xb0d92b95u, 0x9e80318au};
  const u32 g1[8] = {0x5ce18e26u, 0x7d1fff2eu, 0x95d51bb1u, 0x46f4bda9u, 0xfc7184aeu, 0x08e5da66u, 0xb0d92b93u, 0x9e80318au};
  const u32 g2[5] = {0x773a6ef3u, 0x6eb9c714u, 0xc7e0b3d7u, 0xd91d232eu, 0x00000002u};
  const u32 g3[8] = {0xbb8bb500u, 0x23038c29u, 0xcef3cd3fu, 0xc170977du, 0xa3e5577du, 0x55b4ca7bu, 0xb0d92b95u, 0x9e80318au};
  const u32 bx[2] = {(u32)BN_BLS_X, (u32)(BN_BLS_X >> 32)};
  u32 c[4][3], a[4][3];
  auto top = [](u32 (&out)[3], const u32 (&t)[13]) {      // (t + 2^319) >> 320, low three limbs
    u64 cy = ((u64)t[9] + 0x80000000u) >> 32;
#pragma unroll
    for (int i = 0; i < 3; ++i) { cy += t[10 + i]; out[i] = (u32)cy; cy >>= 32; }
  };
  {
    u32 t[13];
    mp_mul_lo<8, 8, 13>(t, k, g0); top(c[0], t);
    mp_mul_lo<8, 8, 13>(t, k, g1); top(c[1], t);
    mp_mul_lo<8, 5, 13>(t, k, g2); top(c[2], t);
    mp_mul_lo<8, 8, 13>(t, k, g3); top(c[3], t);
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) mp_mul_lo<3, 2, 3>(a[j], c[j], bx);
  // signed small-coefficient combinations modulo 2^96
  const int ca[4][4] = {{-2, -2, -1, -2}, {0, -1, -1, 1}, {-2, 1, -1, 1}, {0, -1, 2, 1}};    // coefficient of a_j in k_i
  const int cc[4][4] = {{-1, 0, -1, -1}, {0, -1, 0, 0}, {0, 0, 0, 1}, {-1, 0, 0, 0}};        // coefficient of c_j in k_i
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    u32 v[3];
    i64 cy = 0;
#pragma unroll
    for (int l = 0; l < 3; ++l) {
      cy += (i == 0) ? (i64)k[l] : 0;
#pragma unroll
      for (int j = 0; j < 4; ++j) cy += (i64)a[j][l] * ca[i][j] + (i64)c[j][l] * cc[i][j];
      v[l] = (u32)cy;
      cy >>= 32;
    }
    neg[i] = (v[2] >> 31) != 0;
    const u32 sgn = neg[i] ? 0xffffffffu : 0u;
    u64 cm = sgn & 1u;
#pragma unroll
    for (int l = 0; l < 2; ++l) { cm += (u64)(v[l] ^ sgn); m[i][l] = (u32)cm; cm >>= 32; }
  }
}
// signed 4-bit digits of a 64-bit magnitude: 17 digits (the last one is the carry)
BN_DEV void gls4_digits(signed char (&dig)[17], const u32 (&m)[2]) {
  int carry = 0;
#pragma unroll 1
  for (int i = 0; i < 16; ++i) {
    int d = (int)((m[i >> 3] >> (4 * (i & 7))) & 15) + carry;
    carry = d >= 8;
    dig[i] = (signed char)(d - (carry << 4));
  }
  dig[16] = (signed char)carry;
}
// signed 4-bit digits of a 128-bit magnitude: m = sum d_i 16^i, d_i in [-8, 7], 33 digits (the last one is the carry)
BN_DEV void glv_digits(signed char (&dig)[33], const u32 (&m)[4]) {
  int carry = 0;
#pragma unroll 1
  for (int i = 0; i < 32; ++i) {
    int d = (int)((m[i >> 3] >> (4 * (i & 7))) & 15) + carry;
    carry = d >= 8;
    dig[i] = (signed char)(d - (carry << 4));
  }
  dig[32] = (signed char)carry;
}
// k * P for the batch kernels, on the carry-free core: GLV split, then the wave-uniform signed-window schedule on both halves
// at once -- every lane does 4 doublings and two complete additions per window (table entry 0 is the identity), 33 windows,
// one table of 1P..8P; the phi-image of a table entry is a single multiplication of its X by beta.  The saturated projective
// result is a representative of the same point (only affine-normalised values cross the boundary).
// Where the window table of 0P..8P lives.  In the stack frame (G1TableLocal) a lookup is a dynamically indexed read of private memory:
// the hardware interleaves a wavefront's private dwords lane by lane, so 64 lanes reading entry m of THEIR table touch up to nine different
// 256-byte rows per instruction and fetch whole 64-byte sectors for 4 useful bytes each (rocprofv3: 59 KB of HBM traffic per G1 product,
// 3.8 TB/s).  G1TableGlobal keeps each lane's table CONTIGUOUS in a leased global block (1 KB per lane: nine 112-byte entries of
// 7 x 16 bytes), so a lookup is seven 16-byte loads from two or three sectors of the lane's own region.
typedef ProjTableLocal<G1W> G1TableLocal;
typedef ProjTableGlobal<G1W> G1TableGlobal;
constexpr size_t G1_TABLE_BYTES_PER_LANE = 1024;
template <class TAB>
BN_DEV G1P g1_scalar_mul_t(G1P p, const u32 (&k)[8], TAB& tab) {
  u32 m1[4], m2[4];
  bool n1, n2;
  glv_decompose(m1, n1, m2, n2, k);
  signed char d1[33], d2[33];
  glv_digits(d1, m1);
  glv_digits(d2, m2);
  // beta 2^261 mod p
  const F29 beta{{0x18ccb791, 0x175b1c3a, 0x0b83d6e2, 0x0e8ed071, 0x1282bee2, 0x04220e84, 0x1fe4017f, 0x15084d4a, 0x00169119}};
#if BN_G1_INL
  auto dbl = [](const G1W& a) { return proj_double_lazy<OpsF29I>(a); };
  auto add = [](const G1W& a, const G1W& b) { return proj_add_lazy<OpsF29I>(a, b); };
#else
  auto dbl = [](const G1W& a) { return proj_double_lazy<OpsF29>(a); };
  auto add = [](const G1W& a, const G1W& b) { return proj_add_lazy<OpsF29>(a, b); };
#endif
  {
    G1W t1{f29_from_fp_reduced(p.x), f29_from_fp_reduced(p.y), f29_from_fp_reduced(p.z)};
    // an identity handed over as (x : y : 0) becomes the canonical (0 : 1 : 0): the complete formulas keep Z = 0 only
    // for the canonical representative once two additions follow each other without a doubling in between
    const bool pinf = OpsF29::is_zero(t1.z);
    t1.x = OpsF29::select(t1.x, OpsF29::zero(), pinf);
    t1.y = OpsF29::select(t1.y, OpsF29::one(), pinf);
    t1.z = OpsF29::select(t1.z, OpsF29::zero(), pinf);
    if (n1) t1.y = OpsF29::neg(t1.y);                  // the table holds multiples of sign(k1) P
    tab.put(0, proj_zero<OpsF29>());
    tab.put(1, t1);
    const G1W t2 = dbl(t1);
    tab.put(2, t2);
    const G1W t3 = add(t2, t1);
    tab.put(3, t3);
    const G1W t4 = dbl(t2);
    tab.put(4, t4);
    tab.put(5, add(t4, t1));
    const G1W t6 = dbl(t3);
    tab.put(6, t6);
    tab.put(7, add(t6, t1));
    tab.put(8, dbl(t4));
  }
  const bool flip2 = n1 != n2;                            // phi(table) carries sign(k1); k2 wants sign(k2)
  G1W res = proj_zero<OpsF29>();
#pragma unroll 1
  for (int i = 32; i >= 0; --i) {
    if (i != 32) {
#pragma unroll 1
      for (int j = 0; j < 4; ++j) res = dbl(res);
    }
    {
      const int d = d1[i], m = d < 0 ? -d : d;
      G1W q = tab.get(m);
      q.y = OpsF29::select(q.y, OpsF29::neg(q.y), d < 0);
      res = add(res, q);
    }
    {
      const int d = d2[i], m = d < 0 ? -d : d;
      G1W q = tab.get(m);
      q.x = OpsF29::mul(q.x, beta);
      q.y = OpsF29::select(q.y, OpsF29::neg(q.y), (d < 0) != flip2);
      res = add(res, q);
    }
  }
  return G1P{f29_to_fp(res.x), f29_to_fp(res.y), f29_to_fp(res.z)};
}
BN_NOINLINE G1P g1_scalar_mul(G1P p, const u32 (&k)[8]) {
  G1TableLocal tab;
  return g1_scalar_mul_t(p, k, tab);
}
// `region`: this lane's G1_TABLE_BYTES_PER_LANE bytes of a leased global block
BN_NOINLINE G1P g1_scalar_mul_ws(G1P p, const u32 (&k)[8], void* region) {
  G1TableGlobal tab{(G1TableGlobal::gptr)region};
  return g1_scalar_mul_t(p, k, tab);
}
// group.rs:475-495: affine = (X/Z, Y/Z); infinity iff Z^-1 == 0 -> (0, 1, inf)
BN_DEV void g1_to_affine(Fp& x, Fp& y, bool& inf, const G1P& p) {
  Fp zi = fp_inv(p.z);
  inf = fp_is_zero(zi);
  x = fp_select(fp_mul(p.x, zi), fp_zero(), inf);
  y = fp_select(fp_mul(p.y, zi), fp_one(), inf);
}
// g1.rs:111-132
BN_DEV bool g1_on_curve_affine(const Fp& x, const Fp& y) {
  return fp_eq(fp_sub(fp_sqr(y), fp_mul(fp_sqr(x), x)), fp_small(3));
}
// g2.rs:140-152: psi(x, y) = (eps0 * conj(x), eps1 * conj(y))
BN_DEV void g2_psi_affine(Fp2& xo, Fp2& yo, const Fp2& x, const Fp2& y) {
  xo = fp2_mul(fp2_const(C_EPS_EXP0), fp2_conj(x));
  yo = fp2_mul(fp2_const(C_EPS_EXP1), fp2_conj(y));
}

// ---------------------------------------------------------------- final exponentiation pieces (the Fp12 selector of tower.hip) --------
// pairing.rs:274-284
BN_DEV void fp4_square(Fp2& c0, Fp2& c1, const Fp2& a, const Fp2& b) {
  Fp2 t0 = fp2_sqr(a);
  Fp2 t1 = fp2_sqr(b);
  c0 = fp2_add(fp2_mul_xi(t1), t0);
  c1 = fp2_sub(fp2_sub(fp2_sqr(fp2_add(a, b)), t0), t1);
}
// pairing.rs:309-350 (Granger-Scott squaring in the cyclotomic subgroup)
BN_NOINLINE void cyclotomic_sqr(Fp12& r, const Fp12& f) {
  Fp2 z0 = f.c0.c0, z4 = f.c0.c1, z3 = f.c0.c2, z2 = f.c1.c0, z1 = f.c1.c1, z5 = f.c1.c2;
  Fp2 t0, t1, t2, t3;
  fp4_square(t0, t1, z0, z1);
  z0 = fp2_sub(t0, z0); z0 = fp2_add(fp2_dbl(z0), t0);
  z1 = fp2_add(t1, z1); z1 = fp2_add(fp2_dbl(z1), t1);
  fp4_square(t0, t1, z2, z3);
  fp4_square(t2, t3, z4, z5);
  z4 = fp2_sub(t0, z4); z4 = fp2_add(fp2_dbl(z4), t0);
  z5 = fp2_add(t1, z5); z5 = fp2_add(fp2_dbl(z5), t1);
  t0 = fp2_mul_xi(t3);
  z2 = fp2_add(t0, z2); z2 = fp2_add(fp2_dbl(z2), t0);
  z3 = fp2_sub(t2, z3); z3 = fp2_add(fp2_dbl(z3), t2);
  r.c0.c0 = z0; r.c0.c1 = z4; r.c0.c2 = z3;
  r.c1.c0 = z2; r.c1.c1 = z1; r.c1.c2 = z5;
}
// pairing.rs:366-392: f^x then conjugate.  The reference walks all 256 bits of the 63-bit x with
// square-and-multiply from one (62 useful squarings + 27 products).  f lives in the cyclotomic subgroup
// here (every caller passes a value that went through the easy part), where f^-1 is the conjugate, so
// the same power f^x is reached with the width-3 signed-digit form of x (digits in {+-1, +-3}, 18
// non-zero): 62 cyclotomic squarings + 17 products + f^3.  Same field element, bit for bit.
#define BN_X_W3_NZ 0x4908924444891211ull   // bit i set iff digit i != 0        (x = sum d_i 2^i, top digit d_62 = +1)
#define BN_X_W3_NEG 0x0108000400880210ull  // digit i negative
#define BN_X_W3_THREE 0x0108804404880200ull  // |digit i| == 3
// saturated-core version (kept as the test twin of the carry-free one below)
BN_NOINLINE void exp_by_neg_z_sat(Fp12& r, const Fp12& f) {
  Fp12 f3, res, t;
  cyclotomic_sqr(t, f);
  fp12_mul(f3, t, f);
  res = f;
  const u64 nz = BN_X_W3_NZ, ng = BN_X_W3_NEG, th = BN_X_W3_THREE;
#pragma unroll 1
  for (int i = 61; i >= 0; --i) {
    cyclotomic_sqr(res, res);
    if ((nz >> i) & 1) {
      const Fp12& m = ((th >> i) & 1) ? f3 : f;
      if ((ng >> i) & 1) { fp12_conj(t, m); fp12_mul(res, res, t); }
      else fp12_mul(res, res, m);
    }
  }
  fp12_conj(r, res);
}
// The same chain on the carry-free 9 x 29-bit core (bn254_f29.hpp): f is converted once, the 62 cyclotomic
// squarings and 18 products run without carry handling or modular add/subs, the result is converted back.
BN_NOINLINE void exp_by_neg_z(Fp12& r, const Fp12& f) {
  U12 uf, uf3, res, t;
  u12_from_fp12(uf, f);
  u12_reduce(uf);
  u12_cyclotomic_sqr(t, uf);
  u12_mul(uf3, t, uf);
  res = uf;
  const u64 nz = BN_X_W3_NZ, ng = BN_X_W3_NEG, th = BN_X_W3_THREE;
#pragma unroll 1
  for (int i = 61; i >= 0; --i) {
    u12_cyclotomic_sqr(res, res);
    if ((nz >> i) & 1) {
      const U12& m = ((th >> i) & 1) ? uf3 : uf;
      if ((ng >> i) & 1) { u12_conj(t, m); u12_mul(res, res, t); }
      else u12_mul(res, res, m);
    }
  }
  u12_conj(t, res);
  u12_to_fp12(r, t);
}
}  // namespace bn254
