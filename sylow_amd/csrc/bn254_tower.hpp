// Extension tower Fp2 = Fp[u]/(u^2+1), Fp6 = Fp2[v]/(v^3-xi), Fp12 = Fp6[w]/(w^2-v), xi = 9+u.
// Batched replacement for sylow's src/fields/{extensions,fp2,fp6,fp12}.rs.  All values are exact
// residues, so Karatsuba here vs. the reference's schoolbook forms is bit-identical.
//
// This is the ONE-ELEMENT-PER-LANE tower: the twin of the lane-pair implementation (bn254_pair.hpp / bn254_pair29.hpp, which
// the pairing kernels use by default) and the layer behind the Fp2/Fp6/Fp12 batch entry points.
// Code-size / register policy: the out-of-line leaf is fp_mul(Fp, Fp) (16 ABI argument registers,
// ~50-VGPR footprint); the Fp2 layer is inlined; Fp6/Fp12 products, the sparse line multiplication and
// the cyclotomic square are out-of-line mid-level routines taking references, so Fp12-sized values
// live in the per-lane scratch frame while Fp2-sized values stay in VGPRs.
#pragma once
#include "bn254_constants.hpp"
#include "bn254_fp.hpp"

namespace bn254 {

struct Fp2 { Fp c0, c1; };
struct Fp6 { Fp2 c0, c1, c2; };
struct Fp12 { Fp6 c0, c1; };

BN_DEV Fp fp_const(const uint32_t (&c)[8]) {
  return fp_from_limbs(c[0], c[1], c[2], c[3], c[4], c[5], c[6], c[7]);
}
BN_DEV Fp2 fp2_const(const uint32_t (&c)[2][8]) { return Fp2{fp_const(c[0]), fp_const(c[1])}; }

// ------------------------------------------------------------------ Fp2 (fp2.rs) -------------
BN_DEV Fp2 fp2_zero() { return Fp2{fp_zero(), fp_zero()}; }
BN_DEV Fp2 fp2_one() { return Fp2{fp_one(), fp_zero()}; }
BN_DEV Fp2 fp2_add(const Fp2& a, const Fp2& b) { return Fp2{fp_add(a.c0, b.c0), fp_add(a.c1, b.c1)}; }
BN_DEV Fp2 fp2_sub(const Fp2& a, const Fp2& b) { return Fp2{fp_sub(a.c0, b.c0), fp_sub(a.c1, b.c1)}; }
BN_DEV Fp2 fp2_neg(const Fp2& a) { return Fp2{fp_neg(a.c0), fp_neg(a.c1)}; }
BN_DEV Fp2 fp2_dbl(const Fp2& a) { return Fp2{fp_dbl(a.c0), fp_dbl(a.c1)}; }
BN_DEV Fp2 fp2_conj(const Fp2& a) { return Fp2{a.c0, fp_neg(a.c1)}; }  // frobenius(odd), fp2.rs:119-133
BN_DEV bool fp2_is_zero(const Fp2& a) { return fp_is_zero(a.c0) && fp_is_zero(a.c1); }
BN_DEV Fp2 fp2_select(const Fp2& a, const Fp2& b, bool c) { return Fp2{fp_select(a.c0, b.c0, c), fp_select(a.c1, b.c1, c)}; }

// fp2.rs:285-306 (value): (a0 b0 - a1 b1, a0 b1 + a1 b0) with lazy reduction -- each coordinate is one fused
// two-product column pass + one Montgomery reduction (fp_dot2_inline), -a1 entering as p - a1.
// Out-of-line leaf: `a` travels in the 16 ABI argument registers, `b` by pointer (it usually already lives
// in the caller's scratch frame: a coefficient of an Fp6/Fp12 operand or a constant).
BN_DEV Fp2 fp2_mul(const Fp2& a, const Fp2& b) {
  Fp c0 = fp_dot2_inline(a.c0, b.c0, fp_neg_lazy(a.c1), b.c1);
  Fp c1 = fp_dot2_inline(a.c0, b.c1, a.c1, b.c0);
  return Fp2{c0, c1};
}
// fp2.rs:164-171: (a0+a1)(a0-a1), 2 a0 a1
BN_DEV Fp2 fp2_sqr(const Fp2& a) {
  Fp s = fp_add_lazy(a.c0, a.c1);                                     // < 2p, times d < p
  Fp d = fp_sub(a.c0, a.c1);
  Fp t = fp_mul(a.c0, a.c1);
  return Fp2{fp_mul(s, d), fp_dbl(t)};
}
   // == scale(TWO_INV)
// x (9+u): (9a - b, a + 9b)  (fp2.rs:99-107), one multiply-by-9 pass per coordinate
BN_DEV Fp2 fp2_mul_xi(const Fp2& a) {
  return Fp2{fp_mul9_addsub<false>(a.c0, a.c1), fp_mul9_addsub<true>(a.c1, a.c0)};
}
// fp2.rs:355-360: conj / (a0^2 + a1^2); inv(0) = 0
BN_DEV Fp2 fp2_inv(const Fp2& a) {
  Fp t = fp_inv(fp_add(fp_sqr(a.c0), fp_sqr(a.c1)));
  return Fp2{fp_mul(a.c0, t), fp_neg(fp_mul(a.c1, t))};
}

BN_DEV Fp6 fp6_add(const Fp6& a, const Fp6& b) { return Fp6{fp2_add(a.c0, b.c0), fp2_add(a.c1, b.c1), fp2_add(a.c2, b.c2)}; }
BN_DEV Fp6 fp6_sub(const Fp6& a, const Fp6& b) { return Fp6{fp2_sub(a.c0, b.c0), fp2_sub(a.c1, b.c1), fp2_sub(a.c2, b.c2)}; }
BN_DEV Fp6 fp6_neg(const Fp6& a) { return Fp6{fp2_neg(a.c0), fp2_neg(a.c1), fp2_neg(a.c2)}; }
BN_DEV Fp6 fp6_dbl(const Fp6& a) { return Fp6{fp2_dbl(a.c0), fp2_dbl(a.c1), fp2_dbl(a.c2)}; }
// x v: (xi c2, c0, c1)  (fp6.rs:189-191)
BN_DEV Fp6 fp6_mul_v(const Fp6& a) { return Fp6{fp2_mul_xi(a.c2), a.c0, a.c1}; }

// fp6.rs:283-367 (value); Karatsuba over v: 6 Fp2 products
BN_NOINLINE void fp6_mul(Fp6& r, const Fp6& a, const Fp6& b) {
  Fp2 v0 = fp2_mul(a.c0, b.c0);
  Fp2 v1 = fp2_mul(a.c1, b.c1);
  Fp2 v2 = fp2_mul(a.c2, b.c2);
  Fp2 t0 = fp2_mul(fp2_add(a.c1, a.c2), fp2_add(b.c1, b.c2));
  Fp2 t1 = fp2_mul(fp2_add(a.c0, a.c1), fp2_add(b.c0, b.c1));
  Fp2 t2 = fp2_mul(fp2_add(a.c0, a.c2), fp2_add(b.c0, b.c2));
  Fp2 r0 = fp2_add(v0, fp2_mul_xi(fp2_sub(fp2_sub(t0, v1), v2)));
  Fp2 r1 = fp2_add(fp2_sub(fp2_sub(t1, v0), v1), fp2_mul_xi(v2));
  Fp2 r2 = fp2_add(fp2_sub(fp2_sub(t2, v0), v2), v1);
  r.c0 = r0; r.c1 = r1; r.c2 = r2;
}
// fp6.rs:219-236 (CH-SQR2: 2 products + 3 squares)
BN_NOINLINE void fp6_sqr(Fp6& r, const Fp6& a) {
  Fp2 s0 = fp2_sqr(a.c0);
  Fp2 ab = fp2_mul(a.c0, a.c1);
  Fp2 s1 = fp2_dbl(ab);
  Fp2 s2 = fp2_sqr(fp2_add(fp2_sub(a.c0, a.c1), a.c2));
  Fp2 bc = fp2_mul(a.c1, a.c2);
  Fp2 s3 = fp2_dbl(bc);
  Fp2 s4 = fp2_sqr(a.c2);
  Fp2 r0 = fp2_add(s0, fp2_mul_xi(s3));
  Fp2 r1 = fp2_add(s1, fp2_mul_xi(s4));
  Fp2 r2 = fp2_sub(fp2_sub(fp2_add(fp2_add(s1, s2), s3), s0), s4);
  r.c0 = r0; r.c1 = r1; r.c2 = r2;
}
// extensions.rs:86-94 with F = Fp2
BN_DEV void fp6_scale(Fp6& r, const Fp6& a, const Fp2& k) {
  r.c0 = fp2_mul(a.c0, k); r.c1 = fp2_mul(a.c1, k); r.c2 = fp2_mul(a.c2, k);
}
// fp6.rs:415-423
BN_NOINLINE void fp6_inv(Fp6& r, const Fp6& a) {
  Fp2 t0 = fp2_sub(fp2_sqr(a.c0), fp2_mul(a.c1, fp2_mul_xi(a.c2)));
  Fp2 t1 = fp2_sub(fp2_mul_xi(fp2_sqr(a.c2)), fp2_mul(a.c0, a.c1));
  Fp2 t2 = fp2_sub(fp2_sqr(a.c1), fp2_mul(a.c0, a.c2));
  Fp2 d = fp2_add(fp2_mul_xi(fp2_add(fp2_mul(a.c2, t1), fp2_mul(a.c1, t2))), fp2_mul(a.c0, t0));
  Fp2 di = fp2_inv(d);
  r.c0 = fp2_mul(di, t0); r.c1 = fp2_mul(di, t1); r.c2 = fp2_mul(di, t2);
}
// fp6.rs:203-209, exponent e in {1,2,3}
template <int E>
BN_DEV void fp6_frobenius(Fp6& r, const Fp6& a) {
  constexpr bool odd = (E & 1) != 0;
  Fp2 x0 = odd ? fp2_conj(a.c0) : a.c0;
  Fp2 x1 = odd ? fp2_conj(a.c1) : a.c1;
  Fp2 x2 = odd ? fp2_conj(a.c2) : a.c2;
  const uint32_t (&k1)[2][8] = (E == 1) ? C_FROB6_C1_1 : (E == 2) ? C_FROB6_C1_2 : C_FROB6_C1_3;
  const uint32_t (&k2)[2][8] = (E == 1) ? C_FROB6_C2_1 : (E == 2) ? C_FROB6_C2_2 : C_FROB6_C2_3;
  r.c0 = x0;
  r.c1 = fp2_mul(x1, fp2_const(k1));
  r.c2 = fp2_mul(x2, fp2_const(k2));
}

// fp12.rs:229-238
BN_NOINLINE void fp12_mul(Fp12& r, const Fp12& a, const Fp12& b) {
  Fp6 t0, t1, t2;
  fp6_mul(t0, a.c0, b.c0);
  fp6_mul(t1, a.c1, b.c1);
  {
    Fp6 sa = fp6_add(a.c0, a.c1);
    Fp6 sb = fp6_add(b.c0, b.c1);
    fp6_mul(t2, sa, sb);
  }
  r.c1 = fp6_sub(fp6_sub(t2, t0), t1);
  r.c0 = fp6_add(fp6_mul_v(t1), t0);
}
// fp12.rs:536-550 (complex squaring: 2 Fp6 products)
BN_NOINLINE void fp12_sqr(Fp12& r, const Fp12& a) {
  Fp6 c0 = fp6_sub(a.c0, a.c1);
  Fp6 c3 = fp6_sub(a.c0, fp6_mul_v(a.c1));
  Fp6 c2;
  fp6_mul(c2, a.c0, a.c1);
  Fp6 t;
  fp6_mul(t, c0, c3);
  t = fp6_add(t, c2);
  r.c1 = fp6_dbl(c2);
  r.c0 = fp6_add(t, fp6_mul_v(c2));
}
// fp12.rs:381-383
BN_DEV void fp12_conj(Fp12& r, const Fp12& a) { r.c0 = a.c0; r.c1 = fp6_neg(a.c1); }
// fp12.rs:281-286
BN_NOINLINE void fp12_inv(Fp12& r, const Fp12& a) {
  Fp6 s0, s1, t;
  fp6_sqr(s0, a.c0);
  fp6_sqr(s1, a.c1);
  Fp6 d = fp6_sub(s0, fp6_mul_v(s1));
  fp6_inv(t, d);
  fp6_mul(s0, a.c0, t);
  fp6_mul(s1, a.c1, t);
  r.c0 = s0;
  r.c1 = fp6_neg(s1);
}
// fp12.rs:515-522, exponent in {1,2,3}
template <int E>
BN_NOINLINE void fp12_frobenius(Fp12& r, const Fp12& a) {
  Fp6 x0, x1;
  fp6_frobenius<E>(x0, a.c0);
  fp6_frobenius<E>(x1, a.c1);
  const uint32_t (&k)[2][8] = (E == 1) ? C_FROB12_C1_1 : (E == 2) ? C_FROB12_C1_2 : C_FROB12_C1_3;
  Fp6 y1;
  fp6_scale(y1, x1, fp2_const(k));
  r.c0 = x0;
  r.c1 = y1;
}
// fp12.rs:426-503: f * (ell_0 + ell_vv v^2... ) with the sparse operand in slots 0, 2, 4 of the
// [z0..z5] = [c0.0,c0.1,c0.2,c1.0,c1.1,c1.2] view (x0 = ell_0, x2 = ell_vv, x4 = ell_vw): the 13 Fp2
// products of the reference's mul_by_024 sequence, re-ordered so that only the three line coefficients,
// d0/d2/d4 and the running sum s1 stay live (7 Fp2 = 112 VGPRs) while the z_i are re-read from the
// scratch copy of f when needed -- loads instead of spill stores.  OUT-OF-PLACE (o must not alias f):
// callers ping-pong between two accumulators, so f stays read-only and its coefficients can be re-loaded.
BN_DEV void sched_fence() { asm volatile("" ::: "memory"); }
BN_NOINLINE void fp12_sparse_mul(Fp12& __restrict__ o, const Fp12& __restrict__ f, const Fp2& ell_0, const Fp2& ell_vw, const Fp2& ell_vv) {
  const Fp2 x0 = ell_0, x2 = ell_vv, x4 = ell_vw;
  const Fp2 d0 = fp2_mul(f.c0.c0, x0);
  const Fp2 d2 = fp2_mul(f.c0.c2, x2);
  const Fp2 d4 = fp2_mul(f.c1.c1, x4);
  // out0 = xi(z1 x2 + d4) + d0 ; out1 = xi(z5 x4 + d2) + z1 x0
  Fp2 s1 = fp2_mul(f.c0.c1, x2);
  o.c0.c0 = fp2_add(fp2_mul_xi(fp2_add(s1, d4)), d0);
  sched_fence();
  {
    Fp2 t3 = fp2_mul(f.c1.c2, x4);
    s1 = fp2_add(s1, t3);
    Fp2 t4 = fp2_mul_xi(fp2_add(t3, d2));
    t3 = fp2_mul(f.c0.c1, x0);
    s1 = fp2_add(s1, t3);
    o.c0.c1 = fp2_add(t4, t3);
  }
  sched_fence();
  {  // out2 = (z0 + z2)(x0 + x2) - d0 - d2 + z3 x4
    Fp2 t3 = fp2_sub(fp2_sub(fp2_mul(fp2_add(f.c0.c0, f.c0.c2), fp2_add(x0, x2)), d0), d2);
    Fp2 t4 = fp2_mul(f.c1.c0, x4);
    s1 = fp2_add(s1, t4);
    o.c0.c2 = fp2_add(t3, t4);
  }
  sched_fence();
  {  // out3 = xi((z2 + z4)(x2 + x4) - d2 - d4) + z3 x0
    Fp2 t3 = fp2_sub(fp2_sub(fp2_mul(fp2_add(f.c0.c2, f.c1.c1), fp2_add(x2, x4)), d2), d4);
    Fp2 t4 = fp2_mul_xi(t3);
    t3 = fp2_mul(f.c1.c0, x0);
    s1 = fp2_add(s1, t3);
    o.c1.c0 = fp2_add(t4, t3);
  }
  sched_fence();
  {  // out4 = xi(z5 x2) + (z0 + z4)(x0 + x4) - d0 - d4
    Fp2 t3 = fp2_mul(f.c1.c2, x2);
    s1 = fp2_add(s1, t3);
    Fp2 t4 = fp2_mul_xi(t3);
    t3 = fp2_sub(fp2_sub(fp2_mul(fp2_add(f.c0.c0, f.c1.c1), fp2_add(x0, x4)), d0), d4);
    o.c1.c1 = fp2_add(t4, t3);
  }
  sched_fence();
  {  // out5 = (z1 + z3 + z5)(x0 + x2 + x4) - s1
    Fp2 s0 = fp2_add(fp2_add(f.c0.c1, f.c1.c0), f.c1.c2);
    Fp2 t0 = fp2_add(fp2_add(x0, x2), x4);
    o.c1.c2 = fp2_sub(fp2_mul(s0, t0), s1);
  }
}

}  // namespace bn254
