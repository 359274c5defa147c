// bn254_pair29.hpp -- the Fp12 layer of the lane-pair pairing on the carry-free 9 x 29-bit core (bn254_f29.hpp).
//
// Same lane-pair layout as bn254_pair.hpp (even lane = c0 coordinate, odd lane = c1 coordinate of every Fp2), but each
// coordinate is an F29 (9 signed 29-bit limbs, Montgomery factor 2^261).  On this layout the kernel is VALU-bound
// (no scratch traffic left), and the carry-free core needs ~30 % fewer instructions per Fp2 product (one v_mad_i64_i32 per
// partial product, no v_addc) and 9-instruction lazy additions, so it is used for everything that touches the Fp12
// accumulator: the squarings and line multiplications of the Miller loop and the whole hard part of the final exponentiation.
//
// Bounds discipline (worst case, see bn254_f29.hpp for L and V):
//   R  "reduced":     limbs 0..7 in [0, 2^29), |value| < 0.51 p
//   N  "normalized":  limbs 0..7 in [0, 2^29), top limb signed, |value| <= 4 p
//   D  "difference":  |limbs| < 2^29 (a difference of two N values, or a negated N value), |value| <= 4 p
//  * product operands (w2_mul) may be R, N or D; w2_sqr needs non-negative limbs (R or N);
//  * a sum of two/three N values must go through f29_norm before it is a product operand;
//  * every value STORED by a routine here is R or N -- lazy combinations end in f29_reduce_from (R) or f29_norm (N).
// With |V| <= 4 on the operands a fused two-product pass returns |V| < 2*16/169 + 1 < 1.2, a pre-added Karatsuba operand
// has |V| <= 8 and its products |V| < 2*64/169 + 1 < 1.8, so values never approach the |V| <= 40 limit of the core.
#pragma once
#include "bn254_f29.hpp"
#include "bn254_pair.hpp"

namespace bn254 {
namespace pl {

struct W2 { F29 c; };
BN_DEV const F29& f29_of(const W2& a) { return a.c; }      // for the window-table policies of bn254_pairing.hpp
BN_DEV F29& f29_of(W2& a) { return a.c; }
struct W6 { W2 c0, c1, c2; };
struct W12 { W6 c0, c1; };

BN_DEV F29 xchg9(const F29& a) {
  F29 r;
#pragma unroll
  for (int i = 0; i < 9; ++i) r.v[i] = (i32)swap_u32((u32)a.v[i]);
  return r;
}
BN_DEV F29 sel9(bool odd, const F29& if_even, const F29& if_odd) {
  F29 r;
#pragma unroll
  for (int i = 0; i < 9; ++i) r.v[i] = odd ? if_odd.v[i] : if_even.v[i];
  return r;
}
BN_DEV F29 f29_reduce(const F29& a) {
  const F29* const x[1] = {&a};
  const i32 k[1] = {bn_keep(1)};
  return f29_reduce_terms(x, k);
}

// ---- leaves: 18 scalar ABI arguments (two 9-limb structs would travel through the stack) -----------------------------
// c0 lane: a0 b0 - a1 b1;  c1 lane: a1 b0 + a0 b1.  Operands R / N / D.  Output N, |V| < (VaVb + Va'Vb')/169 + 1.
// With X / Y this lane's coordinates of a / b and X' / Y' the partner's, both lanes evaluate  X U + W V  with
//   U = Y  on c0 lanes, Y' on c1 lanes      v_cndmask_b32_dpp: the partner's register is read as src0 of the select itself
//   W = X' on both                          v_mov_b32_dpp
//   V = -Y' on c0 lanes, Y on c1 lanes      IN PLACE on the b registers (the callee owns them), two bank-masked instructions per limb that
//                                            write the c0 lanes only -- possible because a lane's role is its DPP bank (bn254_pair.hpp)
// 36 instructions + one for the zero and no lane-parity mask to build (adjacent-lane pairs: 45 + 4).
// s_nop 1 (two wait states) covers the VALU-write -> DPP-read hazard at block entry (the assembler does not see into inline asm).
#define BN_DPP_SWAP "row_half_mirror row_mask:0xf bank_mask:0xf"
#define BN_DPP_SWAP_C0 "row_half_mirror row_mask:0xf bank_mask:0x5"
#define BN_DPP_SELF_C0 "quad_perm:[0,1,2,3] row_mask:0xf bank_mask:0x5"      // no permutation: the DPP form only for its bank mask
BN_DEV void w2_exchange(const F29& a, F29& b, F29& U, F29& W) {
  asm("s_nop 1\n\ts_mov_b32 vcc_lo, 0x0f0f0f0f\n\ts_mov_b32 vcc_hi, 0x0f0f0f0f\n\t"
      "v_cndmask_b32_dpp %0, %9, %9, vcc " BN_DPP_SWAP "\n\tv_cndmask_b32_dpp %1, %10, %10, vcc " BN_DPP_SWAP "\n\t"
      "v_cndmask_b32_dpp %2, %11, %11, vcc " BN_DPP_SWAP "\n\tv_cndmask_b32_dpp %3, %12, %12, vcc " BN_DPP_SWAP "\n\t"
      "v_cndmask_b32_dpp %4, %13, %13, vcc " BN_DPP_SWAP "\n\tv_cndmask_b32_dpp %5, %14, %14, vcc " BN_DPP_SWAP "\n\t"
      "v_cndmask_b32_dpp %6, %15, %15, vcc " BN_DPP_SWAP "\n\tv_cndmask_b32_dpp %7, %16, %16, vcc " BN_DPP_SWAP "\n\t"
      "v_cndmask_b32_dpp %8, %17, %17, vcc " BN_DPP_SWAP
      : "=&v"(U.v[0]), "=&v"(U.v[1]), "=&v"(U.v[2]), "=&v"(U.v[3]), "=&v"(U.v[4]), "=&v"(U.v[5]), "=&v"(U.v[6]), "=&v"(U.v[7]), "=&v"(U.v[8])
      : "v"(b.v[0]), "v"(b.v[1]), "v"(b.v[2]), "v"(b.v[3]), "v"(b.v[4]), "v"(b.v[5]), "v"(b.v[6]), "v"(b.v[7]), "v"(b.v[8])
      : "vcc");
  asm("s_nop 1\n\t"
      "v_mov_b32_dpp %0, %9 " BN_DPP_SWAP "\n\tv_mov_b32_dpp %1, %10 " BN_DPP_SWAP "\n\t"
      "v_mov_b32_dpp %2, %11 " BN_DPP_SWAP "\n\tv_mov_b32_dpp %3, %12 " BN_DPP_SWAP "\n\t"
      "v_mov_b32_dpp %4, %13 " BN_DPP_SWAP "\n\tv_mov_b32_dpp %5, %14 " BN_DPP_SWAP "\n\t"
      "v_mov_b32_dpp %6, %15 " BN_DPP_SWAP "\n\tv_mov_b32_dpp %7, %16 " BN_DPP_SWAP "\n\t"
      "v_mov_b32_dpp %8, %17 " BN_DPP_SWAP
      : "=&v"(W.v[0]), "=&v"(W.v[1]), "=&v"(W.v[2]), "=&v"(W.v[3]), "=&v"(W.v[4]), "=&v"(W.v[5]), "=&v"(W.v[6]), "=&v"(W.v[7]), "=&v"(W.v[8])
      : "v"(a.v[0]), "v"(a.v[1]), "v"(a.v[2]), "v"(a.v[3]), "v"(a.v[4]), "v"(a.v[5]), "v"(a.v[6]), "v"(a.v[7]), "v"(a.v[8]));
  // V in place on the b registers, c0 lanes only (bank_mask 0x5): first the partner's coordinate, then its negative.  (One instruction
  // cannot do both: in v_sub / v_subrev with DPP the permuted operand is always the minuend -- tools/ubench/dpp_probe2.hip.)
  const i32 zero = 0;
  asm("s_nop 1\n\t"
      "v_mov_b32_dpp %0, %0 " BN_DPP_SWAP_C0 "\n\tv_mov_b32_dpp %1, %1 " BN_DPP_SWAP_C0 "\n\t"
      "v_mov_b32_dpp %2, %2 " BN_DPP_SWAP_C0 "\n\tv_mov_b32_dpp %3, %3 " BN_DPP_SWAP_C0 "\n\t"
      "v_mov_b32_dpp %4, %4 " BN_DPP_SWAP_C0 "\n\tv_mov_b32_dpp %5, %5 " BN_DPP_SWAP_C0 "\n\t"
      "v_mov_b32_dpp %6, %6 " BN_DPP_SWAP_C0 "\n\tv_mov_b32_dpp %7, %7 " BN_DPP_SWAP_C0 "\n\t"
      "v_mov_b32_dpp %8, %8 " BN_DPP_SWAP_C0 "\n\t"
      "v_sub_u32_dpp %0, %9, %0 " BN_DPP_SELF_C0 "\n\tv_sub_u32_dpp %1, %9, %1 " BN_DPP_SELF_C0 "\n\t"
      "v_sub_u32_dpp %2, %9, %2 " BN_DPP_SELF_C0 "\n\tv_sub_u32_dpp %3, %9, %3 " BN_DPP_SELF_C0 "\n\t"
      "v_sub_u32_dpp %4, %9, %4 " BN_DPP_SELF_C0 "\n\tv_sub_u32_dpp %5, %9, %5 " BN_DPP_SELF_C0 "\n\t"
      "v_sub_u32_dpp %6, %9, %6 " BN_DPP_SELF_C0 "\n\tv_sub_u32_dpp %7, %9, %7 " BN_DPP_SELF_C0 "\n\t"
      "v_sub_u32_dpp %8, %9, %8 " BN_DPP_SELF_C0
      : "+v"(b.v[0]), "+v"(b.v[1]), "+v"(b.v[2]), "+v"(b.v[3]), "+v"(b.v[4]), "+v"(b.v[5]), "+v"(b.v[6]), "+v"(b.v[7]), "+v"(b.v[8])
      : "v"(zero));
}
BN_NOINLINE F29 w2_mul_leaf(i32 a0, i32 a1, i32 a2, i32 a3, i32 a4, i32 a5, i32 a6, i32 a7, i32 a8,
                            i32 b0, i32 b1, i32 b2, i32 b3, i32 b4, i32 b5, i32 b6, i32 b7, i32 b8) {
  const F29 a{{a0, a1, a2, a3, a4, a5, a6, a7, a8}};
  F29 b{{b0, b1, b2, b3, b4, b5, b6, b7, b8}}, U, W;
  w2_exchange(a, b, U, W);
  // c0: a0 b0 + a1 (-b1);  c1: a1 b0 + a0 b1
  return f29_dot2(a, U, W, b);
}
// the same product on the two-accumulator column form (f29_dot2_ilp): used by the G2 group law
BN_NOINLINE F29 w2_mul_ilp_leaf(i32 a0, i32 a1, i32 a2, i32 a3, i32 a4, i32 a5, i32 a6, i32 a7, i32 a8,
                                i32 b0, i32 b1, i32 b2, i32 b3, i32 b4, i32 b5, i32 b6, i32 b7, i32 b8) {
  const F29 a{{a0, a1, a2, a3, a4, a5, a6, a7, a8}};
  F29 b{{b0, b1, b2, b3, b4, b5, b6, b7, b8}}, U, W;
  w2_exchange(a, b, U, W);
  return f29_dot2_ilp(a, U, W, b);
}
// even lane: (a0 + a1)(a0 - a1);  odd lane: a0 * 2 a1.  Operand limbs non-negative (R / N).  Output N.
BN_NOINLINE F29 w2_sqr_leaf(i32 a0, i32 a1, i32 a2, i32 a3, i32 a4, i32 a5, i32 a6, i32 a7, i32 a8) {
  const F29 a{{a0, a1, a2, a3, a4, a5, a6, a7, a8}};
  const bool odd = lane_odd();
  const F29 o = xchg9(a);
  return f29_mul(sel9(odd, f29_add(a, o), o), sel9(odd, f29_sub(a, o), f29_dbl(a)));   // L(x) L(y) = 2
}
BN_DEV W2 w2_mul(const W2& a, const W2& b) { return W2{w2_mul_leaf(W_ARGS(a.c), W_ARGS(b.c))}; }
// the same product INLINED at its call site (no argument / result moves, no call): for the few routines whose whole loop stays inside the
// instruction cache with it (the cyclotomic squarings of the f^x chains)
BN_DEV W2 w2_mul_inl(const W2& a, const W2& bin) {
  F29 b = bin.c, U, W;
  w2_exchange(a.c, b, U, W);
  return W2{f29_dot2(a.c, U, W, b)};
}
BN_DEV W2 w2_mul_ilp(const W2& a, const W2& b) { return W2{w2_mul_ilp_leaf(W_ARGS(a.c), W_ARGS(b.c))}; }
BN_DEV W2 w2_sqr(const W2& a) { return W2{w2_sqr_leaf(W_ARGS(a.c))}; }
BN_DEV W2 w2_scale(const W2& a, const F29& k) { return W2{f29_mul_leaf(W_ARGS(a.c), W_ARGS(k))}; }

// ---- leaf PAIRS: two INDEPENDENT leaves named in one call (round 6) -------------------------------------------------------------------
// Every routine below names its product leaves two at a time where two are independent.  On lane pairs (the default build: BN_QUAD 0) the
// pair is evaluated one leaf after the other -- the same calls in the same order as before.  A translation unit compiled with BN_QUAD 1
// (plk_quad.hip: batches too small to fill the chip with one lane pair per element) gives every element a QUAD of lanes -- two lane pairs
// that hold the SAME state (replicated: same code, same inputs, hence the same digits) -- and splits each leaf pair between them: sub-pair 0
// forms the first leaf, sub-pair 1 the second, one DPP exchange (quad_perm [1,0,3,2]: lanes 0 <-> 1 and 6 <-> 7 of a group of eight are
// the c0 / c1 lanes of the two sub-pairs of one element, 2 <-> 3 and 4 <-> 5 those of the other) hands each the other's result:
// 18 selects + one leaf + 9 exchanges + 18 selects instead of two leaves.  The linear layer is replicated.  Same formulas, same operand
// classes, same integers into every Montgomery reduction: the quad build's values are digit for digit the lane-pair build's.
#ifndef BN_QUAD
#define BN_QUAD 0
#endif
#if BN_QUAD
// geometry of a quad (bn254_pair.hpp: lane l and 7 - l of a group of eight are partners; pairs 0 / 1 form one element, pairs 2 / 3 the other)
template <class T> BN_DEV int quad_pair(T t) { return (int)((t & 3) ^ (((t >> 2) & 1) ? 3 : 0)); }      // 0..3: the lane pair inside the group of eight
template <class T> BN_DEV T quad_index(T t) { return (T)(((t >> 3) << 1) | (T)(quad_pair(t) >> 1)); }   // element of thread t: 16 per wavefront
template <class T> BN_DEV int quad_sub(T t) { return quad_pair(t) & 1; }                                // which of the element's two lane pairs
BN_DEV bool quad_sub1() { return quad_sub(__lane_id()) != 0; }
BN_DEV F29 quad_xchg9(const F29& a) {                                   // the other sub-pair's lane of the same role
  F29 r;
#pragma unroll
  for (int i = 0; i < 9; ++i) r.v[i] = __builtin_amdgcn_mov_dpp(a.v[i], 0xB1, 0xF, 0xF, true);         // quad_perm [1,0,3,2]
  return r;
}
BN_DEV void quad_share(W2& r0, W2& r1, const F29& mine) {              // mine = leaf 0 on sub-pair 0, leaf 1 on sub-pair 1
  const bool s = quad_sub1();
  const F29 other = quad_xchg9(mine);
  r0 = W2{sel9(s, mine, other)};
  r1 = W2{sel9(s, other, mine)};
}
#endif
// On lane pairs the pair forms are MACROS that expand to the two original statements, operand expressions included: an inline function would
// evaluate the second leaf's operands before the first call and keep them live across it (measured on the ISA: +9 % moves in the Miller loop).
#if BN_QUAD
BN_DEV void w2_mul2(W2& r0, W2& r1, const W2& a0, const W2& b0, const W2& a1, const W2& b1) {
  const bool s = quad_sub1();
  quad_share(r0, r1, w2_mul(W2{sel9(s, a0.c, a1.c)}, W2{sel9(s, b0.c, b1.c)}).c);
}
BN_DEV void w2_sqr2(W2& r0, W2& r1, const W2& a0, const W2& a1) { quad_share(r0, r1, w2_sqr(W2{sel9(quad_sub1(), a0.c, a1.c)}).c); }
BN_DEV void w2_scale2(W2& r0, W2& r1, const W2& a0, const F29& k0, const W2& a1, const F29& k1) {
  const bool s = quad_sub1();
  quad_share(r0, r1, w2_scale(W2{sel9(s, a0.c, a1.c)}, sel9(s, k0, k1)).c);
}
#define W2_MUL2(r0, r1, a0, b0, a1, b1) w2_mul2(r0, r1, a0, b0, a1, b1)
#define W2_MUL2_SEL(INL, r0, r1, a0, b0, a1, b1) w2_mul2(r0, r1, a0, b0, a1, b1)
#define W2_SQR2(r0, r1, a0, a1) w2_sqr2(r0, r1, a0, a1)
// a product and a square: the square goes through the product leaf as a * a -- the same integer into the same reduction
// ((a0 + a1)(a0 - a1) = a0^2 - a1^2 and a0 * 2 a1 = a1 a0 + a0 a1 exactly), so the digits are those of the squaring leaf
#define W2_MUL_SQR(r0, r1, a0, b0, a1) do { const W2 sq_ = (a1); w2_mul2(r0, r1, a0, b0, sq_, sq_); } while (0)
#define W2_SQR_MUL(r0, r1, a0, a1, b1) do { const W2 sq_ = (a0); w2_mul2(r0, r1, sq_, sq_, a1, b1); } while (0)
#define W2_SCALE2(r0, r1, a0, k0, a1, k1) w2_scale2(r0, r1, a0, k0, a1, k1)
#else
#define W2_MUL2(r0, r1, a0, b0, a1, b1) do { (r0) = w2_mul(a0, b0); (r1) = w2_mul(a1, b1); } while (0)
#define W2_MUL2_SEL(INL, r0, r1, a0, b0, a1, b1) do { (r0) = w2_mul_sel<INL>(a0, b0); (r1) = w2_mul_sel<INL>(a1, b1); } while (0)
#define W2_SQR2(r0, r1, a0, a1) do { (r0) = w2_sqr(a0); (r1) = w2_sqr(a1); } while (0)
#define W2_MUL_SQR(r0, r1, a0, b0, a1) do { (r0) = w2_mul(a0, b0); (r1) = w2_sqr(a1); } while (0)
#define W2_SQR_MUL(r0, r1, a0, a1, b1) do { (r0) = w2_sqr(a0); (r1) = w2_mul(a1, b1); } while (0)
#define W2_SCALE2(r0, r1, a0, k0, a1, k1) do { (r0) = w2_scale(a0, k0); (r1) = w2_scale(a1, k1); } while (0)
#endif

// ---- lazy linear layer ------------------------------------------------------------------------------------------------
BN_DEV W2 w2_add(const W2& a, const W2& b) { return W2{f29_add(a.c, b.c)}; }
BN_DEV W2 w2_sub(const W2& a, const W2& b) { return W2{f29_sub(a.c, b.c)}; }
BN_DEV W2 w2_neg(const W2& a) { return W2{f29_neg(a.c)}; }
BN_DEV W2 w2_norm(const W2& a) { return W2{f29_norm(a.c)}; }
BN_DEV W2 w2_reduce(const W2& a) { return W2{f29_reduce(a.c)}; }
// reduce(ka a + kb b)
BN_DEV W2 w2_lin2(const W2& a, int ka, const W2& b, int kb) { return W2{f29_lin2(a.c, ka, b.c, kb)}; }
// reduce(k xi x + m y): this lane's coordinate of xi x is 9 x -/+ (partner's x)
BN_DEV W2 w2_xi_lin(const W2& x, int k, const W2& y, int m) {
  const F29 xo = xchg9(x.c);
  if (m == 0) {                                   // a compile-time constant at every call site
    const F29* const t[2] = {&x.c, &xo};
    const i32 c[2] = {bn_keep(9 * k), bn_keep_v(lane_odd() ? k : -k)};
    return W2{f29_reduce_terms(t, c)};
  }
  const F29* const t[3] = {&x.c, &xo, &y.c};
  const i32 c[3] = {bn_keep(9 * k), bn_keep_v(lane_odd() ? k : -k), bn_keep(m)};
  return W2{f29_reduce_terms(t, c)};
}

// ---- conversions (saturated lane-pair <-> carry-free lane-pair) ---------------------------------------------------
BN_DEV W2 w2_from_s2(const S2& a) { return W2{f29_reduce(f29_from_fp(a.c))}; }     // R
BN_DEV S2 w2_to_s2(const W2& a) { return S2{f29_to_fp(a.c)}; }                     // needs |V| < 64
BN_DEV W2 w2_const(const uint32_t (&c)[2][8]) { return w2_from_s2(s2_const(c)); }
BN_DEV void w12_from_s12(W12& r, const S12& a) {
  r.c0.c0 = w2_from_s2(a.c0.c0); r.c0.c1 = w2_from_s2(a.c0.c1); r.c0.c2 = w2_from_s2(a.c0.c2);
  r.c1.c0 = w2_from_s2(a.c1.c0); r.c1.c1 = w2_from_s2(a.c1.c1); r.c1.c2 = w2_from_s2(a.c1.c2);
}
BN_DEV void w12_to_s12(S12& r, const W12& a) {
  r.c0.c0 = w2_to_s2(a.c0.c0); r.c0.c1 = w2_to_s2(a.c0.c1); r.c0.c2 = w2_to_s2(a.c0.c2);
  r.c1.c0 = w2_to_s2(a.c1.c0); r.c1.c1 = w2_to_s2(a.c1.c1); r.c1.c2 = w2_to_s2(a.c1.c2);
}

// ---- Fp6: inputs R / N / D with |V| <= 4, outputs R ---------------------------------------------------------------------
// The six Karatsuba products and their LAZY recombination (32-bit limb sums, no pass over them yet).  SUBTRACTIVE Karatsuba (round 4):
//   a_i b_j + a_j b_i = a_i b_i + a_j b_j - (a_i - a_j)(b_i - b_j)
// the difference of two N-class values is a D-class value (|limbs| < 2^29) and goes into the product leaf as it is, where the additive
// form had to carry-normalise every pre-addition (25 instructions each, six per Fp6 product).  Operands therefore R / N (limbs in [0, 2^29)).
//   c0 = v0 + xi x0,  c1 = y1 + xi v2,  c2 = y2     with  v0, v2 in [0, 2^29),  x0, y1 in (-2^29, 2^30),  y2 in (-2^29, 2^30 + 2^29).
// w6_mul finishes each coefficient with one reduce pass; a caller that only adds further terms to a coefficient (the second Fp6 product of an
// Fp12 squaring, the third of an Fp12 product) folds them into the SAME pass instead of reducing twice.
struct W6Raw { W2 v0, x0, y1, v2, y2; };
template <bool INL = false> BN_DEV W2 w2_mul_sel(const W2& a, const W2& b) { return INL ? w2_mul_inl(a, b) : w2_mul(a, b); }
template <bool INL = false>
BN_DEV W6Raw w6_mul_raw(const W6& a, const W6& b) {
  W2 v0, v1, v2, s0, s1, s2;
  W2_MUL2_SEL(INL, v0, v1, a.c0, b.c0, a.c1, b.c1);
  W2_MUL2_SEL(INL, v2, s0, a.c2, b.c2, w2_sub(a.c1, a.c2), w2_sub(b.c1, b.c2));
  W2_MUL2_SEL(INL, s1, s2, w2_sub(a.c0, a.c1), w2_sub(b.c0, b.c1), w2_sub(a.c0, a.c2), w2_sub(b.c0, b.c2));
  return W6Raw{v0, w2_sub(w2_add(v1, v2), s0), w2_sub(w2_add(v0, v1), s1), v2, w2_add(w2_sub(w2_add(v0, v2), s2), v1)};
}
template <bool INL = false>
BN_DEV W6 w6_mul(const W6& a, const W6& b) {
  const W6Raw m = w6_mul_raw<INL>(a, b);
  W6 r;
  r.c0 = w2_xi_lin(m.x0, 1, m.v0, 1);                                   // v0 + xi (t0 - v1 - v2)
  r.c1 = w2_xi_lin(m.v2, 1, m.y1, 1);                                   // (t1 - v0 - v1) + xi v2
  r.c2 = w2_reduce(m.y2);                                               // t2 - v0 - v2 + v1
  return r;
}
BN_DEV W6 w6_add_norm(const W6& a, const W6& b) {
  return W6{w2_norm(w2_add(a.c0, b.c0)), w2_norm(w2_add(a.c1, b.c1)), w2_norm(w2_add(a.c2, b.c2))};
}
BN_DEV W6 w6_sub(const W6& a, const W6& b) { return W6{w2_sub(a.c0, b.c0), w2_sub(a.c1, b.c1), w2_sub(a.c2, b.c2)}; }

// ---- Fp12: inputs R / N, outputs R / N --------------------------------------------------------------------------------
// c1 = t2 - t0 - t1 with the third Fp6 product left raw: every coefficient of c1 is ONE pass over (raw piece of t2) - t0.ci - t1.ci
// (limb ranges: v0 - t0.c0 - t1.c0 in (-2^30, 2^29), y1 - t0.c1 - t1.c1 in (-2^30 - 2^29, 2^30), y2 - t0.c2 in (-2^30, 2^30 + 2^29): all int32)
BN_DEV void w12_mul_c1(W12& r, const W6Raw& t2, const W6& t0, const W6& t1) {
  r.c1.c0 = w2_xi_lin(t2.x0, 1, w2_sub(w2_sub(t2.v0, t0.c0), t1.c0), 1);                  // the sums of t2's operands are N-class: w6_add_norm
  r.c1.c1 = w2_xi_lin(t2.v2, 1, w2_sub(w2_sub(t2.y1, t0.c1), t1.c1), 1);
  r.c1.c2 = w2_lin2(w2_sub(t2.y2, t0.c2), 1, t1.c2, -1);
}
BN_DEV W12 w12_mul(const W12& a, const W12& b) {
  const W6 t0 = w6_mul(a.c0, b.c0);
  const W6 t1 = w6_mul(a.c1, b.c1);
  const W6Raw t2 = w6_mul_raw(w6_add_norm(a.c0, a.c1), w6_add_norm(b.c0, b.c1));
  W12 r;
  w12_mul_c1(r, t2, t0, t1);
  r.c0.c0 = w2_xi_lin(t1.c2, 1, t0.c0, 1);                               // t0 + v t1
  r.c0.c1 = w2_norm(w2_add(t0.c1, t1.c0));                               // two R values: N with |V| < 1.1
  r.c0.c2 = w2_norm(w2_add(t0.c2, t1.c1));
  return r;
}
// a * b for b with b.c1.c2 = 0 -- the product of two lines (w12_line_product): the middle Fp6 product has a zero coefficient on its
// right-hand side, 5 products instead of 6 (17 for the Fp12 product instead of 18)
BN_DEV W6 w6_mul_b2zero(const W6& a, const W6& b) {
  // (a0 + a1 v + a2 v^2)(b0 + b1 v): v^0: a0 b0 + xi a2 b1, v^1: a0 b1 + a1 b0, v^2: a1 b1 + a2 b0 -- the two products with a2 directly, the
  // cross term by subtractive Karatsuba (a0 - a1)(b0 - b1): five products, no carry-normalised pre-additions.  Operands R / N.
  W2 v0, v1, p21, p20;
  W2_MUL2(v0, v1, a.c0, b.c0, a.c1, b.c1);
  W2_MUL2(p21, p20, a.c2, b.c1, a.c2, b.c0);
  const W2 s1 = w2_mul(w2_sub(a.c0, a.c1), w2_sub(b.c0, b.c1));
  W6 r;
  r.c0 = w2_xi_lin(p21, 1, v0, 1);                                       // v0 + xi a2 b1
  r.c1 = w2_reduce(w2_sub(w2_add(v0, v1), s1));                          // a0 b1 + a1 b0
  r.c2 = w2_norm(w2_add(p20, v1));                                       // a2 b0 + a1 b1: two N values
  return r;
}
BN_DEV W12 w12_mul_line_pair(const W12& a, const W12& b) {
  const W6 t0 = w6_mul(a.c0, b.c0);
  const W6 t1 = w6_mul_b2zero(a.c1, b.c1);
  const W6Raw t2 = w6_mul_raw(w6_add_norm(a.c0, a.c1), W6{w2_norm(w2_add(b.c0.c0, b.c1.c0)), w2_norm(w2_add(b.c0.c1, b.c1.c1)), b.c0.c2});
  W12 r;
  w12_mul_c1(r, t2, t0, t1);
  r.c0.c0 = w2_xi_lin(t1.c2, 1, t0.c0, 1);                               // t0 + v t1
  r.c0.c1 = w2_norm(w2_add(t0.c1, t1.c0));
  r.c0.c2 = w2_norm(w2_add(t0.c2, t1.c1));
  return r;
}
// complex squaring (fp12.rs:536-550): c0 = (a0 - a1)(a0 - v a1) + a0 a1 + v a0 a1, c1 = 2 a0 a1.  The product m = (a0 - a1)(a0 - v a1)
// stays raw: each coefficient of c0 is ONE pass over m's piece plus the terms of c2 = a0 a1 (limb ranges: x0 + c2.c2 in (-2^29, 2^30 + 2^29), v0 + c2.c0 in
// [0, 2^30), y1 + c2.c1 + c2.c0 in (-2^29, 2^31): as two terms; y2 and c2.c2 + c2.c1 as two terms)
template <bool INL = false>
BN_DEV W12 w12_sqr(const W12& a) {
  // both factors of m carry-normalised (N): the subtractive Karatsuba inside w6_mul_raw takes differences of their coefficients
  const W6 d{w2_norm(w2_sub(a.c0.c0, a.c1.c0)), w2_norm(w2_sub(a.c0.c1, a.c1.c1)), w2_norm(w2_sub(a.c0.c2, a.c1.c2))};
  W6 e;                                                                  // a0 - v a1 = (a0.c0 - xi a1.c2, a0.c1 - a1.c0, a0.c2 - a1.c1)
  e.c0 = w2_xi_lin(a.c1.c2, -1, a.c0.c0, 1);
  e.c1 = w2_norm(w2_sub(a.c0.c1, a.c1.c0));
  e.c2 = w2_norm(w2_sub(a.c0.c2, a.c1.c1));
  const W6 c2 = w6_mul<INL>(a.c0, a.c1);
  const W6Raw m = w6_mul_raw<INL>(d, e);
  W12 r;
  r.c1.c0 = w2_norm(w2_add(c2.c0, c2.c0));
  r.c1.c1 = w2_norm(w2_add(c2.c1, c2.c1));
  r.c1.c2 = w2_norm(w2_add(c2.c2, c2.c2));
  r.c0.c0 = w2_xi_lin(w2_add(m.x0, c2.c2), 1, w2_add(m.v0, c2.c0), 1);   // m0 + c2.0 + xi c2.2
  {                                                                      // m1 + c2.1 + c2.0 = xi v2 + y1 + (c2.1 + c2.0)
    const F29 xo = xchg9(m.v2.c);
    const W2 cc = w2_add(c2.c1, c2.c0);
    const F29* const t[4] = {&m.v2.c, &xo, &m.y1.c, &cc.c};
    const i32 c[4] = {bn_keep(9), bn_keep_v(lane_odd() ? 1 : -1), bn_keep(1), bn_keep(1)};
    r.c0.c1 = W2{f29_reduce_terms(t, c)};
  }
  r.c0.c2 = w2_lin2(m.y2, 1, w2_add(c2.c2, c2.c1), 1);                   // m2 + c2.2 + c2.1
  return r;
}
// conjugate, N-class output (non-negative limbs: the result may feed a squaring)
BN_DEV W12 w12_conj(const W12& a) {
  W12 r;
  r.c0 = a.c0;
  r.c1.c0 = w2_norm(w2_neg(a.c1.c0)); r.c1.c1 = w2_norm(w2_neg(a.c1.c1)); r.c1.c2 = w2_norm(w2_neg(a.c1.c2));
  return r;
}
// this lane's coordinate negated on odd lanes only (Fp2 conjugation), N-class
BN_DEV W2 w2_conj(const W2& a) { return W2{sel9(lane_odd(), a.c, f29_norm(f29_neg(a.c)))}; }
template <int E>
BN_DEV W6 w6_frobenius(const W6& a) {
  constexpr bool oddE = (E & 1) != 0;
  const uint32_t (&k1)[2][8] = (E == 1) ? C_FROB6_C1_1 : (E == 2) ? C_FROB6_C1_2 : C_FROB6_C1_3;
  const uint32_t (&k2)[2][8] = (E == 1) ? C_FROB6_C2_1 : (E == 2) ? C_FROB6_C2_2 : C_FROB6_C2_3;
  W6 r;
  r.c0 = oddE ? w2_conj(a.c0) : a.c0;
  W2_MUL2(r.c1, r.c2, oddE ? w2_conj(a.c1) : a.c1, w2_const(k1), oddE ? w2_conj(a.c2) : a.c2, w2_const(k2));
  return r;
}
template <int E>
BN_DEV W12 w12_frobenius(const W12& a) {
  const uint32_t (&k)[2][8] = (E == 1) ? C_FROB12_C1_1 : (E == 2) ? C_FROB12_C1_2 : C_FROB12_C1_3;
  const W2 kk = w2_const(k);
  const W6 x1 = w6_frobenius<E>(a.c1);
  W6 y;
  W2_MUL2(y.c0, y.c1, x1.c0, kk, x1.c1, kk);
  y.c2 = w2_mul(x1.c2, kk);
  return W12{w6_frobenius<E>(a.c0), y};
}
// fp12.rs:426-503 (mul_by_024): x0 = ell_0, x2 = ell_vv, x4 = ell_vw, all R / N.  f R / N.  Output R.
BN_DEV W12 w12_sparse_mul(const W12& f, const W2& x0, const W2& x4, const W2& x2) {
  const W2 z0 = f.c0.c0, z1 = f.c0.c1, z2 = f.c0.c2, z3 = f.c1.c0, z4 = f.c1.c1, z5 = f.c1.c2;
  W2 d0, d2, d4, p12, p54, p10, p34, p30, p52, q02, q24, q04;
  W2_MUL2(d0, d2, z0, x0, z2, x2);
  W2_MUL2(d4, p12, z4, x4, z1, x2);
  W2_MUL2(p54, p10, z5, x4, z1, x0);
  W2_MUL2(p34, p30, z3, x4, z3, x0);
  // the three cross sums z_i x_j + z_j x_i by subtractive Karatsuba: d_i + d_j - (z_i - z_j)(x_i - x_j), differences lazy (D-class operands)
  W2_MUL2(p52, q02, z5, x2, w2_sub(z0, z2), w2_sub(x0, x2));
  W2_MUL2(q24, q04, w2_sub(z2, z4), w2_sub(x2, x4), w2_sub(z0, z4), w2_sub(x0, x4));
  const W2 qs = w2_mul(w2_norm(w2_add(w2_add(z1, z3), z5)), w2_norm(w2_add(w2_add(x0, x2), x4)));
  W12 o;
  o.c0.c0 = w2_xi_lin(w2_add(p12, d4), 1, d0, 1);                                            // xi (z1 x2 + d4) + d0
  o.c0.c1 = w2_xi_lin(w2_add(p54, d2), 1, p10, 1);                                           // xi (z5 x4 + d2) + z1 x0
  o.c0.c2 = w2_reduce(w2_add(w2_sub(w2_add(d0, d2), q02), p34));                             // z0 x2 + z2 x0 + z3 x4          limbs in (-2^29, 2^30 + 2^29)
  o.c1.c0 = w2_xi_lin(w2_sub(w2_add(d2, d4), q24), 1, p30, 1);                               // xi (z2 x4 + z4 x2) + z3 x0
  o.c1.c1 = w2_xi_lin(p52, 1, w2_sub(w2_add(d0, d4), q04), 1);                               // xi z5 x2 + z0 x4 + z4 x0
  {                                                                                          // (z1+z3+z5)(x0+x2+x4) - all six cross products
    const W2 sa = w2_add(w2_add(p12, p54), p10), sb = w2_add(w2_add(p34, p30), p52);         // each < 3 * 2^29: fits int32
    const F29* const t[3] = {&qs.c, &sa.c, &sb.c};
    const i32 c[3] = {bn_keep(1), bn_keep(-1), bn_keep(-1)};
    o.c1.c2 = W2{f29_reduce_terms(t, c)};
  }
  return o;
}
// f * (u + x2 v^2 + x4 v w) with u in {0, 1} (a lane-dependent int): the same formulas with x0 = u, whose products are copies.
// Ten Fp2 products: z1 x4 + z3 x2 comes from (z1 + z3)(x2 + x4) - z1 x2 - z3 x4.  f R / N, x2 / x4 R / N.  Output R.
BN_DEV W2 w2_xi_lin_v(const W2& x, const W2& y, i32 m) {                // reduce(xi x + m y), m lane-dependent
  const F29 xo = xchg9(x.c);
  const F29* const t[3] = {&x.c, &xo, &y.c};
  const i32 c[3] = {bn_keep(9), bn_keep_v(lane_odd() ? 1 : -1), bn_keep_v(m)};
  return W2{f29_reduce_terms(t, c)};
}
BN_DEV W2 w2_lin_v(const W2& x, const W2& y, i32 m) {                   // reduce(x + m y)
  const F29* const t[2] = {&x.c, &y.c};
  const i32 c[2] = {bn_keep(1), bn_keep_v(m)};
  return W2{f29_reduce_terms(t, c)};
}
BN_DEV W12 w12_sparse_mul_unit(const W12& f, i32 u, const W2& x4, const W2& x2) {
  const W2 z0 = f.c0.c0, z1 = f.c0.c1, z2 = f.c0.c2, z3 = f.c1.c0, z4 = f.c1.c1, z5 = f.c1.c2;
  W2 d2, d4, p12, p54, p34, p52, a02, a04, q24, q13;
  W2_MUL2(d2, d4, z2, x2, z4, x4);
  W2_MUL2(p12, p54, z1, x2, z5, x4);
  W2_MUL2(p34, p52, z3, x4, z5, x2);
  W2_MUL2(a02, a04, z0, x2, z0, x4);
  const W2 x24 = w2_sub(x2, x4);                                          // lazy difference (D-class): subtractive Karatsuba, see w6_mul_raw
  W2_MUL2(q24, q13, w2_sub(z2, z4), x24, w2_sub(z1, z3), x24);            // d2 + d4 - (z2 x4 + z4 x2);  p12 + p34 - (z1 x4 + z3 x2)
  W12 o;
  o.c0.c0 = w2_xi_lin_v(w2_add(p12, d4), z0, u);                         // xi (z1 x2 + z4 x4) + u z0
  o.c0.c1 = w2_xi_lin_v(w2_add(p54, d2), z1, u);                         // xi (z5 x4 + z2 x2) + u z1
  o.c0.c2 = w2_lin_v(w2_add(a02, p34), z2, u);                           // z0 x2 + z3 x4 + u z2
  o.c1.c0 = w2_xi_lin_v(w2_sub(w2_add(d2, d4), q24), z3, u);             // xi (z2 x4 + z4 x2) + u z3
  {                                                                      // xi z5 x2 + z0 x4 + u z4
    const F29 xo = xchg9(p52.c);
    const F29* const t[4] = {&p52.c, &xo, &a04.c, &z4.c};
    const i32 c[4] = {bn_keep(9), bn_keep_v(lane_odd() ? 1 : -1), bn_keep(1), bn_keep_v(u)};
    o.c1.c1 = W2{f29_reduce_terms(t, c)};
  }
  o.c1.c2 = w2_lin_v(w2_sub(w2_add(p12, p34), q13), z5, u);              // z1 x4 + z3 x2 + u z5
  return o;
}
// pairing.rs:274-350 (Granger-Scott), input R / N with |V| <= 1.2, output R.
// Fp4 squaring (a + b s)^2, s^2 = xi: c0 = a^2 + xi b^2, c1 = 2 a b.  Written with TWO products instead of three squarings:
// m = a b, w = (a + b)(a + xi b), c0 = w - m - xi m; on this core a lane-pair squaring is 162 multiply-adds and a product 243, so
// 2 x 243 beats 3 x 162.  The linear layer is ONE pass per output (round 4; before: c0 was reduced and the output combination 3 c0 - 2 z
// reduced it again):
//   w = (a - b)(a - xi b) = a^2 + xi b^2 - (1 + xi) a b: the SUBTRACTIVE form -- a - b is a lazy difference (D-class operand, no carry
//   pass), a - xi b only carry-normalised (f29_norm_terms: a product operand with |V| <= 5.6, no multiple of p needs to go)
//   3 c0 - 2 z = 3 w + 30 m_own -/+ 3 m_partner - 2 z      (c0 = w + m + xi m;  xi m on this lane = 9 m_own -/+ m_partner)  one reduce pass
// Returns m (= c1 / 2: the callers fold the factor into their own combination) and that output.  a, b: R / N (limbs in [0, 2^29)).
BN_DEV W2 w2_xi_norm(const W2& x, const W2& y) {                      // norm(xi x + y), N-class (the wide routines keep the additive form)
  const F29 xo = xchg9(x.c);
  const F29* const t[3] = {&x.c, &xo, &y.c};
  const i32 c[3] = {bn_keep(9), bn_keep_v(lane_odd() ? 1 : -1), bn_keep(1)};
  return W2{f29_norm_terms(t, c)};
}
BN_DEV W2 w2_xi_norm_sub(const W2& x, const W2& y) {                  // norm(y - xi x), N-class
  const F29 xo = xchg9(x.c);
  const F29* const t[3] = {&x.c, &xo, &y.c};
  const i32 c[3] = {bn_keep(-9), bn_keep_v(lane_odd() ? -1 : 1), bn_keep(1)};
  return W2{f29_norm_terms(t, c)};
}
template <bool INL = false>
BN_DEV void w_fp4_square_fold(W2& out0, W2& m, const W2& a, const W2& b, const W2& z) {      // out0 = reduce(3 (a^2 + xi b^2) - 2 z), m = a b
  W2 w;
  W2_MUL2_SEL(INL, m, w, a, b, w2_sub(a, b), w2_xi_norm_sub(b, a));
  const F29 mo = xchg9(m.c);
  const F29* const t[4] = {&w.c, &m.c, &mo, &z.c};
  const i32 c[4] = {bn_keep(3), bn_keep(30), bn_keep_v(lane_odd() ? 3 : -3), bn_keep(-2)};
  out0 = W2{f29_reduce_terms(t, c)};
}
template <bool INL = false>
BN_DEV W12 w12_cyclotomic_sqr(const W12& f) {
  const W2 z0 = f.c0.c0, z4 = f.c0.c1, z3 = f.c0.c2, z2 = f.c1.c0, z1 = f.c1.c1, z5 = f.c1.c2;
  W2 m;
  W12 r;
  w_fp4_square_fold<INL>(r.c0.c0, m, z0, z1, z0);          // z0' = 3 t0 - 2 z0
  r.c1.c1 = w2_lin2(m, 6, z1, 2);                          // z1' = 3 t1 + 2 z1,  t1 = 2 m
  w_fp4_square_fold<INL>(r.c0.c1, m, z2, z3, z4);          // z4' = 3 t0 - 2 z4
  r.c1.c2 = w2_lin2(m, 6, z5, 2);                          // z5' = 3 t1 + 2 z5
  w_fp4_square_fold<INL>(r.c0.c2, m, z4, z5, z3);          // z3' = 3 t2 - 2 z3
  r.c1.c0 = w2_xi_lin(m, 6, z2, 2);                        // z2' = 3 xi t3 + 2 z2
  return r;
}

// ---- G2 doubling / addition steps on the carry-free core --------------------------------------------------------------
// value / 2 mod p for an N-class value: add p when odd, renormalise and shift (|V| <= (|V(a)| + 1) / 2), N-class result
BN_DEV F29 f29_halve(const F29& a) {
  i32 p[9]; f29_p(p);
  const i32 m = -(a.v[0] & 1);                 // all-ones when odd
  i32 lo[9];
  i32 c = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    i32 t = a.v[i] + (p[i] & m) + c;
    lo[i] = t & BN_M29;
    c = t >> 29;
  }
  lo[8] = a.v[8] + (p[8] & m) + c;
  F29 r;
#pragma unroll
  for (int i = 0; i < 8; ++i) r.v[i] = (lo[i] >> 1) | ((lo[i + 1] & 1) << 28);
  r.v[8] = lo[8] >> 1;
  return r;
}
BN_DEV W2 w2_halve(const W2& a) { return W2{f29_halve(a.c)}; }
BN_DEV W2 w2_triple(const W2& a) { return W2{f29_add(f29_add(a.c, a.c), a.c)}; }   // lazy, limbs < 3 * 2^29 for N input
struct G2W { W2 x, y, z; };      // coordinates R / N with |V| <= 2
// The twist constant b' = 3 / (9 + u) as R-class lane-pair digits -- exactly what w2_const(C_TWIST_B) evaluates to (b' 2^261 mod p,
// balanced) -- materialised with one select per limb where it is used (it is the first operand of a product: the selects replace
// the argument moves) instead of being kept live, or spilled, across the Miller loops.
BN_DEV W2 w2_twist_b() {
  const F29 k0{{0x12cb9911, 0x1fcb719e, 0x1368f727, 0x1ff96726, 0x0aadc75b, 0x179b4484, 0x06475c9b, 0x020e578d, -1520355}};
  const F29 k1{{0x0322ef99, 0x06e86c1b, 0x13f74e78, 0x0a0a4222, 0x0bcef269, 0x0af6e9ff, 0x0644c3a7, 0x0acd7c01, -535829}};
  return W2{sel9(lane_odd(), k0, k1)};
}
// pairing.rs:798-818.  Line coefficients: l0 R, l1 D, l2 N.
// ---- the isomorphic pair of curves for kernels whose Miller value goes straight into the final exponentiation ---------------
// phi(x, y) = (s^2 x, s^3 y) with s in Fp, s^6 = 82 / 3, maps E: y^2 = x^3 + 3 onto y^2 = x^3 + 82 and the twist E' onto
// E'': y^2 = x^3 + (9 - u) (plk_group.hip uses the same map for the G2 group law).  Run on (phi P, phi Q) with b'' = 9 - u the doubling and
// addition steps give phi of the same points up to projective factors in Fp, and every line is the original line times a factor in Fp:
// X, Y, Z of the stepped point carry weights s^2, s^3, 1, all three line coefficients weight s^6 -- so the Miller value differs from the
// reference's by an element of Fp*, which the final exponentiation kills (c^(p^6 - 1) = 1).  pairing() and every verdict are unchanged;
// what it buys is the product b' * 3c of every doubling step (a full leaf, b' = 3 / (9 + u) is generic) becoming the two-term reduce pass
// (27 - 3 u) c.  NOT used where the raw Miller value is the result (miller_loop_batch, glued_miller_loop_batch, the line tables).
BN_DEV F29 f29_iso_s2() { return F29{{0x05beeef0, 0x1f76bf90, 0x1d5e46cf, 0x17f6764e, 0x1df385e5, 0x0d7a8334, 0x152215eb, 0x01b6eac1, -290196}}; }   // s^2 2^261 mod p, balanced
BN_DEV F29 f29_iso_s3() { return F29{{0x1af1f8a3, 0x00cd9858, 0x1dce6a34, 0x142e620a, 0x1bc0c667, 0x0ae94d20, 0x0db9310b, 0x12572b72, 0x000405e6}}; }   // s^3
BN_DEV W2 w2_mul_27m3u(const W2& a) {                                    // (a0 + a1 u)(27 - 3 u) = (27 a0 + 3 a1) + (27 a1 - 3 a0) u, R-class; any 32-bit limbs
  const F29 ao = xchg9(a.c);
  const F29* const t[2] = {&a.c, &ao};
  const i32 c[2] = {bn_keep(27), bn_keep_v(lane_odd() ? -3 : 3)};
  return W2{f29_reduce_terms(t, c)};
}
template <bool ISO = false>
BN_DEV void g2_doubling_step29(G2W& r, W2& l0, W2& l1, W2& l2) {
#if BN_QUAD
  // the nine (ten) leaves in five rounds: (X Y, X^2) (Y^2, Z^2) ((Y + Z)^2 [, b' 3c]) (b h, a (b - f)) (g^2, e^2)
  W2 xy, xx, b, c, yz2, e;
  W2_MUL_SQR(xy, xx, r.x, r.y, r.x);
  const W2 a = w2_halve(xy);                                             // N
  l2 = w2_norm(w2_triple(xx));                                           // 3 X^2, N, |V| < 3.4
  W2_SQR2(b, c, r.y, r.z);
  if (ISO) {
    yz2 = w2_sqr(w2_norm(w2_add(r.y, r.z)));
    e = w2_mul_27m3u(c);                                                 // E'': (9 - u) * 3c, N
  } else {
    W2_SQR_MUL(yz2, e, w2_norm(w2_add(r.y, r.z)), w2_twist_b(), w2_norm(w2_triple(c)));   // b' * 3c, N
  }
  const W2 h = w2_norm(w2_sub(yz2, w2_add(b, c)));                       // (Y+Z)^2 - (b+c), N, |V| < 3.4
  l1 = w2_neg(h);                                                        // D
  l0 = w2_xi_lin(w2_sub(e, b), 1, b, 0);                                 // xi (e - b), R
  const W2 f = w2_norm(w2_triple(e));                                    // 3e, N, |V| < 3.4
  W2_MUL2(r.z, r.x, b, h, a, w2_sub(b, f));                              // Z3 = b h;  X3 = a (b - f): D operand
  const W2 g = w2_halve(w2_norm(w2_add(b, f)));                          // (b + f) / 2, N
  W2 gg, ee;
  W2_SQR2(gg, ee, g, e);
  r.y = w2_lin2(gg, 1, ee, -3);                                          // g^2 - 3 e^2, R
#else
  const W2 a = w2_halve(w2_mul(r.x, r.y));                              // N
  l2 = w2_norm(w2_triple(w2_sqr(r.x)));                                  // 3 X^2, N, |V| < 3.4
  const W2 b = w2_sqr(r.y);
  const W2 c = w2_sqr(r.z);
  const W2 h = w2_norm(w2_sub(w2_sqr(w2_norm(w2_add(r.y, r.z))), w2_add(b, c)));   // (Y+Z)^2 - (b+c), N, |V| < 3.4
  const W2 e = ISO ? w2_mul_27m3u(c) : w2_mul(w2_twist_b(), w2_norm(w2_triple(c)));   // b' * 3c (E'': (9 - u) * 3c), N
  l1 = w2_neg(h);                                                        // D
  r.z = w2_mul(b, h);
  l0 = w2_xi_lin(w2_sub(e, b), 1, b, 0);                                 // xi (e - b), R
  const W2 f = w2_norm(w2_triple(e));                                    // 3e, N, |V| < 3.4
  r.x = w2_mul(a, w2_sub(b, f));                                         // a (b - f): D operand
  const W2 g = w2_halve(w2_norm(w2_add(b, f)));                          // (b + f) / 2, N
  r.y = w2_lin2(w2_sqr(g), 1, w2_sqr(e), -3);                            // g^2 - 3 e^2, R
#endif
}
// pairing.rs:756-772, Q = (bx, by) affine, R-class.  Line coefficients: l0 R, l1 D, l2 D.
BN_DEV void g2_addition_step29(G2W& r, const W2& bx, const W2& by, W2& l0, W2& l1, W2& l2) {
#if BN_QUAD
  // the thirteen leaves in seven rounds: (z bx, z by) (e bx, d by) (dn^2, en^2) (dn f, x f) (z e^2, z h) (dn j, h y) (en (i - j))
  W2 zbx, zby, ebx, dby;
  W2_MUL2(zbx, zby, r.z, bx, r.z, by);
  const W2 d = w2_sub(r.x, zbx);                                         // D
  const W2 e = w2_sub(r.y, zby);                                         // D
  W2_MUL2(ebx, dby, e, bx, d, by);
  l0 = w2_xi_lin(w2_sub(ebx, dby), 1, bx, 0);                            // xi (e bx - d by), R
  l1 = d;
  l2 = w2_neg(e);
  const W2 dn = w2_norm(d), en = w2_norm(e);                             // N (squarings need non-negative limbs)
  W2 f, ee, h, i, zee, zh, dj, hy;
  W2_SQR2(f, ee, dn, en);
  W2_MUL2(h, i, dn, f, r.x, f);
  W2_MUL2(zee, zh, r.z, ee, r.z, h);
  const W2 j = w2_norm(w2_sub(w2_add(zee, h), w2_add(i, i)));            // z e^2 + h - 2i, N, |V| < 5
  W2_MUL2(dj, hy, dn, j, h, r.y);
  r.z = zh;
  r.x = dj;
  r.y = w2_norm(w2_sub(w2_mul(en, w2_sub(i, j)), hy));                   // e (i - j) - h y, N
#else
  const W2 d = w2_sub(r.x, w2_mul(r.z, bx));                             // D
  const W2 e = w2_sub(r.y, w2_mul(r.z, by));                             // D
  l0 = w2_xi_lin(w2_sub(w2_mul(e, bx), w2_mul(d, by)), 1, bx, 0);        // xi (e bx - d by), R
  l1 = d;
  l2 = w2_neg(e);
  const W2 dn = w2_norm(d), en = w2_norm(e);                             // N (squarings need non-negative limbs)
  const W2 f = w2_sqr(dn);
  const W2 h = w2_mul(dn, f);
  const W2 i = w2_mul(r.x, f);
  const W2 j = w2_norm(w2_sub(w2_add(w2_mul(r.z, w2_sqr(en)), h), w2_add(i, i)));   // z e^2 + h - 2i, N, |V| < 5
  r.z = w2_mul(r.z, h);
  r.x = w2_mul(dn, j);
  r.y = w2_norm(w2_sub(w2_mul(en, w2_sub(i, j)), w2_mul(h, r.y)));       // e (i - j) - h y, N
#endif
}
// Product of two lines (a0 + a2 v^2 + a4 v w)(b0 + b2 v^2 + b4 v w) with v^3 = xi, w^2 = v:
//   1: a0 b0 + xi a4 b4   v: xi a2 b2   v^2: a0 b2 + a2 b0   w: xi (a2 b4 + a4 b2)   v w: a0 b4 + a4 b0   v^2 w: 0
// six Fp2 products (Karatsuba on the three cross terms).  f * l1 * l2 = f * (l1 l2) exactly (field arithmetic).  Used for the
// first iteration of the Miller loop only: merging the two line products of EVERY addition step (6 + 18 products instead of
// 13 + 13) was measured at 135.8 vs 127.0 ms -- the doubling line kept live across the addition step and the dense product's two
// live Fp12 operands doubled the loop's stack frame (720 -> 1328 B per lane) and the spill waits cost more than 2 products save.
// Inputs R / N, outputs R.
BN_DEV W12 w12_line_product(const W2& a0, const W2& a4, const W2& a2, const W2& b0, const W2& b4, const W2& b2) {
  W2 d0, d2, d4, k02, k24, k04;
  W2_MUL2(d0, d2, a0, b0, a2, b2);
  // a_i b_j + a_j b_i = d_i + d_j - (a_i - a_j)(b_i - b_j): lazy differences as operands (coefficients R / N)
  W2_MUL2(d4, k02, a4, b4, w2_sub(a0, a2), w2_sub(b0, b2));
  W2_MUL2(k24, k04, w2_sub(a2, a4), w2_sub(b2, b4), w2_sub(a0, a4), w2_sub(b0, b4));
  W12 r;
  r.c0.c0 = w2_xi_lin(d4, 1, d0, 1);
  r.c0.c1 = w2_xi_lin(d2, 1, d2, 0);
  r.c0.c2 = w2_reduce(w2_sub(w2_add(d0, d2), k02));
  r.c1.c0 = w2_xi_lin(w2_sub(w2_add(d2, d4), k24), 1, d2, 0);
  r.c1.c1 = w2_reduce(w2_sub(w2_add(d0, d4), k04));
  r.c1.c2 = W2{F29{{0, 0, 0, 0, 0, 0, 0, 0, 0}}};
  return r;
}
// whole Miller loop on the carry-free core (points and accumulator).  Same lines, same digit schedule and therefore the same raw
// value as the reference's loop (pairing.rs:590-619); the first iteration, where the accumulator is still one, starts from the
// product of its two lines (w12_line_product) instead of squaring one and multiplying it by each line.
// Loop-invariant operands that are read once or twice per step (P's coordinates for the two line scalings, Q's for the addition
// steps) live in LDS, [limb][thread]: 144 bytes per lane, conflict-free 4-byte reads.  Left in registers they are the values the
// register allocator spills, and every scratch access is waited for in full: a non-kernel function begins with s_waitcnt vmcnt(0), and
// the product leaves are called every ~100 instructions, so each spill store or reload parks the wave for its whole L2 round trip
// (rocprofv3: SQ_WAIT_ANY 11.5 % of the Miller kernel's wave cycles for ~65 scratch instructions per pairing, ~500 cycles each).
// The working G2 point is parked there too while the accumulator is updated (squaring + line product: the phase that needs every
// register); an LDS access is a short lgkmcnt wait.
constexpr int MILLER_LDS_WORDS = 36, MILLER_LDS_WORDS_PARK = 63;
BN_DEV void lds_put9(i32 (*lds)[256], int slot, const F29& a) {
#pragma unroll
  for (int i = 0; i < 9; ++i) lds[9 * slot + i][threadIdx.x] = a.v[i];
}
BN_DEV F29 lds_get9(i32 (*lds)[256], int slot) {
  F29 r;
#pragma unroll
  for (int i = 0; i < 9; ++i) r.v[i] = lds[9 * slot + i][threadIdx.x];
  return r;
}
// PARK: also park the working point (63 words = 64.5 KB per block of 256 threads: two blocks per CU still fit the 160 KB).  Kernels
// that stage line tables in LDS as well (plk_verify.hip) take PARK = false (36 words), or only one block per CU would be resident.
// ISO: the loop runs on (phi P, phi Q) with the twist constant 9 - u (see g2_doubling_step29): same pairing, not the reference's raw Miller value
#ifndef BN_MILLER_SQR_INL
#define BN_MILLER_SQR_INL false
#endif
template <bool PARK, bool ISO = false>
BN_NOINLINE void miller_loop29g(S12& fout, const Fp& pxs, const Fp& pys, const S2& qxs_in, const S2& qys_in) {
  __shared__ i32 lds[PARK ? MILLER_LDS_WORDS_PARK : MILLER_LDS_WORDS][256];   // blocks of 256 threads (BLOCK); each thread touches only its own column
  S2 qxs = qxs_in, qys = qys_in;
  {
    const F29 px = f29_reduce(f29_from_fp(pxs)), py = f29_reduce(f29_from_fp(pys));
    lds_put9(lds, 0, ISO ? f29_mul(px, f29_iso_s2()) : px);
    lds_put9(lds, 1, ISO ? f29_mul(py, f29_iso_s3()) : py);
  }
  {
    W2 qx = w2_from_s2(qxs), qy = w2_from_s2(qys);
    if (ISO) {
      W2_SCALE2(qx, qy, qx, f29_iso_s2(), qy, f29_iso_s3());
      qxs = w2_to_s2(qx); qys = w2_to_s2(qy);                                 // the two Frobenius images at the end start from phi Q
    }
    lds_put9(lds, 2, qx.c);
    lds_put9(lds, 3, qy.c);
  }
  auto PX = [&]() { return lds_get9(lds, 0); };
  auto PY = [&]() { return lds_get9(lds, 1); };
  auto QX = [&]() { return W2{lds_get9(lds, 2)}; };
  auto QY = [&](bool neg) { const W2 y{lds_get9(lds, 3)}; return neg ? w2_neg(y) : y; };      // -Q: a D-class product operand
  auto park = [&](const G2W& r) { if (PARK) { lds_put9(lds, 4, r.x.c); lds_put9(lds, 5, r.y.c); lds_put9(lds, 6, r.z.c); } };
  auto unpark = [&](const G2W& r) { return PARK ? G2W{W2{lds_get9(lds, 4)}, W2{lds_get9(lds, 5)}, W2{lds_get9(lds, 6)}} : r; };
  W12 f;
  G2W r{QX(), QY(false), w2_from_s2(s2_one())};
  W2 l0, l1, l2;
  auto line_mul = [&](const W12& x) {                                    // x * (l0 + l1 P.y (v w) + l2 P.x (v^2)): the two scalings are one leaf pair
    W2 s1, s2;
    W2_SCALE2(s1, s2, l1, PY(), l2, PX());
    return w12_sparse_mul(x, l0, s1, s2);
  };
  const u64 nz = BN_ATE_NAF_NZ, ng = BN_ATE_NAF_NEG;
  static_assert((BN_ATE_NAF_NZ >> 63) & 1, "the first digit of 6x+2 after the leading one is non-zero");
  {   // i = 0: f = 1, so f^2 * l_dbl * l_add is the product of the two lines
    g2_doubling_step29<ISO>(r, l0, l1, l2);
    const W2 d0 = l0;
    W2 d4, d2, e4, e2;
    W2_SCALE2(d4, d2, l1, PY(), l2, PX());
    g2_addition_step29(r, QX(), QY((ng >> 63) & 1), l0, l1, l2);
    park(r);
    W2_SCALE2(e4, e2, l1, PY(), l2, PX());
    f = w12_line_product(d0, d4, d2, l0, e4, e2);
  }
#pragma unroll 1
  for (int i = 1; i < 64; ++i) {
    r = unpark(r);
    g2_doubling_step29<ISO>(r, l0, l1, l2);
    park(r);
    f = w12_sqr<BN_MILLER_SQR_INL>(f);
    f = line_mul(f);
    if ((nz >> (63 - i)) & 1) {
      r = unpark(r);
      g2_addition_step29(r, QX(), QY((ng >> (63 - i)) & 1), l0, l1, l2);
      park(r);
      f = line_mul(f);
    }
  }
  r = unpark(r);
  S2 q1x, q1y, q2x, q2y;
  g2_psi_affine(q1x, q1y, qxs, qys);
  g2_psi_affine(q2x, q2y, q1x, q1y);
  q2y = s2_neg(q2y);
  g2_addition_step29(r, w2_from_s2(q1x), w2_from_s2(q1y), l0, l1, l2);
  f = line_mul(f);
  g2_addition_step29(r, w2_from_s2(q2x), w2_from_s2(q2y), l0, l1, l2);
  f = line_mul(f);
  w12_to_s12(fout, f);
}


// ---- final exponentiation, all of it on the carry-free core (one Fp inversion inside the easy part) -----------------------
// Out-of-line Fp12 routines of the straight-line part.  Operands arrive by reference (an Fp12 is 54 registers per lane, the C ABI
// passes 31): each routine fetches them ONCE, whole, into registers and pins them there -- one wait per routine; reading through the
// references at the points of use costs a full memory wait in front of every product leaf (18 per Fp12 product).
BN_DEV void w12_pin(W12& x) {
  W2* const c[6] = {&x.c0.c0, &x.c0.c1, &x.c0.c2, &x.c1.c0, &x.c1.c1, &x.c1.c2};
#pragma unroll
  for (int j = 0; j < 6; ++j)
#pragma unroll
    for (int i = 0; i < 9; ++i) BN_CHAIN_NV(c[j]->c.v[i]);
}
BN_NOINLINE void w12_mul_nl(W12& r, const W12& a, const W12& b) {
  W12 x = a, y = b;
  w12_pin(x); w12_pin(y);
  r = w12_mul(x, y);
}
BN_NOINLINE void w12_cyclotomic_sqr_nl(W12& r, const W12& a) {
  W12 x = a;
  w12_pin(x);
  r = w12_cyclotomic_sqr(x);
}
template <int E> BN_NOINLINE void w12_frobenius_nl(W12& r, const W12& a) {
  W12 x = a;
  w12_pin(x);
  r = w12_frobenius<E>(x);
}
// pairing.rs:366-392: f^x then conjugate, f in the cyclotomic subgroup (f^-1 = conj f).  x = 4965661367192848881 as a signed-digit chain over
// the table {f^17, f^35}: x = sum d_i 2^i with d_57 = 35 and eleven more digits in {+-17, +-35} (tools/expx_chain_search.py found it;
// tests/test_wnaf_constants.py re-derives x from the three masks) -> 62 cyclotomic squarings + 13 products in all (4 squarings + 1 product
// for f^17, 1 + 1 for f^35, then 57 + 11), against 63 + 16 for the width-4 windows over {f, f^3, f^5, f^7} of rounds 1-3 -- a product costs
// three cyclotomic squarings, so the chain is 9 percent cheaper; the reference's 256-step square-and-multiply reaches the same field element
// with 27 products.   digit i != 0: BN_X_C_NZ;  negative: _NEG;  |d| = 17 (else 35): _17
#ifndef BN_EXP_INL
#define BN_EXP_INL true
#endif
#define BN_X_C_NZ 0x0008144402208421ull
#define BN_X_C_NEG 0x0008004400000020ull
#define BN_X_C_17 0x0000100000008021ull
BN_NOINLINE void exp_by_neg_z29(W12& r, const W12& f) {
  W12 tab[2];                                     // f^17, f^35
  {
    W12 t, u;
    w12_cyclotomic_sqr_nl(t, f);
    w12_cyclotomic_sqr_nl(u, t);
    w12_cyclotomic_sqr_nl(t, u);
    w12_cyclotomic_sqr_nl(u, t);                  // f^16
    w12_mul_nl(tab[0], u, f);
    w12_cyclotomic_sqr_nl(t, tab[0]);             // f^34
    w12_mul_nl(tab[1], t, f);
  }
  W12 res = tab[1];                               // top digit (bit 57) is +35
  const u64 nz = BN_X_C_NZ, ng = BN_X_C_NEG, i17 = BN_X_C_17;
#pragma unroll 1
  for (int i = 56; i >= 0; --i) {
    res = w12_cyclotomic_sqr<BN_EXP_INL>(res);
    if ((nz >> i) & 1) {
      W12 m = tab[((i17 >> i) & 1) ? 0 : 1];
      if ((ng >> i) & 1) m = w12_conj(m);
      res = w12_mul(res, m);
    }
  }
  r = w12_conj(res);
}
// ---- inversion on the carry-free lane-pair core (the easy part's f^-1): fp2.rs:355-360, fp6.rs:415-423, fp12.rs:281-286 -------------
// 1 / (a0 + a1 u) = (a0 - a1 u) / (a0^2 + a1^2): each lane squares its coordinate, the pair's sum is inverted in Fp (safegcd on the
// saturated Montgomery form: f29_to_fp / f29_from_fp only change the Montgomery factor), each lane scales its coordinate, the odd lane
// negates.  Input R / N, output N.  inv(0) = 0 like the reference.
BN_DEV W2 w2_inv(const W2& a) {
  const F29 t = f29_mul_leaf(W_ARGS(a.c), W_ARGS(a.c));
  const F29 n = f29_norm(f29_add(t, xchg9(t)));                       // a0^2 + a1^2 on both lanes, |V| < 2.1
  const F29 i = f29_reduce(f29_from_fp(fp_inv(f29_to_fp(n))));
  const F29 r = f29_mul_leaf(W_ARGS(a.c), W_ARGS(i));
  return W2{sel9(lane_odd(), r, f29_norm(f29_neg(r)))};
}
BN_DEV W2 w2_mul_xi(const W2& a) { return w2_xi_lin(a, 1, a, 0); }     // R
BN_DEV W6 w6_inv(const W6& a) {                                          // input R / N, output N
  W2 s0, m12, s2, m01, s1, m02, u1, u2, o0, o1;
  W2_SQR_MUL(s0, m12, a.c0, a.c1, w2_mul_xi(a.c2));
  W2_SQR_MUL(s2, m01, a.c2, a.c0, a.c1);
  W2_SQR_MUL(s1, m02, a.c1, a.c0, a.c2);
  const W2 t0 = w2_sub(s0, m12);                                         // D-class differences of two N values: fine as product operands
  const W2 t1 = w2_sub(w2_mul_xi(s2), m01);
  const W2 t2 = w2_sub(s1, m02);
  W2_MUL2(u1, u2, a.c2, t1, a.c1, t2);
  const W2 d = w2_xi_lin(w2_add(u1, u2), 1, w2_mul(a.c0, t0), 1);
  const W2 di = w2_inv(d);
  W2_MUL2(o0, o1, di, t0, di, t1);
  return W6{o0, o1, w2_mul(di, t2)};
}
BN_NOINLINE void w12_inv_nl(W12& r, const W12& ain) {
  W12 a = ain;
  w12_pin(a);
  const W6 s0 = w6_mul(a.c0, a.c0), s1 = w6_mul(a.c1, a.c1);           // R
  // c0^2 - v c1^2 with v (x0, x1, x2) = (xi x2, x0, x1); normalised: w6_inv squares its coefficients
  const W6 d{w2_norm(w2_sub(s0.c0, w2_mul_xi(s1.c2))), w2_norm(w2_sub(s0.c1, s1.c0)), w2_norm(w2_sub(s0.c2, s1.c1))};
  const W6 t = w6_inv(d);
  r.c0 = w6_mul(a.c0, t);
  const W6 m = w6_mul(a.c1, t);
  r.c1 = W6{w2_norm(w2_neg(m.c0)), w2_norm(w2_neg(m.c1)), w2_norm(w2_neg(m.c2))};
}
BN_NOINLINE void final_exponentiation29(S12& out, const S12& fin) {
  W12 in, t, a, b, d, e, g;
  {   // easy part (pairing.rs:410-430): f^(p^6 - 1) = conj(f) f^-1, then ^(p^2 + 1) = frobenius^2(.) * (.)
    w12_from_s12(t, fin);
    w12_inv_nl(b, t);
    a = w12_conj(t);
    w12_mul_nl(d, a, b);
    w12_frobenius_nl<2>(a, d);
    w12_mul_nl(in, a, d);
  }
  exp_by_neg_z29(a, in);
  w12_cyclotomic_sqr_nl(b, a);
  w12_cyclotomic_sqr_nl(t, b);
  w12_mul_nl(d, t, b);
  exp_by_neg_z29(e, d);
  w12_cyclotomic_sqr_nl(t, e);
  exp_by_neg_z29(g, t);
  d = w12_conj(d);
  g = w12_conj(g);
  w12_mul_nl(t, g, e);
  w12_mul_nl(a, t, d);
  w12_mul_nl(d, a, b);
  w12_mul_nl(t, a, e);
  w12_mul_nl(e, in, t);
  w12_frobenius_nl<1>(t, d);
  w12_mul_nl(b, t, e);
  w12_frobenius_nl<2>(t, a);
  w12_mul_nl(e, t, b);
  t = w12_conj(in);
  w12_mul_nl(a, t, d);
  w12_frobenius_nl<3>(t, a);
  w12_mul_nl(g, t, e);
  w12_to_s12(out, g);
}


// The product of the one-wavefront-per-element routines below is INLINED: they run one wavefront per SIMD by nature, and a lone wavefront pays
// for every instruction -- the 27 argument / result moves and the call of the out-of-line leaf included (wide signing: 1.05 -> 1.01 ms).
// BN_WIDE_LEAF_CALL restores the call (A/B runs).
#ifdef BN_WIDE_LEAF_CALL
BN_DEV W2 w2_mul_w(const W2& a, const W2& b) { return w2_mul(a, b); }
#else
BN_DEV W2 w2_mul_w(const W2& a, const W2& b) { return w2_mul_inl(a, b); }
#endif
// ---- ONE final exponentiation on a whole wavefront ("wide"): the tail of every one-boolean shape -----------------------------------
// A single element on one lane pair is pure latency: the wavefront issues a multiply-add every ~8 cycles and 62 of its 64 lanes idle
// (2.4-2.9 ms per final exponentiation).  Here ALL 32 lane pairs of a one-wavefront block hold the SAME element (replicated: same code,
// same inputs), and the cyclotomic squarings of the three f^x chains -- 189 of them, six independent Fp2 products each -- are spread:
// lane pair j < 6 forms product j, the products meet in LDS, lane pair j forms output coefficient j, the coefficients meet in LDS and
// every lane pair holds the square again.  Same formulas, same operand classes and therefore the same digits as w12_cyclotomic_sqr.
// [slot][lane parity][limb, padded to 12]: a value is 48 contiguous bytes at a 16-byte boundary, so that a put / get is two 16-byte LDS
// accesses and one of 4 bytes instead of nine of 4 (a third of the LDS instructions of a wide product).  Squaring: 6 products + 6 outputs; product: slots 12 .. 62
struct alignas(16) WideLds { i32 v[63][2][12]; };
typedef __attribute__((address_space(3))) WideLds* WideLdsPtr;
// EPW = elements per wavefront.  1: all 32 lane pairs hold the same element.  2: lanes 0-31 and 32-63 hold one element each (16 lane pairs,
// each half with its own WideLds): batches of 1025 .. 4096 elements then still run one or two wavefronts per SIMD.  The only level with more
// than 16 products (the 18 of the dense Fp12 product) takes a second pass on two lane pairs.
template <int EPW> BN_DEV int wide_j(int lane) { return (int)pair_index((u32)lane) & (32 / EPW - 1); }
typedef i32 __attribute__((ext_vector_type(4))) WideVec4;
typedef __attribute__((address_space(3))) WideVec4* WideVec4Ptr;
BN_DEV void wide_put(WideLdsPtr x, int slot, int odd, const W2& a) {
  const WideVec4Ptr p = (WideVec4Ptr)&x->v[slot][odd][0];
  p[0] = WideVec4{a.c.v[0], a.c.v[1], a.c.v[2], a.c.v[3]};
  p[1] = WideVec4{a.c.v[4], a.c.v[5], a.c.v[6], a.c.v[7]};
  x->v[slot][odd][8] = a.c.v[8];
}
BN_DEV W2 wide_get(WideLdsPtr x, int slot, int odd) {
  const WideVec4Ptr p = (WideVec4Ptr)&x->v[slot][odd][0];
  const WideVec4 lo = p[0], hi = p[1];
  W2 r;
  r.c.v[0] = lo.x; r.c.v[1] = lo.y; r.c.v[2] = lo.z; r.c.v[3] = lo.w;
  r.c.v[4] = hi.x; r.c.v[5] = hi.y; r.c.v[6] = hi.z; r.c.v[7] = hi.w;
  r.c.v[8] = x->v[slot][odd][8];
  return r;
}
BN_DEV W2 w2_pick(const W2& a, const W2& b, bool c) { return W2{sel9(c, a.c, b.c)}; }                // c ? b : a
BN_DEV W2 w2_real(const F29& k) { return W2{sel9(lane_odd(), k, F29{{0, 0, 0, 0, 0, 0, 0, 0, 0}})}; }   // the Fp element k as (k, 0)
BN_DEV W2 w2_sel3(int k, const W2& a, const W2& b, const W2& c) { return w2_pick(w2_pick(a, b, k == 1), c, k == 2); }
template <int EPW = 1>
BN_DEV W12 w12_cyclotomic_sqr_wide(const W12& f, WideLdsPtr x) {
  const int lane = (int)(threadIdx.x & 63u), odd = pair_role((u32)lane), j = wide_j<EPW>(lane);
  const int p = j < 6 ? j : 0, k = p >> 1;
  const bool s = (p & 1) != 0;
  const W2 z0 = f.c0.c0, z4 = f.c0.c1, z3 = f.c0.c2, z2 = f.c1.c0, z1 = f.c1.c1, z5 = f.c1.c2;
  // product p: pair k of (z0, z1), (z2, z3), (z4, z5);  s = 0: m = a b,  s = 1: w = (a + b)(a + xi b)   (w_fp4_square)
  const W2 a = w2_sel3(k, z0, z2, z4), b = w2_sel3(k, z1, z3, z5);
  {
    const W2 xs = w2_pick(a, w2_norm(w2_add(a, b)), s), ys = w2_pick(b, w2_xi_norm(b, a), s);
    const W2 pr = w2_mul_w(xs, ys);
    if (j < 6) wide_put(x, p, odd, pr);
  }
  __syncthreads();
  // Outputs, with c0(k) = w(k) - m(k) - xi m(k).  Lane pair p forms the output whose z it already selected as a product operand (a for even
  // p, b for odd p), from the products of pair ko:
  //   p = 0: c0.c0 = 3 c0(0) - 2 z0   1: c1.c1 = 6 m(0) + 2 z1   2: c1.c0 = 6 xi m(2) + 2 z2   3: c0.c2 = 3 c0(2) - 2 z3
  //       4: c0.c1 = 3 c0(1) - 2 z4   5: c1.c2 = 6 m(1) + 2 z5
  // All three shapes are ONE pass c_w w + c_m m + c_mo (partner's m) + c_z z with lane-dependent coefficients (xi m on this lane is
  // 9 m -/+ partner's m): 3 c0 - 2 z = 3 w - 30 m -/+ 3 mo - 2 z;  6 m + 2 z;  6 xi m + 2 z = 54 m -/+ 6 mo + 2 z.
  {
    const int ko = p < 2 ? 0 : p < 4 ? 2 : 1;
    const bool ta = p == 0 || p == 3 || p == 4, tc = p == 2;
    const W2 m = wide_get(x, 2 * ko, odd), w = wide_get(x, 2 * ko + 1, odd);
    const W2 z = w2_pick(a, b, s);
    const F29 mo = xchg9(m.c);
    const bool lo = lane_odd();
    const F29* const t[4] = {&w.c, &m.c, &mo, &z.c};
    const i32 c[4] = {bn_keep_v(ta ? 3 : 0), bn_keep_v(ta ? -30 : tc ? 54 : 6), bn_keep_v(ta ? (lo ? -3 : 3) : tc ? (lo ? 6 : -6) : 0), bn_keep_v(ta ? -2 : 2)};
    const W2 r{f29_reduce_terms(t, c)};
    if (j < 6) wide_put(x, 6 + (p < 2 ? p : p < 4 ? p + 2 : p - 2), odd, r);
  }
  __syncthreads();
  W12 r;
  r.c0.c0 = wide_get(x, 6, odd); r.c1.c1 = wide_get(x, 7, odd); r.c0.c1 = wide_get(x, 8, odd);
  r.c1.c2 = wide_get(x, 9, odd); r.c1.c0 = wide_get(x, 10, odd); r.c0.c2 = wide_get(x, 11, odd);
  return r;
}
// a * b with the 18 products of the Karatsuba-over-Karatsuba form (w12_mul / w6_mul) on 18 lane pairs.  Inputs and result replicated.
// Slots: the operands' coefficients IA / IB (lane pair c < 6 writes coefficient c of each), the coefficient sums of the third Fp6 product
// SA / SB, the products P[6 g + h] (g: which Fp6 product, h: v0 v1 v2 and the three cross products), the nine Fp6 coefficients T[3 g + c],
// the six outputs.  Every value goes through exactly the operations of w12_mul, so the digits are the same.
// KIND (what a wavefront half of 16 lane pairs needs to stay at ONE product level):
//   WK_DENSE   any a, b: 18 products (EPW = 2: 16 + 2 in two passes)
//   WK_LINE    b = (l0, 0, l2; 0, l4, 0), a line of the Miller loop: four of the 18 products have a zero factor -- 14 products, zeros stored
//   WK_SQUARE  a^2 (b ignored) as the complex squaring of fp12.rs:536-550, c1 = 2 t, c0 = (a0 + a1)(a0 + v a1) - t - v t with t = a0 a1:
//              two Fp6 products, 12 products; t takes the place of the first Fp6 product, the other one that of the third
constexpr int WL_IA = 12, WL_IB = 18, WL_SA = 24, WL_SB = 27, WL_P = 30, WL_T = 48, WL_OUT = 57;
constexpr int WK_DENSE = 0, WK_LINE = 1, WK_SQUARE = 2;
#ifndef BN_WIDE_MILLER_INL
#define BN_WIDE_MILLER_INL 2  // the wide Miller loop's Fp12 products inlined: 0 none, 1 the squaring, 2 the squaring and the line products (A/B runs)
#endif
#ifndef BN_WIDE_KINDS
#define BN_WIDE_KINDS 1      // 0: the Miller loop of the wide routines with dense products only (A/B runs)
#endif
BN_DEV W2 w12_coef(const W12& a, int c) {      // c = 3 * half + i
  const W2 lo = w2_sel3(c % 3, a.c0.c0, a.c0.c1, a.c0.c2), hi = w2_sel3(c % 3, a.c1.c0, a.c1.c1, a.c1.c2);
  return w2_pick(lo, hi, c >= 3);
}
// The product in three pieces: stage 0 (operands from registers into LDS), the core (LDS to LDS: callable out of line with nothing but the
// LDS pointer -- an out-of-line call that takes Fp12 values by reference sends them through the stack frame, ~2 us for a lone wavefront)
// and the read of the six output coefficients.
template <int EPW = 1, int KIND = WK_DENSE>
BN_DEV void w12_mul_wide_stage0(const W12& a, const W12& b, WideLdsPtr x) {
  const int lane = (int)(threadIdx.x & 63u), odd = pair_role((u32)lane), j = wide_j<EPW>(lane);
  {   // stage 0: coefficients and the sums a.c0 + a.c1, b.c0 + b.c1 into LDS
    const int c = j < 6 ? j : 0;
    const int i = c % 3;
    const W2 a0i = w2_sel3(i, a.c0.c0, a.c0.c1, a.c0.c2), a1i = w2_sel3(i, a.c1.c0, a.c1.c1, a.c1.c2);
    const W2 sa = w2_norm(w2_add(a0i, a1i));
    if (KIND == WK_SQUARE) {
      // a0 + v a1 = (a0.c0 + xi a1.c2, a0.c1 + a1.c0, a0.c2 + a1.c1)
      const W2 sb0 = w2_xi_lin(a.c1.c2, 1, a.c0.c0, 1);
      const W2 sb12 = w2_norm(w2_add(a0i, w2_pick(a.c1.c0, a.c1.c1, i == 2)));
      if (j < 3) { wide_put(x, WL_IA + i, odd, a0i); wide_put(x, WL_IB + i, odd, a1i); wide_put(x, WL_SA + i, odd, sa); }
      if (j >= 3 && j < 6) wide_put(x, WL_SB + i, odd, w2_pick(sb12, sb0, i == 0));
    } else {
      const W2 ac = w2_pick(a0i, a1i, c >= 3), bc = w12_coef(b, c);
      const W2 sb = w2_norm(w2_add(w2_sel3(i, b.c0.c0, b.c0.c1, b.c0.c2), w2_sel3(i, b.c1.c0, b.c1.c1, b.c1.c2)));
      if (j < 6) { wide_put(x, WL_IA + c, odd, ac); wide_put(x, WL_IB + c, odd, bc); }
      if (j < 3) wide_put(x, WL_SA + i, odd, sa);
      if (j >= 3 && j < 6) wide_put(x, WL_SB + i, odd, sb);
    }
  }
}
template <int EPW = 1, int KIND = WK_DENSE, bool STAGE3 = true>      // STAGE3 = false: stop at the Fp6 coefficients in the T slots (w12_inv_wide)
BN_DEV void w12_mul_wide_core(WideLdsPtr x) {
  const int lane = (int)(threadIdx.x & 63u), odd = pair_role((u32)lane), j = wide_j<EPW>(lane);
  const W2 zero{F29{{0, 0, 0, 0, 0, 0, 0, 0, 0}}};
  __syncthreads();
#pragma unroll
  for (int pass = 0; pass < (KIND == WK_DENSE ? EPW : 1); ++pass) {   // stage 1: product w = 6 g + h
    int w;
    bool live;
    if (KIND == WK_DENSE) {
      const int jw = j + 16 * pass;                                   // EPW = 2: products 16, 17 in a second pass
      live = jw < 18;
      w = live ? jw : 0;
    } else if (KIND == WK_LINE) {                                     // products 1, 6, 8, 11 have a zero factor (h = 3, 4, 5: pairs 12, 01, 02)
      live = j < 14;
      w = live ? j + (j >= 1) + (j >= 5) + (j >= 6) + (j >= 8) : 0;   // 0 2 3 4 5 7 9 10 12 .. 17
    } else {                                                          // t = a0 a1 (g = 0), (a0 + a1)(a0 + v a1) (g = 2)
      live = j < 12;
      w = j < 6 ? j : live ? j + 6 : 0;
    }
    const int g = w / 6, h = w % 6;
    const int i0 = h < 3 ? h : h == 3 ? 1 : 0, i1 = h < 3 ? h : h == 4 ? 1 : 2;
    const int ba = g == 0 ? WL_IA : g == 1 ? WL_IA + 3 : WL_SA, bb = g == 0 ? WL_IB : g == 1 ? WL_IB + 3 : WL_SB;
    const W2 xa = wide_get(x, ba + i0, odd), xb = wide_get(x, ba + i1, odd);
    const W2 ya = wide_get(x, bb + i0, odd), yb = wide_get(x, bb + i1, odd);
    const W2 xs = w2_pick(xa, w2_norm(w2_add(xa, xb)), h >= 3), ys = w2_pick(ya, w2_norm(w2_add(ya, yb)), h >= 3);
    const W2 pr = w2_mul_w(xs, ys);
    if (live) wide_put(x, WL_P + w, odd, pr);
    if (KIND == WK_LINE && j < 4) wide_put(x, WL_P + (int)((0x0B080601u >> (8 * j)) & 255u), odd, zero);      // slots 1, 6, 8, 11
  }
  __syncthreads();
  {   // stage 2: Fp6 coefficient u = 3 g + c of the three Fp6 products (w6_mul); WK_SQUARE: of the first and the third
    const int nu = KIND == WK_SQUARE ? 6 : 9;
    const int u = j >= nu ? 0 : (KIND == WK_SQUARE && j >= 3) ? j + 3 : j, g = u / 3, c = u % 3, base = WL_P + 6 * g;
    const W2 v0 = wide_get(x, base, odd), v1 = wide_get(x, base + 1, odd), v2 = wide_get(x, base + 2, odd), q = wide_get(x, base + 3 + c, odd);
    // c = 0: v0 + xi (q - v1 - v2)   1: (q - v0 - v1) + xi v2   2: q - v0 - v2 + v1 -- ONE pass with lane-dependent coefficients over
    // (X, partner's X, q, v0, v1, v2), X the value xi applies to (xi X on this lane is 9 X -/+ partner's X)
    const W2 xv = w2_pick(v2, w2_sub(w2_sub(q, v1), v2), c == 0);
    const F29 xo = xchg9(xv.c);
    const bool lo = lane_odd();
    const F29* const t[6] = {&xv.c, &xo, &q.c, &v0.c, &v1.c, &v2.c};
    const i32 kx = c == 2 ? 0 : 1;
    const i32 k[6] = {bn_keep_v(9 * kx), bn_keep_v(lo ? kx : -kx), bn_keep_v(c == 0 ? 0 : 1), bn_keep_v(c == 0 ? 1 : -1),
                      bn_keep_v(c == 0 ? 0 : c == 1 ? -1 : 1), bn_keep_v(c == 2 ? -1 : 0)};
    if (j < nu) wide_put(x, WL_T + u, odd, W2{f29_reduce_terms(t, k)});
  }
  __syncthreads();
  if (!STAGE3) return;
  if (KIND == WK_SQUARE) {   // stage 3: c1.ci = 2 t.ci;  c0.c0 = m.c0 - t.c0 - xi t.c2;  c0.c1 = m.c1 - t.c1 - t.c0;  c0.c2 = m.c2 - t.c2 - t.c1
    const int o = j < 6 ? j : 0, i = o % 3;
    const W2 ti = wide_get(x, WL_T + i, odd), tp = wide_get(x, WL_T + (i + 2) % 3, odd), mi = wide_get(x, WL_T + 6 + i, odd);
    // o >= 3: 2 t_i   o = 0: m_i - t_i - xi t_p   else: m_i - t_i - t_p -- one pass over (t_i, t_p, partner's t_p, m_i)
    const F29 tpo = xchg9(tp.c);
    const bool lo = lane_odd(), hi = o >= 3, o0 = o == 0;
    const F29* const t[4] = {&ti.c, &tp.c, &tpo, &mi.c};
    const i32 k[4] = {bn_keep_v(hi ? 2 : -1), bn_keep_v(hi ? 0 : o0 ? -9 : -1), bn_keep_v(o0 ? (lo ? -1 : 1) : 0), bn_keep_v(hi ? 0 : 1)};
    if (j < 6) wide_put(x, WL_OUT + o, odd, W2{f29_reduce_terms(t, k)});
  } else {   // stage 3: output o (w12_mul): c1.ci = T2.ci - T0.ci - T1.ci;  c0.c0 = T0.c0 + xi T1.c2;  c0.c1 = T0.c1 + T1.c0;  c0.c2 = T0.c2 + T1.c1
    const int o = j < 6 ? j : 0, i = o % 3;
    const int pa = o >= 3 ? i : o, pb = o >= 3 ? 3 + i : o == 0 ? 5 : o == 1 ? 3 : 4, pc = o >= 3 ? 6 + i : 0;
    const W2 ta = wide_get(x, WL_T + pa, odd), tb = wide_get(x, WL_T + pb, odd), tc = wide_get(x, WL_T + pc, odd);
    // o >= 3: tc - ta - tb   o = 0: ta + xi tb   else: ta + tb -- one pass over (ta, tb, partner's tb, tc)
    const F29 tbo = xchg9(tb.c);
    const bool lo = lane_odd(), hi = o >= 3, o0 = o == 0;
    const F29* const t[4] = {&ta.c, &tb.c, &tbo, &tc.c};
    const i32 k[4] = {bn_keep_v(hi ? -1 : 1), bn_keep_v(hi ? -1 : o0 ? 9 : 1), bn_keep_v(o0 ? (lo ? 1 : -1) : 0), bn_keep_v(hi ? 1 : 0)};
    if (j < 6) wide_put(x, WL_OUT + o, odd, W2{f29_reduce_terms(t, k)});
  }
  __syncthreads();
}
BN_DEV W12 w12_wide_result(WideLdsPtr x) {
  const int odd = pair_role((u32)(threadIdx.x & 63u));
  W12 r;
  r.c0.c0 = wide_get(x, WL_OUT, odd); r.c0.c1 = wide_get(x, WL_OUT + 1, odd); r.c0.c2 = wide_get(x, WL_OUT + 2, odd);
  r.c1.c0 = wide_get(x, WL_OUT + 3, odd); r.c1.c1 = wide_get(x, WL_OUT + 4, odd); r.c1.c2 = wide_get(x, WL_OUT + 5, odd);
  return r;
}
template <int EPW = 1, int KIND = WK_DENSE>
BN_DEV W12 w12_mul_wide(const W12& a, const W12& b, WideLdsPtr x) {
  w12_mul_wide_stage0<EPW, KIND>(a, b, x);
  w12_mul_wide_core<EPW, KIND>(x);
  return w12_wide_result(x);
}
template <int EPW = 1, int KIND = WK_DENSE>
BN_NOINLINE void w12_mul_wide_core_nl(WideLds* xg) { w12_mul_wide_core<EPW, KIND>((WideLdsPtr)xg); }
template <int EPW = 1, int KIND = WK_DENSE>
BN_DEV void w12_mul_wide_nl(W12& r, const W12& a, const W12& b, WideLds* xg) {
  w12_mul_wide_stage0<EPW, KIND>(a, b, (WideLdsPtr)xg);
  w12_mul_wide_core_nl<EPW, KIND>(xg);
  r = w12_wide_result((WideLdsPtr)xg);
}
// exp_by_neg_z29 with the loop's squarings and products spread over the wavefront
template <int EPW = 1>
BN_NOINLINE void exp_by_neg_z29_wide(W12& r, const W12& f, WideLds* xg) {
  const WideLdsPtr x = (WideLdsPtr)xg;
  W12 tab[2];                                     // f^17, f^35: the chain of exp_by_neg_z29
  {
    // f^16 by four squarings, f^17 = f^16 f, f^34, f^35 = f^34 f -- the squarings spread over the wavefront in ONE loop (no out-of-line
    // call with values by reference: each costs a lone wavefront ~2 us through the stack frame)
    W12 t = f;
#pragma unroll 1
    for (int k = 0; k < 5; ++k) {
      t = w12_cyclotomic_sqr_wide<EPW>(t, x);
      if (k == 3) {
        w12_mul_wide_nl<EPW>(tab[0], t, f, xg);
        t = tab[0];
      }
    }
    w12_mul_wide_nl<EPW>(tab[1], t, f, xg);
  }
  W12 res = tab[1];
  const u64 nz = BN_X_C_NZ, ng = BN_X_C_NEG, i17 = BN_X_C_17;
#pragma unroll 1
  for (int i = 56; i >= 0; --i) {
    res = w12_cyclotomic_sqr_wide<EPW>(res, x);
    if ((nz >> i) & 1) {
      W12 m = tab[((i17 >> i) & 1) ? 0 : 1];
      if ((ng >> i) & 1) m = w12_conj(m);
      res = w12_mul_wide<EPW>(res, m, x);
    }
  }
  r = w12_conj(res);
}
// w12_frobenius<E> on the wavefront: the five products by the Frobenius constants on five lane pairs (the conjugated coefficients c0.c1,
// c0.c2, c1.c0, c1.c1, c1.c2 by k1, k2, kk, k1, k2), then the second factor kk of c1.c1 and c1.c2 on two -- two product levels instead
// of seven products on one lane pair behind an out-of-line call by reference.  Same products in the same order: same values.
template <int E, int EPW = 1>
BN_DEV W12 w12_frobenius_wide(const W12& a, WideLdsPtr x) {
  constexpr bool oddE = (E & 1) != 0;
  const int lane = (int)(threadIdx.x & 63u), odd = pair_role((u32)lane), j = wide_j<EPW>(lane);
  const W2 k1 = w2_const((E == 1) ? C_FROB6_C1_1 : (E == 2) ? C_FROB6_C1_2 : C_FROB6_C1_3);
  const W2 k2 = w2_const((E == 1) ? C_FROB6_C2_1 : (E == 2) ? C_FROB6_C2_2 : C_FROB6_C2_3);
  const W2 kk = w2_const((E == 1) ? C_FROB12_C1_1 : (E == 2) ? C_FROB12_C1_2 : C_FROB12_C1_3);
  {
    const int p = j < 5 ? j : 0;
    const W2 c = w2_pick(w2_pick(w2_pick(w2_pick(a.c0.c1, a.c0.c2, p == 1), a.c1.c0, p == 2), a.c1.c1, p == 3), a.c1.c2, p == 4);
    const W2 u = oddE ? w2_conj(c) : c;
    const W2 v = w2_pick(w2_pick(k1, k2, p == 1 || p == 4), kk, p == 2);
    const W2 pr = w2_mul_w(u, v);
    if (j < 5) wide_put(x, WL_P + p, odd, pr);
  }
  __syncthreads();
  {
    const W2 u = wide_get(x, WL_P + 3 + (j == 1 ? 1 : 0), odd);
    const W2 pr = w2_mul_w(u, kk);
    if (j < 2) wide_put(x, WL_T + j, odd, pr);
  }
  __syncthreads();
  W12 r;
  r.c0.c0 = oddE ? w2_conj(a.c0.c0) : a.c0.c0;
  r.c0.c1 = wide_get(x, WL_P, odd); r.c0.c2 = wide_get(x, WL_P + 1, odd);
  r.c1.c0 = wide_get(x, WL_P + 2, odd); r.c1.c1 = wide_get(x, WL_T, odd); r.c1.c2 = wide_get(x, WL_T + 1, odd);
  return r;
}
// Two independent Fp6 products A B and C D in ONE pass of the product machinery (the first and the third Fp6 product of the square form:
// 12 lane pairs): the coefficients of A B land in T[0 .. 2], those of C D in T[6 .. 8].  Operands replicated, R / N.
template <int EPW>
BN_DEV void w6_mul2_wide(const W6& a, const W6& b, const W6& c, const W6& d, WideLdsPtr x) {
  const int lane = (int)(threadIdx.x & 63u), odd = pair_role((u32)lane), j = wide_j<EPW>(lane);
  const int i = (j < 6 ? j : 0) % 3;
  const bool hi = j >= 3;
  const W2 u = w2_sel3(i, w2_pick(a.c0, c.c0, hi), w2_pick(a.c1, c.c1, hi), w2_pick(a.c2, c.c2, hi));
  const W2 v = w2_sel3(i, w2_pick(b.c0, d.c0, hi), w2_pick(b.c1, d.c1, hi), w2_pick(b.c2, d.c2, hi));
  if (j < 3) { wide_put(x, WL_IA + i, odd, u); wide_put(x, WL_IB + i, odd, v); }
  if (j >= 3 && j < 6) { wide_put(x, WL_SA + i, odd, u); wide_put(x, WL_SB + i, odd, v); }
  w12_mul_wide_core<EPW, WK_SQUARE, false>(x);
}
BN_DEV W6 w6_wide_t(WideLdsPtr x, int base) {
  const int odd = pair_role((u32)(threadIdx.x & 63u));
  return W6{wide_get(x, WL_T + base, odd), wide_get(x, WL_T + base + 1, odd), wide_get(x, WL_T + base + 2, odd)};
}
// w12_inv (fp12.rs:281-286, fp6.rs:415-423) on the wavefront: the two Fp6 squarings and the two final Fp6 products as double products,
// the nine products of the Fp6 inversion in three levels of 6 / 3 / 3 lane pairs, the Fp2 inversion replicated.  Same formulas as
// w12_inv_nl / w6_inv: same values.
template <int EPW = 1>
BN_DEV W12 w12_inv_wide(const W12& a, WideLdsPtr x) {
  const int lane = (int)(threadIdx.x & 63u), odd = pair_role((u32)lane), j = wide_j<EPW>(lane);
  w6_mul2_wide<EPW>(a.c0, a.c0, a.c1, a.c1, x);
  const W6 s0 = w6_wide_t(x, 0), s1 = w6_wide_t(x, 6);
  // d = c0^2 - v c1^2 with v (x0, x1, x2) = (xi x2, x0, x1)
  const W2 d0 = w2_norm(w2_sub(s0.c0, w2_mul_xi(s1.c2))), d1 = w2_norm(w2_sub(s0.c1, s1.c0)), d2 = w2_norm(w2_sub(s0.c2, s1.c1));
  {   // six products: d0 d0, d1 (xi d2), d2 d2, d0 d1, d1 d1, d0 d2
    const int p = j < 6 ? j : 0;
    const W2 xd2 = w2_mul_xi(d2);
    const W2 u = w2_pick(w2_pick(d0, d1, p == 1 || p == 4), d2, p == 2);
    const W2 v = w2_pick(w2_pick(w2_pick(w2_pick(d0, xd2, p == 1), d2, p == 2 || p == 5), d1, p == 3 || p == 4), d0, p == 0);
    const W2 pr = w2_mul_w(u, v);
    if (j < 6) wide_put(x, WL_P + p, odd, pr);
  }
  __syncthreads();
  const W2 t0 = w2_sub(wide_get(x, WL_P, odd), wide_get(x, WL_P + 1, odd));                       // D-class differences: fine as product operands
  const W2 t1 = w2_sub(w2_mul_xi(wide_get(x, WL_P + 2, odd)), wide_get(x, WL_P + 3, odd));
  const W2 t2 = w2_sub(wide_get(x, WL_P + 4, odd), wide_get(x, WL_P + 5, odd));
  {   // three products: d2 t1, d1 t2, d0 t0
    const int p = j < 3 ? j : 0;
    const W2 pr = w2_mul_w(w2_sel3(p, d2, d1, d0), w2_sel3(p, t1, t2, t0));
    if (j < 3) wide_put(x, WL_P + 6 + p, odd, pr);
  }
  __syncthreads();
  const W2 dd = w2_xi_lin(w2_add(wide_get(x, WL_P + 6, odd), wide_get(x, WL_P + 7, odd)), 1, wide_get(x, WL_P + 8, odd), 1);
  const W2 di = w2_inv(dd);
  {   // three products: di t0, di t1, di t2
    const int p = j < 3 ? j : 0;
    const W2 pr = w2_mul_w(di, w2_sel3(p, t0, t1, t2));
    if (j < 3) wide_put(x, WL_P + 9 + p, odd, pr);
  }
  __syncthreads();
  const W6 t{wide_get(x, WL_P + 9, odd), wide_get(x, WL_P + 10, odd), wide_get(x, WL_P + 11, odd)};
  w6_mul2_wide<EPW>(a.c0, t, a.c1, t, x);
  const W6 m = w6_wide_t(x, 6);
  W12 r;
  r.c0 = w6_wide_t(x, 0);
  r.c1 = W6{w2_norm(w2_neg(m.c0)), w2_norm(w2_neg(m.c1)), w2_norm(w2_neg(m.c2))};
  return r;
}
// final_exponentiation29 for a one-wavefront block whose 32 lane pairs all hold the same element
template <int EPW = 1>
BN_NOINLINE void final_exponentiation29_wide(S12& out, const S12& fin, WideLds* x) {
  W12 in, t, a, b, d, e, g;
  {
    w12_from_s12(t, fin);
    b = w12_inv_wide<EPW>(t, (WideLdsPtr)x);
    a = w12_conj(t);
    w12_mul_wide_nl<EPW>(d, a, b, x);
    a = w12_frobenius_wide<2, EPW>(d, (WideLdsPtr)x);
    w12_mul_wide_nl<EPW>(in, a, d, x);
  }
  exp_by_neg_z29_wide<EPW>(a, in, x);
  b = w12_cyclotomic_sqr_wide<EPW>(a, (WideLdsPtr)x);             // the wide squaring inline: no lane-pair routine by reference on a lone wavefront
  t = w12_cyclotomic_sqr_wide<EPW>(b, (WideLdsPtr)x);
  w12_mul_wide_nl<EPW>(d, t, b, x);
  exp_by_neg_z29_wide<EPW>(e, d, x);
  t = w12_cyclotomic_sqr_wide<EPW>(e, (WideLdsPtr)x);
  exp_by_neg_z29_wide<EPW>(g, t, x);
  d = w12_conj(d);
  g = w12_conj(g);
  w12_mul_wide_nl<EPW>(t, g, e, x);
  w12_mul_wide_nl<EPW>(a, t, d, x);
  w12_mul_wide_nl<EPW>(d, a, b, x);
  w12_mul_wide_nl<EPW>(t, a, e, x);
  w12_mul_wide_nl<EPW>(e, in, t, x);
  t = w12_frobenius_wide<1, EPW>(d, (WideLdsPtr)x);
  w12_mul_wide_nl<EPW>(b, t, e, x);
  t = w12_frobenius_wide<2, EPW>(a, (WideLdsPtr)x);
  w12_mul_wide_nl<EPW>(e, t, b, x);
  t = w12_conj(in);
  w12_mul_wide_nl<EPW>(a, t, d, x);
  t = w12_frobenius_wide<3, EPW>(a, (WideLdsPtr)x);
  w12_mul_wide_nl<EPW>(g, t, e, x);
  w12_to_s12(out, g);
}

// g2_doubling_step29 with its ten products in three levels: five on five lane pairs (x y, x^2, y^2, z^2, (y + z)^2), the twist-constant
// product replicated (one product: nothing to spread), four on four lane pairs (b h, a (b - f), g^2, e^2).  Inputs and outputs replicated;
// the squares are taken with the product leaf (same values).  Products of level 1 meet in the P slots, those of level 3 in the T slots.
template <bool ISO = false, int EPW = 1>
BN_DEV void g2_doubling_step29_wide(G2W& r, W2& l0, W2& l1, W2& l2, const F29& px, const F29& py, WideLdsPtr x) {
  const int lane = (int)(threadIdx.x & 63u), odd = pair_role((u32)lane), j = wide_j<EPW>(lane);
  {
    const W2 yz = w2_norm(w2_add(r.y, r.z));
    const int p = j < 5 ? j : 0;
    const W2 a = w2_pick(w2_pick(w2_pick(r.x, r.y, p == 2), r.z, p == 3), yz, p == 4);        // 0, 1: x   2: y   3: z   4: y + z
    const W2 b = w2_pick(a, r.y, p == 0);                                                       // 0: y, else the same operand (a square)
    const W2 pr = w2_mul_w(a, b);
    if (j < 5) wide_put(x, WL_P + p, odd, pr);
  }
  __syncthreads();
  const W2 xy = wide_get(x, WL_P, odd), xx = wide_get(x, WL_P + 1, odd), b = wide_get(x, WL_P + 2, odd), c = wide_get(x, WL_P + 3, odd),
           s = wide_get(x, WL_P + 4, odd);
  const W2 a = w2_halve(xy);
  l2 = w2_norm(w2_triple(xx));
  const W2 h = w2_norm(w2_sub(s, w2_add(b, c)));
  const W2 e = ISO ? w2_mul_27m3u(c) : w2_mul_w(w2_twist_b(), w2_norm(w2_triple(c)));      // E'': (9 - u) * 3c in one reduce pass
  l0 = w2_xi_lin(w2_sub(e, b), 1, b, 0);
  const W2 f = w2_norm(w2_triple(e));
  const W2 g = w2_halve(w2_norm(w2_add(b, f)));
  {   // two idle lane pairs scale the line: 4: l2 x_P, 5: l1 y_P (the real factor as the Fp2 element (k, 0))
    const int p = j < 6 ? j : 0;
    const W2 u = w2_pick(w2_pick(w2_pick(w2_pick(w2_pick(b, a, p == 1), g, p == 2), e, p == 3), l2, p == 4), w2_neg(h), p == 5);   // 0: b  1: a  2: g  3: e
    const W2 v = w2_pick(w2_pick(w2_pick(w2_pick(h, w2_sub(b, f), p == 1), g, p == 2), e, p == 3), w2_real(sel9(p == 5, px, py)), p >= 4);   // 0: h  1: b - f  2: g  3: e
    const W2 pr = w2_mul_w(u, v);
    if (j < 6) wide_put(x, WL_T + p, odd, pr);
  }
  __syncthreads();
  l2 = wide_get(x, WL_T + 4, odd);
  l1 = wide_get(x, WL_T + 5, odd);
  r.z = wide_get(x, WL_T, odd);
  r.x = wide_get(x, WL_T + 1, odd);
  r.y = w2_lin2(wide_get(x, WL_T + 2, odd), 1, wide_get(x, WL_T + 3, odd), -3);
}
// g2_addition_step29 with its thirteen products in four levels (2 + 4 + 3 + 4 lane pairs); inputs and outputs replicated.  Levels 1 and 3
// meet in the P slots, levels 2 and 4 in the T slots (a barrier separates every reuse).
template <int EPW = 1>
BN_DEV void g2_addition_step29_wide(G2W& r, const W2& bx, const W2& by, W2& l0, W2& l1, W2& l2, const F29& px, const F29& py, WideLdsPtr x) {
  const int lane = (int)(threadIdx.x & 63u), odd = pair_role((u32)lane), j = wide_j<EPW>(lane);
  {   // level 1: z bx, z by
    const W2 pr = w2_mul_w(r.z, w2_pick(bx, by, j == 1));
    if (j < 2) wide_put(x, WL_P + j, odd, pr);
  }
  __syncthreads();
  const W2 d = w2_sub(r.x, wide_get(x, WL_P, odd)), e = w2_sub(r.y, wide_get(x, WL_P + 1, odd));      // D
  const W2 dn = w2_norm(d), en = w2_norm(e);
  {   // level 2: e bx, d by, dn^2, en^2 -- and the scaled line on two idle lane pairs: 4: l2 x_P = -e x_P, 5: l1 y_P = d y_P (slots P + 4, P + 5)
    const int p = j < 6 ? j : 0;
    const W2 u = w2_pick(w2_pick(w2_pick(w2_pick(w2_pick(e, d, p == 1), dn, p == 2), en, p == 3), w2_neg(e), p == 4), d, p == 5);
    const W2 v = w2_pick(w2_pick(w2_pick(w2_pick(bx, by, p == 1), dn, p == 2), en, p == 3), w2_real(sel9(p == 5, px, py)), p >= 4);
    const W2 pr = w2_mul_w(u, v);
    if (j < 6) wide_put(x, p < 4 ? WL_T + p : WL_P + p, odd, pr);
  }
  __syncthreads();
  l2 = wide_get(x, WL_P + 4, odd);
  l1 = wide_get(x, WL_P + 5, odd);
  l0 = w2_xi_lin(w2_sub(wide_get(x, WL_T, odd), wide_get(x, WL_T + 1, odd)), 1, bx, 0);               // xi (e bx - d by)
  const W2 f = wide_get(x, WL_T + 2, odd), e2 = wide_get(x, WL_T + 3, odd);
  {   // level 3: dn f, x f, z en^2
    const int p = j < 3 ? j : 0;
    const W2 u = w2_pick(w2_pick(dn, r.x, p == 1), r.z, p == 2);
    const W2 v = w2_pick(f, e2, p == 2);
    const W2 pr = w2_mul_w(u, v);
    if (j < 3) wide_put(x, WL_P + p, odd, pr);
  }
  __syncthreads();
  const W2 h = wide_get(x, WL_P, odd), i = wide_get(x, WL_P + 1, odd), ze2 = wide_get(x, WL_P + 2, odd);
  const W2 jj = w2_norm(w2_sub(w2_add(ze2, h), w2_add(i, i)));                                           // z e^2 + h - 2 i
  {   // level 4: z h, dn j, en (i - j), h y
    const int p = j < 4 ? j : 0;
    const W2 u = w2_pick(w2_pick(w2_pick(r.z, dn, p == 1), en, p == 2), h, p == 3);
    const W2 v = w2_pick(w2_pick(w2_pick(h, jj, p == 1), w2_sub(i, jj), p == 2), r.y, p == 3);
    const W2 pr = w2_mul_w(u, v);
    if (j < 4) wide_put(x, WL_T + 4 + p, odd, pr);
  }
  __syncthreads();
  r.z = wide_get(x, WL_T + 4, odd);
  r.x = wide_get(x, WL_T + 5, odd);
  r.y = w2_norm(w2_sub(wide_get(x, WL_T + 6, odd), wide_get(x, WL_T + 7, odd)));                         // e (i - j) - h y
}
// The Miller loop of ONE pair on a whole wavefront (the other tail of the one-boolean shapes): the G2 steps run replicated, the
// accumulator's squaring and its product with each line are w12_mul_wide -- a line (l0, l4 = l1 y_P, l2 = l2 x_P) is the Fp12 element
// (l0, 0, l2; 0, l4, 0) of mul_by_024 (fp12.rs:426-503), and spread over 18 lane pairs the dense product costs less than the 13-product
// sparse form on one.  With ISO = false: same field values step by step, so the raw Miller value (canonical at the exit) is the reference's.
// ISO: on (phi P, phi Q) with the twist constant 9 - u (g2_doubling_step29): the value is the reference's Miller value times a factor in Fp* --
// every caller feeds it to a final exponentiation (the raw-value entry points use the lane-pair kernels with ISO = false)
template <bool ISO, int EPW = 1>
BN_NOINLINE void miller_loop29_wide(S12& fout, const Fp& pxs, const Fp& pys, const S2& qxs_in, const S2& qys_in, WideLds* xg) {
  F29 px = f29_reduce(f29_from_fp(pxs)), py = f29_reduce(f29_from_fp(pys));
  W2 qx = w2_from_s2(qxs_in), qy = w2_from_s2(qys_in);
  S2 qxs = qxs_in, qys = qys_in;
  if (ISO) {
    px = f29_mul(px, f29_iso_s2()); py = f29_mul(py, f29_iso_s3());
    qx = w2_scale(qx, f29_iso_s2()); qy = w2_scale(qy, f29_iso_s3());
    qxs = w2_to_s2(qx); qys = w2_to_s2(qy);
  }
  const W2 zero{F29{{0, 0, 0, 0, 0, 0, 0, 0, 0}}};
  G2W r{qx, qy, w2_from_s2(s2_one())};
  W12 f;
  {
    S12 one = s12_one();
    w12_from_s12(f, one);
  }
  W2 l0, l1, l2;
  auto line = [&]() {
    W12 ln;
    ln.c0.c0 = l0; ln.c0.c1 = zero; ln.c0.c2 = l2;             // l1, l2 come out of the steps already scaled by y_P, x_P
    ln.c1.c0 = zero; ln.c1.c1 = l1; ln.c1.c2 = zero;
    if (BN_WIDE_MILLER_INL >= 2) f = w12_mul_wide<EPW, BN_WIDE_KINDS ? WK_LINE : WK_DENSE>(f, ln, (WideLdsPtr)xg);
    else w12_mul_wide_nl<EPW, BN_WIDE_KINDS ? WK_LINE : WK_DENSE>(f, f, ln, xg);
  };
  const u64 nz = BN_ATE_NAF_NZ, ng = BN_ATE_NAF_NEG;
#pragma unroll 1
  for (int i = 0; i < 64; ++i) {
    if (BN_WIDE_MILLER_INL >= 1) f = w12_mul_wide<EPW, BN_WIDE_KINDS ? WK_SQUARE : WK_DENSE>(f, f, (WideLdsPtr)xg);
    else w12_mul_wide_nl<EPW, BN_WIDE_KINDS ? WK_SQUARE : WK_DENSE>(f, f, f, xg);
    g2_doubling_step29_wide<ISO, EPW>(r, l0, l1, l2, px, py, (WideLdsPtr)xg);
    line();
    if ((nz >> (63 - i)) & 1) {
      g2_addition_step29_wide<EPW>(r, qx, ((ng >> (63 - i)) & 1) ? w2_neg(qy) : qy, l0, l1, l2, px, py, (WideLdsPtr)xg);
      line();
    }
  }
  S2 q1x, q1y, q2x, q2y;
  g2_psi_affine(q1x, q1y, qxs, qys);
  g2_psi_affine(q2x, q2y, q1x, q1y);
  g2_addition_step29_wide<EPW>(r, w2_from_s2(q1x), w2_from_s2(q1y), l0, l1, l2, px, py, (WideLdsPtr)xg);
  line();
  g2_addition_step29_wide<EPW>(r, w2_from_s2(q2x), w2_from_s2(s2_neg(q2y)), l0, l1, l2, px, py, (WideLdsPtr)xg);
  line();
  w12_to_s12(fout, f);
}

}  // namespace pl
}  // namespace bn254
