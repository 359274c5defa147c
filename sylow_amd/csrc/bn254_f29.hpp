// Carry-free field core: Fp elements as 9 signed limbs of 29 bits
// (value = sum v[i] 2^(29 i)), Montgomery factor R' = 2^261.
//
// Why: in the saturated 8 x 32 form every partial product costs a v_mad_u64_u32 AND a v_addc (the mad has no
// carry-in); with 29-bit limbs a 64-bit column accumulator absorbs a whole column (9 products of < 2^58, plus the
// 9 reduction products) with no carry handling, so a product costs one mad -- measured 1.37x faster in isolation
// (tools/ubench/f29_bench.hip) -- and add/sub are 9 limb-wise instructions with no reduction at all.
//
// Bounds discipline (all worst-case, not probabilistic).  L(x) = max_i<8 |x.v[i]| / 2^29, V(x) = |value| / p.
//  * "normalized": v[0..7] in [0, 2^29), v[8] signed with |v[8]| < 2^28           (L <= 1)
//  * lazy add/sub/neg: limb-wise, L and V add; a difference of two normalized values has L <= 1, a sum L <= 2;
//    every limb must stay inside int32: L <= 3
//  * f29_mul(a, b): requires 9 L(a) L(b) + 9 < 31.9, i.e. L(a) L(b) <= 2.5  (signed 64-bit columns);
//    output normalized, value in (-(V(a)V(b)/169) p, (V(a)V(b)/169 + 1) p)  [p / 2^261 = 1/169.3]
//  * every function states the bounds it needs; the builders of Fp2/Fp12 routines keep V <= 40 so that
//    |v[8]| <= V p / 2^232 < 2^28 holds everywhere.
// Values enter from / leave to the saturated Montgomery (R = 2^256) form through f29_from_fp / f29_to_fp, which
// also change the Montgomery factor (x 2^5 in, exact / 2^5 out), so both cores compute the same residues and
// every routine here is checked bit-for-bit against its saturated twin (tests/test_gpu_fields.py).
#pragma once
#include "bn254_tower.hpp"

namespace bn254 {

typedef int32_t i32;
typedef int64_t i64;

struct F29 { i32 v[9]; };
#define BN_M29 0x1fffffff
#define BN_PINV29 0x04866389u          // -p^-1 mod 2^29

BN_DEV void f29_p(i32 (&p)[9]) {
  p[0] = 0x187cfd47; p[1] = 0x010460b6; p[2] = 0x1c72a34f; p[3] = 0x02d522d0; p[4] = 0x1585d978;
  p[5] = 0x02db40c0; p[6] = 0x00a6e141; p[7] = 0x0e5c2634; p[8] = 0x0030644e;
}

// ---- lazy linear operations (no carries, no reduction) -------------------------------------------------
BN_DEV F29 f29_add(const F29& a, const F29& b) { F29 r;
#pragma unroll
  for (int i = 0; i < 9; ++i) r.v[i] = a.v[i] + b.v[i];
  return r; }
BN_DEV F29 f29_sub(const F29& a, const F29& b) { F29 r;
#pragma unroll
  for (int i = 0; i < 9; ++i) r.v[i] = a.v[i] - b.v[i];
  return r; }
BN_DEV F29 f29_neg(const F29& a) { F29 r;
#pragma unroll
  for (int i = 0; i < 9; ++i) r.v[i] = -a.v[i];
  return r; }
BN_DEV F29 f29_dbl(const F29& a) { return f29_add(a, a); }

// carry propagation of a lazy value (any L <= 3): limbs 0..7 back into [0, 2^29), top limb keeps the sign
BN_DEV F29 f29_norm(const F29& a) {
  F29 r;
  i32 c = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    i32 t = a.v[i] + c;
    r.v[i] = t & BN_M29;
    c = t >> 29;                 // arithmetic shift: floor
  }
  r.v[8] = a.v[8] + c;
  return r;
}
// norm(8 a) for an N-class a (low limbs in [0, 2^29)): 8 a_i + carry < 2^32, carried through unsigned
BN_DEV F29 f29_norm_x8(const F29& a) {
  F29 r;
  u32 c = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const u32 t = ((u32)a.v[i] << 3) + c;
    r.v[i] = (i32)(t & BN_M29);
    c = t >> 29;
  }
  r.v[8] = a.v[8] * 8 + (i32)c;
  return r;
}
// norm(a - 3 b) for |limbs| < 2^29: every a_i - 3 b_i + carry stays inside 32 bits
BN_DEV F29 f29_norm_sub3(const F29& a, const F29& b) {
  F29 r;
  i32 c = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const i32 t = a.v[i] - 3 * b.v[i] + c;
    r.v[i] = t & BN_M29;
    c = t >> 29;
  }
  r.v[8] = a.v[8] - 3 * b.v[8] + c;
  return r;
}

// ---- Montgomery product, R' = 2^261 ----------------------------------------------------------------------
// ONE 64-bit accumulator is carried through all 17 columns: every partial product is a v_mad_i64_i32 whose addend is the running
// sum, the carry of a column (acc >> 29) is the addend of the next column's first product.  Left to itself the compiler starts
// each column from zero and merges the carry with an extra 64-bit add (17 x v_lshl_add_u64, quarter rate); BN_CHAIN -- an empty,
// input-only volatile asm: the running value must exist at that point, no instruction is emitted -- pins the chain.  Measured on
// the lane-pair product leaf at 2 waves/SIMD (tools/ubench/dot2_bench.hip, alternating A/B): 4-5 % faster than the two-accumulator
// column form, whose point was instruction-level parallelism the two resident waves already provide (schedules with 4..24-long
// dependent runs time the same).
// History: on the one-element-per-lane kernels this core executed 27 % fewer VALU instructions in the f^x loops but ran only
// 4.5 % faster -- those kernels were bound by scratch traffic (an Fp12 on this core is 108 VGPRs), not by issue; on lane
// pairs (bn254_pair29.hpp) the kernel is issue-bound and the instruction saving shows up in full.
#define BN_CHAIN(x) asm volatile("" ::"v"(x))
// the same pin for code that is inlined next to loads / stores: a volatile asm orders memory operations around it (scratch and
// LDS reads could no longer be hoisted across the chain), an in/out operand does not -- at the price of an s_nop the compiler
// adds after each one
#define BN_CHAIN_NV(x) asm("" : "+v"(x))
// requires L(a) L(b) <= 2.5; output normalized
BN_DEV F29 f29_mul(const F29& a, const F29& b) {
  i32 p[9]; f29_p(p);
  i32 m[9];
  F29 r;
  i64 acc = 0;
#pragma unroll
  for (int k = 0; k < 17; ++k) {
    const int lo = k > 8 ? k - 8 : 0, hi = k < 8 ? k : 8;
#pragma unroll
    for (int i = lo; i <= hi; ++i) { acc += (i64)a.v[i] * b.v[k - i]; BN_CHAIN(acc); }
#pragma unroll
    for (int i = lo; i <= hi; ++i) {
      if (k < 9 && i == k) continue;                   // m[k] is not known yet
      acc += (i64)m[i] * p[k - i]; BN_CHAIN(acc);
    }
    if (k < 9) {
      m[k] = (i32)(((u32)acc * BN_PINV29) & BN_M29);
      acc += (i64)m[k] * p[0]; BN_CHAIN(acc);
    } else {
      r.v[k - 9] = (i32)((u32)acc & BN_M29);
    }
    acc >>= 29;
  }
  r.v[8] = (i32)acc;
  return r;
}
// (a*b + c*d) / R': one column pass, one reduction.  Requires 9 (L(a)L(b) + L(c)L(d)) + 9 < 31.9, i.e. all four
// operands normalized (27 < 31.9).  Output normalized, value in (-(VaVb+VcVd)/169 p, ((VaVb+VcVd)/169 + 1) p).
BN_DEV F29 f29_dot2(const F29& a, const F29& b, const F29& c, const F29& d) {
  i32 p[9]; f29_p(p);
  i32 m[9];
  F29 r;
  i64 acc = 0;
#pragma unroll
  for (int k = 0; k < 17; ++k) {
    const int lo = k > 8 ? k - 8 : 0, hi = k < 8 ? k : 8;
#pragma unroll
    for (int i = lo; i <= hi; ++i) {
      acc += (i64)a.v[i] * b.v[k - i]; BN_CHAIN(acc);
      acc += (i64)c.v[i] * d.v[k - i]; BN_CHAIN(acc);
    }
#pragma unroll
    for (int i = lo; i <= hi; ++i) {
      if (k < 9 && i == k) continue;
      acc += (i64)m[i] * p[k - i]; BN_CHAIN(acc);
    }
    if (k < 9) {
      m[k] = (i32)(((u32)acc * BN_PINV29) & BN_M29);
      acc += (i64)m[k] * p[0]; BN_CHAIN(acc);
    } else {
      r.v[k - 9] = (i32)((u32)acc & BN_M29);
    }
    acc >>= 29;
  }
  r.v[8] = (i32)acc;
  return r;
}

// The two-accumulator column form of the same pass (the column's products are split over two sums that are merged once per column,
// the carry joins at the end): 16 more instructions than the chained form but short dependent runs.  The chained form wins wherever
// both resident waves spend their time in product leaves (the pairing kernels: -3.6 %); the G2 group law (few leaf calls between
// carry normalisations, waves out of step) measured 5 % FASTER on this form, so OpsW2 keeps it (plk_group.hip).
BN_DEV F29 f29_dot2_ilp(const F29& a, const F29& b, const F29& c, const F29& d) {
  i32 p[9]; f29_p(p);
  i32 m[9];
  F29 r;
  i64 acc = 0;
#pragma unroll
  for (int k = 0; k < 17; ++k) {
    i64 x = acc, y = 0;
    const int lo = k > 8 ? k - 8 : 0, hi = k < 8 ? k : 8;
#pragma unroll
    for (int i = lo; i <= hi; ++i) { x += (i64)a.v[i] * b.v[k - i]; y += (i64)c.v[i] * d.v[k - i]; }
#pragma unroll
    for (int i = lo; i <= hi; ++i) {
      if (k < 9 && i == k) continue;
      if ((i - lo) & 1) x += (i64)m[i] * p[k - i]; else y += (i64)m[i] * p[k - i];
    }
    acc = x + y;
    if (k < 9) {
      m[k] = (i32)(((u32)acc * BN_PINV29) & BN_M29);
      acc += (i64)m[k] * p[0];
    } else {
      r.v[k - 9] = (i32)((u32)acc & BN_M29);
    }
    acc >>= 29;
  }
  r.v[8] = (i32)acc;
  return r;
}

// ---- conversions ---------------------------------------------------------------------------------------------
// saturated Montgomery (x * 2^256, canonical 8 x 32) -> this core (x * 2^261): the digits of (X << 5).
// Output normalized, value = 32 X in [0, 32 p): V <= 32.
BN_DEV F29 f29_from_fp(const Fp& x) {
  F29 r;
  r.v[0] = (i32)((x.v[0] << 5) & BN_M29);
#pragma unroll
  for (int i = 1; i < 9; ++i) {
    const int bit = 29 * i - 5;          // first source bit of digit i
    const int w = bit >> 5, s = bit & 31;
    u32 lo = x.v[w] >> s;
    u32 hi = (s != 0 && w + 1 < 8) ? (x.v[w + 1] << (32 - s)) : 0u;
    r.v[i] = (i32)((lo | hi) & BN_M29);
  }
  return r;
}
// this core -> saturated canonical Montgomery form.  Input normalized with value in (-64 p, 64 p).
// v + 64 p >= 0; exact division by 32 (a 5-bit Montgomery step: add m p with m = -v p^-1 mod 32); the quotient is
// < 5 p, reduced by conditional subtraction of 4p, 2p, p.
BN_DEV Fp f29_to_fp(const F29& a) {
  i32 p[9]; f29_p(p);
  // digits of 64 p (p << 6), normalized: precomputed
  const i32 k64[9] = {0x1f3f51c0, 0x01182db0, 0x1ca8d3c2, 0x1548b438, 0x01765e05, 0x16d0302b, 0x09b85045, 0x17098d01, 0x0c19139c};
  // m = (-(v + 64p) * p^-1) mod 32, from the low limb
  u32 low = (u32)(a.v[0] + k64[0]);
  u32 m = (low * BN_PINV29) & 31u;
  // w = (a + 64 p + m p) as normalized digits, then >> 5 while re-packing into 32-bit words
  i64 acc = 0;
  u32 d[9];
#pragma unroll
  for (int i = 0; i < 9; ++i) {
    acc += (i64)a.v[i] + k64[i] + (i64)m * p[i];
    if (i < 8) { d[i] = (u32)acc & BN_M29; acc >>= 29; } else d[i] = (u32)acc;
  }
  // value = sum d[i] 2^(29 i), divisible by 32; out = value >> 5 (< 5p < 2^257: 9th word may hold one bit)
  u32 o[9];
#pragma unroll
  for (int w = 0; w < 9; ++w) o[w] = 0;
#pragma unroll
  for (int i = 0; i < 9; ++i) {
    const int bit = 29 * i - 5;          // destination bit of digit i (digit 0 loses its low 5 zero bits)
    if (i == 0) { o[0] |= d[0] >> 5; continue; }
    const int w = bit >> 5, s = bit & 31;
    o[w] |= d[i] << s;
    if (s > 3 && w + 1 < 9) o[w + 1] |= d[i] >> (32 - s);     // a 29-bit digit (32-bit for the top one) spills past the word when s > 3
  }
  u32 r[8] = {o[0], o[1], o[2], o[3], o[4], o[5], o[6], o[7]};
  // o[8] can only be 0 or 1 (value < 5p < 2^257); fold it in by subtracting 4p first using the 9-word compare
  // 4p = 0xc19139cb... < 2^256: if o[8] is set the value is >= 2^256 > 4p, so subtract 4p unconditionally then
  {
    u32 s4[8];
    u32 bor;
    const u32 c0 = 0x61f3f51cu, c1 = 0xf082305bu, c2 = 0xa1c72a34u, c3 = 0x5e05aa45u, c4 = 0x06056176u, c5 = 0xe14116dau, c6 = 0x84c680a6u, c7 = 0xc19139cbu;
    asm("v_sub_co_u32 %0, vcc, %9, %17\n\t"
        "v_subb_co_u32 %1, vcc, %10, %18, vcc\n\t"
        "v_subb_co_u32 %2, vcc, %11, %19, vcc\n\t"
        "v_subb_co_u32 %3, vcc, %12, %20, vcc\n\t"
        "v_subb_co_u32 %4, vcc, %13, %21, vcc\n\t"
        "v_subb_co_u32 %5, vcc, %14, %22, vcc\n\t"
        "v_subb_co_u32 %6, vcc, %15, %23, vcc\n\t"
        "v_subb_co_u32 %7, vcc, %16, %24, vcc\n\t"
        "v_subb_co_u32 %8, vcc, %25, 0, vcc"
        : "=&v"(s4[0]), "=&v"(s4[1]), "=&v"(s4[2]), "=&v"(s4[3]), "=&v"(s4[4]), "=&v"(s4[5]), "=&v"(s4[6]), "=&v"(s4[7]), "=&v"(bor)
        : "v"(r[0]), "v"(r[1]), "v"(r[2]), "v"(r[3]), "v"(r[4]), "v"(r[5]), "v"(r[6]), "v"(r[7]), "v"(c0), "v"(c1), "v"(c2), "v"(c3),
          "v"(c4), "v"(c5), "v"(c6), "v"(c7), "v"(o[8])
        : "vcc");
#pragma unroll
    for (int i = 0; i < 8; ++i) r[i] = (bor != 0) ? r[i] : s4[i];
  }
  cond_sub_const(r, 0xb0f9fa8eu, 0x7841182du, 0xd0e3951au, 0x2f02d522u, 0x0302b0bbu, 0x70a08b6du, 0xc2634053u, 0x60c89ce5u);  // 2p
  cond_sub_const(r, BN_P0, BN_P1, BN_P2, BN_P3, BN_P4, BN_P5, BN_P6, BN_P7);
  return fp_from_limbs(r[0], r[1], r[2], r[3], r[4], r[5], r[6], r[7]);
}


// ---- reduce-and-normalise pass ------------------------------------------------------------------------------------
// Lazy linear combinations keep the LIMBS small but not the VALUE (3t - 2z doubles the bound every squaring), so
// stored results go through one pass that subtracts q p with q = round(value / p) estimated from the top limb and
// propagates carries at the same time: r normalised, |value(r)| < 0.51 p.
// `limb(i)` returns limb i of the lazy combination as int64; requirements: |limb(i)| < 2^36 for i < 8 and
// |limb(8)| < 2^31.  Estimate accuracy: value / 2^232 = limb(8) + eps with |eps| < 2^8, p / 2^232 = 3171406.4,
// K = floor(2^44 / 3171406): |value - q p| < p/2 + (2^8 + 1) 2^232 + 2^-13 p < 0.51 p.
template <class LIMB>
BN_DEV F29 f29_reduce_from(LIMB limb) {
  i32 p[9]; f29_p(p);
  const i64 t8 = limb(8);
  const i64 q = (t8 * 5547168ll + (1ll << 43)) >> 44;
  F29 r;
  i64 acc = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    acc += limb(i) - q * p[i];
    r.v[i] = (i32)((u32)acc & BN_M29);
    acc >>= 29;
  }
  r.v[8] = (i32)(acc + t8 - q * p[8]);
  return r;
}

// The same pass for a linear combination sum_j k_j x_j of N lazy values, as ONE chain of multiply-adds: every term is a
// v_mad_i64_i32 (x_j[i] * k_j + acc -- the multiplier sign-extends the limb and the running sum is its addend), then -q p[i] the same
// way, mask, shift.  The coefficients are kept in registers the compiler cannot see through (bn_keep), otherwise it strength-reduces
// x * 3 or x * 1 into 32 -> 64-bit sign extensions, 64-bit shifts and a separate 64-bit add per term (measured: 94 VALU instructions
// for 3 t - 2 z, against 5 per limb here).  Requirements as f29_reduce_from on the combined limbs.
BN_DEV i32 bn_keep(i32 k) { asm("" : "+s"(k)); return k; }          // wave-uniform coefficient
BN_DEV i32 bn_keep_v(i32 k) { asm("" : "+v"(k)); return k; }        // lane-dependent coefficient
template <int N>
BN_DEV F29 f29_reduce_terms(const F29* const (&x)[N], const i32 (&k)[N]) {
  i32 p[9]; f29_p(p);
  i64 t8 = 0;
#pragma unroll
  for (int j = 0; j < N; ++j) t8 += (i64)x[j]->v[8] * k[j];
  const i32 nq = -(i32)((t8 * 5547168ll + (1ll << 43)) >> 44);
  F29 r;
  i64 acc = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
#pragma unroll
    for (int j = 0; j < N; ++j) { acc += (i64)x[j]->v[i] * k[j]; BN_CHAIN(acc); }
    acc += (i64)nq * p[i]; BN_CHAIN(acc);
    r.v[i] = (i32)((u32)acc & BN_M29);
    acc >>= 29;
  }
  r.v[8] = (i32)(acc + t8 + (i64)nq * p[8]);
  return r;
}

// Carry normalisation of a linear combination sum_j k_j x_j WITHOUT the reduce step (no multiple of p is subtracted): one chain of
// multiply-adds, mask, shift; limbs 0..7 in [0, 2^29), top limb signed: an N-class value with V = sum |k_j| V(x_j).  For combinations
// whose VALUE stays small enough to be a product operand (|V| <= 8) but whose limbs do not fit 32-bit lazy arithmetic (the x 9 of xi).
template <int N>
BN_DEV F29 f29_norm_terms(const F29* const (&x)[N], const i32 (&k)[N]) {
  F29 r;
  i64 acc = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
#pragma unroll
    for (int j = 0; j < N; ++j) { acc += (i64)x[j]->v[i] * k[j]; BN_CHAIN(acc); }
    r.v[i] = (i32)((u32)acc & BN_M29);
    acc >>= 29;
  }
#pragma unroll
  for (int j = 0; j < N; ++j) acc += (i64)x[j]->v[8] * k[j];
  r.v[8] = (i32)acc;
  return r;
}

// ---- out-of-line product leaf: 18 scalar ABI arguments (two 9-limb structs would travel through the stack) -------------
// operands R / N / D class, L(a) L(b) <= 2.5; output normalized
BN_NOINLINE F29 f29_mul_leaf(i32 a0, i32 a1, i32 a2, i32 a3, i32 a4, i32 a5, i32 a6, i32 a7, i32 a8,
                             i32 b0, i32 b1, i32 b2, i32 b3, i32 b4, i32 b5, i32 b6, i32 b7, i32 b8) {
  return f29_mul(F29{{a0, a1, a2, a3, a4, a5, a6, a7, a8}}, F29{{b0, b1, b2, b3, b4, b5, b6, b7, b8}});
}
#define W_ARGS(x) (x).v[0], (x).v[1], (x).v[2], (x).v[3], (x).v[4], (x).v[5], (x).v[6], (x).v[7], (x).v[8]

// ---- squaring and fixed-exponent powers ----------------------------------------------------------------------------------
// a^2 / R': the 36 cross products are taken once against the doubled operand (45 multiply-adds instead of 81), same
// column bound as f29_mul.  Input N-class (L <= 1), output normalized.
BN_DEV F29 f29_sqr(const F29& a) {
  i32 p[9]; f29_p(p);
  i32 m[9];
  i32 a2[9];
#pragma unroll
  for (int i = 0; i < 9; ++i) a2[i] = a.v[i] * 2;
  F29 r;
  i64 acc = 0;
#pragma unroll
  for (int k = 0; k < 17; ++k) {
    const int lo = k > 8 ? k - 8 : 0, hi = k < 8 ? k : 8;
#pragma unroll
    for (int i = lo; 2 * i < k; ++i) { acc += (i64)a.v[i] * a2[k - i]; BN_CHAIN(acc); }
    if ((k & 1) == 0) { acc += (i64)a.v[k / 2] * a.v[k / 2]; BN_CHAIN(acc); }
#pragma unroll
    for (int i = lo; i <= hi; ++i) {
      if (k < 9 && i == k) continue;
      acc += (i64)m[i] * p[k - i]; BN_CHAIN(acc);
    }
    if (k < 9) {
      m[k] = (i32)(((u32)acc * BN_PINV29) & BN_M29);
      acc += (i64)m[k] * p[0]; BN_CHAIN(acc);
    } else {
      r.v[k - 9] = (i32)((u32)acc & BN_M29);
    }
    acc >>= 29;
  }
  r.v[8] = (i32)acc;
  return r;
}
// out-of-line squaring leaf (45 + 81 multiply-adds against the 81 + 81 of f29_mul_leaf(a, a)): the group law's squarings
BN_NOINLINE F29 f29_sqr_leaf(i32 a0, i32 a1, i32 a2, i32 a3, i32 a4, i32 a5, i32 a6, i32 a7, i32 a8) {
  return f29_sqr(F29{{a0, a1, a2, a3, a4, a5, a6, a7, a8}});
}
// a^e for a 256-bit exponent given as 8 wave-uniform words: 4-bit fixed windows over the carry-free core (254 squarings of
// 126 multiply-adds + at most 64 products + 14 for the table, against 381 saturated Montgomery products of 256 multiply-add
// / add-carry pairs each).  0^e = 0, so inv(0) = 0 as in the reference (fp.rs:418-433, test fp.rs:1126-1132).
BN_NOINLINE Fp fp_pow_words(Fp a, u32 e0, u32 e1, u32 e2, u32 e3, u32 e4, u32 e5, u32 e6, u32 e7) {
  const u32 e[8] = {e0, e1, e2, e3, e4, e5, e6, e7};
  const F29 af = f29_from_fp(a);
  const F29 x = f29_reduce_from([&](int i) { return (i64)af.v[i]; });
  // The window table lives in the lane's scratch frame (dynamic index).  Each entry is fetched ONCE into registers, before
  // the last squaring of its window so that the squaring covers the latency, and pinned there: left to itself the compiler
  // re-reads limbs at their points of use inside the product (~36 scratch loads per product, each one a full wait at two waves
  // per SIMD -- the square root was scratch-latency-bound, profiles/r03_configs hash_to_g1 before / after).
  F29 tab[16];
  tab[0] = x; tab[1] = x;
  F29 r = x;
#pragma unroll 1
  for (int i = 2; i < 16; ++i) { r = f29_mul(r, x); tab[i] = r; }
  r = x;
  bool started = false;
#pragma unroll 1
  for (int w = 7; w >= 0; --w) {
    const u32 word = e[w];
#pragma unroll 1
    for (int nib = 7; nib >= 0; --nib) {
      const u32 idx = (word >> (4 * nib)) & 15u;
      if (started) {
#pragma unroll 1
        for (int j = 0; j < 3; ++j) r = f29_sqr(r);
        F29 t = tab[idx];
        r = f29_sqr(r);
#pragma unroll
        for (int i = 0; i < 9; ++i) BN_CHAIN_NV(t.v[i]);
        if (idx) r = f29_mul(r, t);
      } else if (idx) {
        r = tab[idx];
        started = true;
      }
    }
  }
  if (!started) {                                  // e == 0: the Montgomery one
    const F29 one = f29_from_fp(fp_one());
    r = f29_reduce_from([&](int i) { return (i64)one.v[i]; });
  }
  return f29_to_fp(r);
}

// a^((p - 3) / 4), the chain behind the square root (a * t) and the Legendre symbol (a * t^2) of the SvdW map: the exponent is fixed, so
// its 4-bit SLIDING windows are precomputed -- first window x^3, then 48 steps "n squarings, multiply by the odd power x^(2 idx + 1)",
// one byte each (n in bits 0-4, idx in bits 5-7): 250 squarings + 48 products + 8 for the table of odd powers, against 252 + 63 + 14 of
// the fixed windows of fp_pow_words.  (Schedule: scan (p - 3) / 4 from the top, a window = the longest run of at most 4 bits that ends
// in a set bit; tests/test_gpu_fields.py checks the result against Python's pow on structured and random inputs.)
BN_NOINLINE Fp fp_pow_pm3_quarter_chain(Fp a) {
  const F29 af = f29_from_fp(a);
  const F29 x = f29_reduce_from([&](int i) { return (i64)af.v[i]; });
  F29 tab[8];                                       // x, x^3 .. x^15 (scratch frame: fetched once per window, see fp_pow_words)
  tab[0] = x;
  F29 r = x;
  {
    const F29 x2 = f29_sqr(x);
#pragma unroll 1
    for (int i = 1; i < 8; ++i) { r = f29_mul(r, x2); tab[i] = r; }
  }
  r = tab[1];
#pragma unroll 1
  for (int wd = 0; wd < 6; ++wd) {
    u64 sw = wd == 0 ? 0x8801a66522870327ull : wd == 1 ? 0xa7064722c64ac701ull : wd == 2 ? 0xa6a4a823492843c5ull
           : wd == 3 ? 0x666702a78544aa22ull : wd == 4 ? 0xa9250605e6c70245ull : 0x040243e484e801c5ull;
#pragma unroll 1
    for (int b = 0; b < 8; ++b) {
      const int nsq = (int)(sw & 31u);
      const u32 idx = (u32)(sw >> 5) & 7u;
      sw >>= 8;
#pragma unroll 1
      for (int j = 1; j < nsq; ++j) r = f29_sqr(r);
      F29 t = tab[idx];
      r = f29_sqr(r);
#pragma unroll
      for (int k = 0; k < 9; ++k) BN_CHAIN_NV(t.v[k]);
      r = f29_mul(r, t);
    }
  }
  return f29_to_fp(r);
}

// ---- modular inversion: Bernstein-Yang safegcd on 30-bit signed limbs ------------------------------------------------------
// (delta, f, g) -> divstep^600 in 20 batches of 30: each batch derives a 2x2 transition matrix t from the low 30 bits of f, g
// (branch-free: conditional negate / add through masks), then applies it to (f, g) exactly (/ 2^30) and to (d, e) modulo p
// (adding the multiple of p that makes the low 30 bits vanish).  g reaches 0, f = +-1, d = +-x^-1.  Start delta = 1/2
// (zeta = -1), for which 590 divsteps suffice for 256-bit inputs.  tools/safegcd_model.py is the bit-level Python model that
// checks every invariant asserted in the comments below and prints the constants.
// In: X = x R canonical (the saturated Montgomery form).  Out: x^-1 R = X^-1 R^2 = montmul(X^-1, R^3).  X = 0 -> 0.
BN_NOINLINE Fp fp_inv_safegcd(Fp a) {
  const i32 P30[9] = {0x187cfd47, 0x3082305b, 0x071ca8d3, 0x205aa45a, 0x01585d97, 0x0116da06, 0x1a029b85, 0x139cb84c, 0x00003064};
  const u32 PINV30 = 0x1b799c77u;          // p^-1 mod 2^30
  const i32 M30 = 0x3fffffff;
  i32 f[9], g[9], d[9], e[9];
#pragma unroll
  for (int i = 0; i < 9; ++i) {
    const int bit = 30 * i, w = bit >> 5, sh = bit & 31;
    u32 lo = a.v[w] >> sh;
    u32 hi = (sh > 2 && w + 1 < 8) ? (a.v[w + 1] << (32 - sh)) : 0u;
    g[i] = (i32)((lo | hi) & (u32)M30);
    f[i] = P30[i];
    d[i] = 0;
    e[i] = (i == 0) ? 1 : 0;
  }
  i32 zeta = -1;
#pragma unroll 1
  for (int it = 0; it < 20; ++it) {
    // g = 0 is a fixed point (f and d no longer change): random inputs get there after ~512 divsteps, so the last one or two
    // batches are usually idle for the whole wavefront
    if (it >= 16 && !__any((g[0] | g[1] | g[2] | g[3] | g[4] | g[5] | g[6] | g[7] | g[8]) != 0)) break;
    // 30 divsteps on the low limbs -> t = [[u, v], [q, r]] (entries in (-2^30, 2^30])
    u32 u = 1, v = 0, q = 0, r = 1, ff = (u32)f[0], gg = (u32)g[0];
#pragma unroll 5
    for (int st = 0; st < 30; ++st) {
      u32 c1 = (u32)(zeta >> 31);
      const u32 c2 = 0u - (gg & 1u);
      const u32 x = (ff ^ c1) - c1, y = (u ^ c1) - c1, z = (v ^ c1) - c1;
      gg += x & c2; q += y & c2; r += z & c2;
      c1 &= c2;
      zeta = (i32)(((u32)zeta ^ c1) - 1u);
      ff += gg & c1; u += q & c1; v += r & c1;
      gg >>= 1; u <<= 1; v <<= 1;
    }
    const i32 tu = (i32)u, tv = (i32)v, tq = (i32)q, tr = (i32)r;
    // (d, e) <- t (d, e) / 2^30 mod p; d, e stay in (-2p, p)
    {
      const i32 sd = d[8] >> 31, se = e[8] >> 31;
      i32 md = (tu & sd) + (tv & se), me = (tq & sd) + (tr & se);
      i64 cd = (i64)tu * d[0] + (i64)tv * e[0], ce = (i64)tq * d[0] + (i64)tr * e[0];
      md -= (i32)((PINV30 * (u32)cd + (u32)md) & (u32)M30);
      me -= (i32)((PINV30 * (u32)ce + (u32)me) & (u32)M30);
      cd += (i64)P30[0] * md; ce += (i64)P30[0] * me;
      cd >>= 30; ce >>= 30;                      // low 30 bits are zero by construction
#pragma unroll
      for (int i = 1; i < 9; ++i) {
        cd += (i64)tu * d[i] + (i64)tv * e[i] + (i64)P30[i] * md;
        ce += (i64)tq * d[i] + (i64)tr * e[i] + (i64)P30[i] * me;
        d[i - 1] = (i32)cd & M30; cd >>= 30;
        e[i - 1] = (i32)ce & M30; ce >>= 30;
      }
      d[8] = (i32)cd; e[8] = (i32)ce;
    }
    // (f, g) <- t (f, g) / 2^30, exact
    {
      i64 cf = (i64)tu * f[0] + (i64)tv * g[0], cg = (i64)tq * f[0] + (i64)tr * g[0];
      cf >>= 30; cg >>= 30;
#pragma unroll
      for (int i = 1; i < 9; ++i) {
        cf += (i64)tu * f[i] + (i64)tv * g[i];
        cg += (i64)tq * f[i] + (i64)tr * g[i];
        f[i - 1] = (i32)cf & M30; cf >>= 30;
        g[i - 1] = (i32)cg & M30; cg >>= 30;
      }
      f[8] = (i32)cf; g[8] = (i32)cg;
    }
  }
  // d = sign(f) x^-1 in (-2p, p): add p if negative, negate if f < 0, carry, add p again if still negative
  {
    const i32 add1 = d[8] >> 31, neg = f[8] >> 31;
#pragma unroll
    for (int i = 0; i < 9; ++i) d[i] = ((d[i] + (P30[i] & add1)) ^ neg) - neg;
#pragma unroll
    for (int i = 0; i < 8; ++i) { d[i + 1] += d[i] >> 30; d[i] &= M30; }
    const i32 add2 = d[8] >> 31;
#pragma unroll
    for (int i = 0; i < 9; ++i) d[i] += P30[i] & add2;
#pragma unroll
    for (int i = 0; i < 8; ++i) { d[i + 1] += d[i] >> 30; d[i] &= M30; }
  }
  // 9 x 30 -> 8 x 32
  Fp y;
#pragma unroll
  for (int w = 0; w < 8; ++w) y.v[w] = 0;
#pragma unroll
  for (int i = 0; i < 9; ++i) {
    const int bit = 30 * i, w = bit >> 5, sh = bit & 31;
    if (w < 8) y.v[w] |= (u32)d[i] << sh;
    if (sh > 2 && w + 1 < 8) y.v[w + 1] |= (u32)d[i] >> (32 - sh);
  }
  // X^-1 -> X^-1 R^2 (one Montgomery product with R^3 mod p)
  return fp_mul(y, fp_from_limbs(0xda1530dfu, 0xb1cd6dafu, 0xa7283db6u, 0x62f210e6u, 0x0ada0afbu, 0xef7f0b0cu, 0x2d592544u, 0x20fd6e90u));
}

// ---- Fp2 on the carry-free core ---------------------------------------------------------------------------------------
struct U2 { F29 c0, c1; };
BN_DEV U2 u2_add(const U2& a, const U2& b) { return U2{f29_add(a.c0, b.c0), f29_add(a.c1, b.c1)}; }
BN_DEV U2 u2_sub(const U2& a, const U2& b) { return U2{f29_sub(a.c0, b.c0), f29_sub(a.c1, b.c1)}; }
BN_DEV U2 u2_norm(const U2& a) { return U2{f29_norm(a.c0), f29_norm(a.c1)}; }
// (a0 b0 - a1 b1, a0 b1 + a1 b0): two fused passes.  Inputs: |limbs| < 2^29 (signs free), |V| <= 8 -> |V(out)| < 1.8
BN_DEV U2 u2_mul(const U2& a, const U2& b) {
  return U2{f29_dot2(a.c0, b.c0, f29_neg(a.c1), b.c1), f29_dot2(a.c0, b.c1, a.c1, b.c0)};
}
// ((a0+a1)(a0-a1), 2 a0 a1).  Input limbs in [0, 2^29) (so that a0 - a1 has |limbs| < 2^29), |V| <= 4
BN_DEV U2 u2_sqr(const U2& a) {
  return U2{f29_mul(f29_add(a.c0, a.c1), f29_sub(a.c0, a.c1)), f29_mul(f29_dbl(a.c0), a.c1)};
}
// reduce-and-normalise of  k * x + xi_flag * (xi * y) + z  style combinations, written out per use below
// r = reduce(ka * a + kb * b)
BN_DEV F29 f29_lin2(const F29& a, int ka, const F29& b, int kb) {
  const F29* const x[2] = {&a, &b};
  const i32 k[2] = {bn_keep(ka), bn_keep(kb)};
  return f29_reduce_terms(x, k);
}
BN_DEV U2 u2_lin2(const U2& a, int ka, const U2& b, int kb) { return U2{f29_lin2(a.c0, ka, b.c0, kb), f29_lin2(a.c1, ka, b.c1, kb)}; }
// r = reduce(k * xi * x + m * y),  xi = 9 + u:  (9 x0 - x1, x0 + 9 x1).   |x limbs|, |y limbs| < 2^31
BN_DEV U2 u2_xi_lin(const U2& x, int k, const U2& y, int m) {
  return U2{f29_reduce_from([&](int i) { return ((i64)x.c0.v[i] * 9 - x.c1.v[i]) * k + (i64)y.c0.v[i] * m; }),
            f29_reduce_from([&](int i) { return ((i64)x.c0.v[i] + (i64)x.c1.v[i] * 9) * k + (i64)y.c1.v[i] * m; })};
}

// ---- Fp6 / Fp12 values in scratch: 6 Fp2 coefficients z-ordered like Fp12 (c0.c0, c0.c1, c0.c2, c1.c0, c1.c1, c1.c2)
struct U6 { U2 c0, c1, c2; };
struct U12 { U6 c0, c1; };
BN_DEV U2 u2_from_fp2(const Fp2& a) { return U2{f29_from_fp(a.c0), f29_from_fp(a.c1)}; }
BN_DEV Fp2 u2_to_fp2(const U2& a) { return Fp2{f29_to_fp(a.c0), f29_to_fp(a.c1)}; }
BN_DEV void u12_from_fp12(U12& r, const Fp12& a) {
  r.c0.c0 = u2_from_fp2(a.c0.c0); r.c0.c1 = u2_from_fp2(a.c0.c1); r.c0.c2 = u2_from_fp2(a.c0.c2);
  r.c1.c0 = u2_from_fp2(a.c1.c0); r.c1.c1 = u2_from_fp2(a.c1.c1); r.c1.c2 = u2_from_fp2(a.c1.c2);
}
BN_DEV void u12_to_fp12(Fp12& r, const U12& a) {
  r.c0.c0 = u2_to_fp2(a.c0.c0); r.c0.c1 = u2_to_fp2(a.c0.c1); r.c0.c2 = u2_to_fp2(a.c0.c2);
  r.c1.c0 = u2_to_fp2(a.c1.c0); r.c1.c1 = u2_to_fp2(a.c1.c1); r.c1.c2 = u2_to_fp2(a.c1.c2);
}
// values entering from the saturated core have V <= 32 (f29_from_fp); one reduce pass brings them to |V| < 0.51
BN_DEV void u12_reduce(U12& r) {
  U2* c[6] = {&r.c0.c0, &r.c0.c1, &r.c0.c2, &r.c1.c0, &r.c1.c1, &r.c1.c2};
#pragma unroll
  for (int k = 0; k < 6; ++k) {
    const U2 t = *c[k];
    c[k]->c0 = f29_reduce_from([&](int i) { return (i64)t.c0.v[i]; });
    c[k]->c1 = f29_reduce_from([&](int i) { return (i64)t.c1.v[i]; });
  }
}
// conjugate (fp12.rs:381-383): negated limbs are fine as product operands (|limbs| < 2^29)
BN_DEV void u12_conj(U12& r, const U12& a) {
  r.c0 = a.c0;
  r.c1.c0 = U2{f29_neg(a.c1.c0.c0), f29_neg(a.c1.c0.c1)};
  r.c1.c1 = U2{f29_neg(a.c1.c1.c0), f29_neg(a.c1.c1.c1)};
  r.c1.c2 = U2{f29_neg(a.c1.c2.c0), f29_neg(a.c1.c2.c1)};
}

// fp6.rs:283-367 (value), Karatsuba over v with normalised pre-additions.  Inputs: |limbs| < 2^29, |V| <= 1.8.
// Outputs reduced: limbs in [0, 2^29), |V| < 0.51.
BN_NOINLINE void u6_mul(U6& r, const U6& a, const U6& b) {
  const U2 v0 = u2_mul(a.c0, b.c0);
  const U2 v1 = u2_mul(a.c1, b.c1);
  const U2 v2 = u2_mul(a.c2, b.c2);
  const U2 t0 = u2_mul(u2_norm(u2_add(a.c1, a.c2)), u2_norm(u2_add(b.c1, b.c2)));
  const U2 t1 = u2_mul(u2_norm(u2_add(a.c0, a.c1)), u2_norm(u2_add(b.c0, b.c1)));
  const U2 t2 = u2_mul(u2_norm(u2_add(a.c0, a.c2)), u2_norm(u2_add(b.c0, b.c2)));
  // r0 = v0 + xi (t0 - v1 - v2);  r1 = (t1 - v0 - v1) + xi v2;  r2 = t2 - v0 - v2 + v1
  const U2 x0 = u2_sub(u2_sub(t0, v1), v2);            // limbs in (-2^30, 2^29)
  r.c0 = u2_xi_lin(x0, 1, v0, 1);
  const U2 y1 = u2_sub(u2_sub(t1, v0), v1);
  r.c1 = u2_xi_lin(v2, 1, y1, 1);
  const U2 y2 = u2_add(u2_sub(u2_sub(t2, v0), v2), v1); // limbs in (-2^30, 2^30)
  r.c2 = U2{f29_reduce_from([&](int i) { return (i64)y2.c0.v[i]; }), f29_reduce_from([&](int i) { return (i64)y2.c1.v[i]; })};
}
// fp12.rs:229-238 (value).  Inputs/outputs as u6_mul.
BN_NOINLINE void u12_mul(U12& r, const U12& a, const U12& b) {
  U6 t0, t1, t2, sa, sb;
  u6_mul(t0, a.c0, b.c0);
  u6_mul(t1, a.c1, b.c1);
  sa.c0 = u2_norm(u2_add(a.c0.c0, a.c1.c0)); sa.c1 = u2_norm(u2_add(a.c0.c1, a.c1.c1)); sa.c2 = u2_norm(u2_add(a.c0.c2, a.c1.c2));
  sb.c0 = u2_norm(u2_add(b.c0.c0, b.c1.c0)); sb.c1 = u2_norm(u2_add(b.c0.c1, b.c1.c1)); sb.c2 = u2_norm(u2_add(b.c0.c2, b.c1.c2));
  u6_mul(t2, sa, sb);                                   // |V(sa)|, |V(sb)| <= 1.1 (two reduced values)... or 3.6: both fine for u2_mul
  // c1 = t2 - t0 - t1
  r.c1.c0 = u2_lin2(u2_sub(t2.c0, t0.c0), 1, t1.c0, -1);
  r.c1.c1 = u2_lin2(u2_sub(t2.c1, t0.c1), 1, t1.c1, -1);
  r.c1.c2 = u2_lin2(u2_sub(t2.c2, t0.c2), 1, t1.c2, -1);
  // c0 = t0 + v * t1 = (t0.c0 + xi t1.c2, t0.c1 + t1.c0, t0.c2 + t1.c1)
  r.c0.c0 = u2_xi_lin(t1.c2, 1, t0.c0, 1);
  r.c0.c1 = u2_lin2(t0.c1, 1, t1.c0, 1);
  r.c0.c2 = u2_lin2(t0.c2, 1, t1.c1, 1);
}
// pairing.rs:274-350: Granger-Scott squaring in the cyclotomic subgroup.  Input: reduced (limbs in [0,2^29), |V| < 0.51)
// or any nonneg-limbed value with |V| <= 1.1; output reduced.
BN_DEV void u_fp4_square(U2& c0, U2& c1, const U2& a, const U2& b) {
  const U2 t0 = u2_sqr(a);
  const U2 t1 = u2_sqr(b);
  c0 = u2_xi_lin(t1, 1, t0, 1);                                        // xi t1 + t0, reduced
  c1 = u2_sub(u2_sub(u2_sqr(u2_norm(u2_add(a, b))), t0), t1);         // lazy, limbs in (-2^30, 2^29), |V| < 4
}
BN_NOINLINE void u12_cyclotomic_sqr(U12& r, const U12& f) {
  const U2 z0 = f.c0.c0, z4 = f.c0.c1, z3 = f.c0.c2, z2 = f.c1.c0, z1 = f.c1.c1, z5 = f.c1.c2;
  U2 t0, t1, t2, t3;
  u_fp4_square(t0, t1, z0, z1);
  r.c0.c0 = u2_lin2(t0, 3, z0, -2);                                    // z0 = 3 t0 - 2 z0
  r.c1.c1 = u2_lin2(t1, 3, z1, 2);                                     // z1 = 3 t1 + 2 z1
  u_fp4_square(t0, t1, z2, z3);
  u_fp4_square(t2, t3, z4, z5);
  r.c0.c1 = u2_lin2(t0, 3, z4, -2);                                    // z4 = 3 t0 - 2 z4
  r.c1.c2 = u2_lin2(t1, 3, z5, 2);                                     // z5 = 3 t1 + 2 z5
  r.c1.c0 = u2_xi_lin(t3, 3, z2, 2);                                   // z2 = 3 xi t3 + 2 z2
  r.c0.c2 = u2_lin2(t2, 3, z3, -2);                                    // z3 = 3 t2 - 2 z3
}

}  // namespace bn254
