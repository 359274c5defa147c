// bn254_fr.hpp -- the r-torsion scalar field Fr of BN254 (fp.rs:60-65, 556-565: the same
// define_finite_prime_field! macro as Fp, modulus r), canonical (non-Montgomery) domain.
//
// Fr sits next to the hot path, not on it (Lagrange coefficients and polynomial evaluation in
// examples/threshold_signing.rs:64-70,146-155): the batch kernels are HBM-bound, so the operands stay canonical
// and products use the same Barrett reduction as the canonical-domain Fp multiply (fp_mulmod_plain) with
// r's constants.  An Fr value is carried in the 8-limb Fp container.
#pragma once
#include "bn254_fp.hpp"

namespace bn254 {

// r = 0x30644e72e131a029b85045b68181585d2833e84879b9709143e1f593f0000001; 4r < 2^256 < 6r;
// MU = floor(2^512 / r) (tools/gen_constants.py --check verifies all four rows)
#define BN_FR_R  0xf0000001u, 0x43e1f593u, 0x79b97091u, 0x2833e848u, 0x8181585du, 0xb85045b6u, 0xe131a029u, 0x30644e72u
#define BN_FR_2R 0xe0000002u, 0x87c3eb27u, 0xf372e122u, 0x5067d090u, 0x0302b0bau, 0x70a08b6du, 0xc2634053u, 0x60c89ce5u
#define BN_FR_4R 0xc0000004u, 0x0f87d64fu, 0xe6e5c245u, 0xa0cfa121u, 0x06056174u, 0xe14116dau, 0x84c680a6u, 0xc19139cbu
#define BN_FR_MU 0xe1de9259u, 0x20703a6bu, 0x9e880ae6u, 0x14485200u, 0x80730147u, 0xb074a586u, 0x23a04a7au, 0x4a474626u, 0x00000005u

// Fr::new (fp.rs:199-201 through the macro): any 256-bit value -> canonical residue
BN_DEV Fp fr_reduce_plain(const Fp& x) {
  u32 r[8] = {x.v[0], x.v[1], x.v[2], x.v[3], x.v[4], x.v[5], x.v[6], x.v[7]};
  cond_sub_const(r, BN_FR_4R);
  cond_sub_const(r, BN_FR_2R);
  cond_sub_const(r, BN_FR_R);
  return fp_from_limbs(r[0], r[1], r[2], r[3], r[4], r[5], r[6], r[7]);
}
// canonical a, b: a + b < 2r < 2^255
BN_DEV Fp fr_add(const Fp& a, const Fp& b) {
  u32 s[8];
  u64 c = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) { c += (u64)a.v[i] + b.v[i]; s[i] = (u32)c; c >>= 32; }
  cond_sub_const(s, BN_FR_R);
  return fp_from_limbs(s[0], s[1], s[2], s[3], s[4], s[5], s[6], s[7]);
}
BN_DEV Fp fr_neg(const Fp& a) {
  const u32 r[8] = {BN_FR_R};
  u32 s[8];
  u32 nz = 0;
  int64_t c = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) { c += (int64_t)r[i] - a.v[i]; s[i] = (u32)c; c >>= 32; nz |= a.v[i]; }
#pragma unroll
  for (int i = 0; i < 8; ++i) s[i] = nz ? s[i] : 0u;
  return fp_from_limbs(s[0], s[1], s[2], s[3], s[4], s[5], s[6], s[7]);
}
BN_DEV Fp fr_sub(const Fp& a, const Fp& b) { return fr_add(a, fr_neg(b)); }

// a*b mod r, any 256-bit operands: Barrett exactly as fp_mulmod_plain (HAC 14.42, b = 2^32, k = 8; the dropped low
// partial products cost at most one unit of the quotient, so the remainder before correction is < 4r)
BN_DEV Fp fr_mulmod_inline(const Fp& a, const Fp& b) {
  const u32 p[8] = {BN_FR_R};
  const u32 mu[9] = {BN_FR_MU};
  u32 T[16];
  u64 acc = 0;
  u32 ovf = 0;
#pragma unroll
  for (int k = 0; k < 15; ++k) {
#pragma unroll
    for (int i = (k > 7 ? k - 7 : 0); i <= (k < 7 ? k : 7); ++i) mac(acc, ovf, a.v[i], b.v[k - i]);
    T[k] = (u32)acc;
    acc = (acc >> 32) | ((u64)ovf << 32);
    ovf = 0;
  }
  T[15] = (u32)acc;
  u32 q3[9];
  acc = 0; ovf = 0;
#pragma unroll
  for (int k = 7; k < 18; ++k) {
#pragma unroll
    for (int i = (k > 8 ? k - 8 : 0); i <= (k < 8 ? k : 8); ++i) mac_s(acc, ovf, T[7 + i], mu[k - i]);
    if (k >= 9) q3[k - 9] = (u32)acc;
    acc = (acc >> 32) | ((u64)ovf << 32);
    ovf = 0;
  }
  u32 qp[9];
  acc = 0; ovf = 0;
#pragma unroll
  for (int k = 0; k < 9; ++k) {
#pragma unroll
    for (int i = (k > 7 ? k - 7 : 0); i <= k; ++i) mac_s(acc, ovf, q3[i], p[k - i]);
    qp[k] = (u32)acc;
    acc = (acc >> 32) | ((u64)ovf << 32);
    ovf = 0;
  }
  u32 r[8];
  int64_t c = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) { c += (int64_t)T[i] - qp[i]; r[i] = (u32)c; c >>= 32; }
  cond_sub_const(r, BN_FR_2R);
  cond_sub_const(r, BN_FR_R);
  return fp_from_limbs(r[0], r[1], r[2], r[3], r[4], r[5], r[6], r[7]);
}
BN_NOINLINE Fp fr_mul(Fp a, Fp b) { return fr_mulmod_inline(a, b); }

// a^(r-2): the value crypto-bigint's inversion yields for a != 0, and 0 for a = 0 (fp.rs:418-433 through the macro)
BN_DEV Fp fr_inv(const Fp& a) {
  const u32 e[8] = {0xefffffffu, 0x43e1f593u, 0x79b97091u, 0x2833e848u, 0x8181585du, 0xb85045b6u, 0xe131a029u, 0x30644e72u};
  Fp res = fp_from_limbs(1, 0, 0, 0, 0, 0, 0, 0);
#pragma unroll
  for (int w = 7; w >= 0; --w) {
    const u32 ew = e[w];
#pragma unroll 1
    for (int b = 31; b >= 0; --b) {
      res = fr_mul(res, res);
      if ((ew >> b) & 1) res = fr_mul(res, a);   // exponent is a constant: wave-uniform
    }
  }
  return res;
}

}  // namespace bn254
