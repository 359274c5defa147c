// BN254 base field Fp on gfx950: 8 x 32-bit limbs per lane, Montgomery form (R = 2^256),
// canonical representatives in [0, p).  One field element per lane ("one-op-per-lane").
//
// Replaces, for batches, sylow's `Fp` (src/fields/fp.rs:174-566), i.e. crypto-bigint's
// ConstMontyForm<U256> (fp.rs:179-190): add/sub/neg (fp.rs:304-347,442-449), mul/square
// (fp.rs:387-393,620-622), inv with inv(0)=0 (fp.rs:418-433), pow/sqrt/is_square/sgn0
// (fp.rs:451-457,611-644).  Results are exact residues mod p, so any correct algorithm is
// bit-exact with the reference.
//
// Why 32-bit limbs + v_mad_u64_u32: measured on MI355X (profiles/r01_issue_rate_ubench.txt)
// v_mad_u64_u32 issues at ~520 G wave-instr/s chip-wide, ~0.9x the rate of any other VOP3 op,
// so the cost of an Fp operation is its total instruction count.  The multiplier below is a
// finely-integrated product-scanning (FIPS/Comba) Montgomery multiplication: per partial product
// one v_mad_u64_u32 into a 64-bit column accumulator plus one v_addc_co_u32 collecting the
// carry-out into a third word (the mad has a carry-out but no carry-in).  128 mad + 128 addc +
// 8 v_mul_lo + ~30 column shifts + 24 for the final conditional subtraction ~= 320 instructions,
// 1.4x fewer than what hipcc emits for the same arithmetic written in plain C (measured
// 121 vs 87 G Fp-mul/s).  No MFMA: this is an integer carry chain, not a contraction.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace bn254 {

typedef uint32_t u32;
typedef uint64_t u64;

#define BN_DEV __device__ __forceinline__
#define BN_NOINLINE __device__ __noinline__

struct Fp {
  u32 v[8];
};

// p, little-endian 32-bit limbs (fp.rs:51-56)
#define BN_P0 0xd87cfd47u
#define BN_P1 0x3c208c16u
#define BN_P2 0x6871ca8du
#define BN_P3 0x97816a91u
#define BN_P4 0x8181585du
#define BN_P5 0xb85045b6u
#define BN_P6 0xe131a029u
#define BN_P7 0x30644e72u
#define BN_PINV32 0xe4866389u  // -p^-1 mod 2^32

BN_DEV Fp fp_from_limbs(u32 a0, u32 a1, u32 a2, u32 a3, u32 a4, u32 a5, u32 a6, u32 a7) {
  Fp r;
  r.v[0] = a0; r.v[1] = a1; r.v[2] = a2; r.v[3] = a3; r.v[4] = a4; r.v[5] = a5; r.v[6] = a6; r.v[7] = a7;
  return r;
}
BN_DEV Fp fp_zero() { return fp_from_limbs(0, 0, 0, 0, 0, 0, 0, 0); }
// R mod p  (Montgomery one)
BN_DEV Fp fp_one() {
  return fp_from_limbs(0xc58f0d9du, 0xd35d438du, 0xf5c70b3du, 0x0a78eb28u, 0x7879462cu, 0x666ea36fu, 0x9a07df2fu, 0x0e0a77c1u);
}
// R^2 mod p
BN_DEV Fp fp_r2() {
  return fp_from_limbs(0x538afa89u, 0xf32cfc5bu, 0xd44501fbu, 0xb5e71911u, 0x0a417ff6u, 0x47ab1effu, 0xcab8351fu, 0x06d89f71u);
}

// ---- acc(64) += x*y, ovf(32) += carry ------------------------------------------------------
BN_DEV void mac(u64& acc, u32& ovf, u32 x, u32 y) {
  asm("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc"
      : "+v"(acc), "+v"(ovf)
      : "v"(x), "v"(y)
      : "vcc");
}
// same, second factor wave-uniform (an SGPR or inline constant): used for the modulus limbs
BN_DEV void mac_s(u64& acc, u32& ovf, u32 x, u32 y_uniform) {
  asm("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc"
      : "+v"(acc), "+v"(ovf)
      : "v"(x), "s"(y_uniform)
      : "vcc");
}

// r = (t >= p) ? t - p : t, where t = top:r[0..7] < 2p
BN_DEV Fp fp_cond_sub_p(const u32 r[8], u32 top) {
  u32 s[8];
  u32 bor;
  const u32 p0 = BN_P0, p1 = BN_P1, p2 = BN_P2, p3 = BN_P3, p4 = BN_P4, p5 = BN_P5, p6 = BN_P6, p7 = BN_P7;
  asm("v_sub_co_u32 %0, vcc, %9, %17\n\t"
      "v_subb_co_u32 %1, vcc, %10, %18, vcc\n\t"
      "v_subb_co_u32 %2, vcc, %11, %19, vcc\n\t"
      "v_subb_co_u32 %3, vcc, %12, %20, vcc\n\t"
      "v_subb_co_u32 %4, vcc, %13, %21, vcc\n\t"
      "v_subb_co_u32 %5, vcc, %14, %22, vcc\n\t"
      "v_subb_co_u32 %6, vcc, %15, %23, vcc\n\t"
      "v_subb_co_u32 %7, vcc, %16, %24, vcc\n\t"
      "v_subb_co_u32 %8, vcc, %25, 0, vcc"
      : "=&v"(s[0]), "=&v"(s[1]), "=&v"(s[2]), "=&v"(s[3]), "=&v"(s[4]), "=&v"(s[5]), "=&v"(s[6]), "=&v"(s[7]),
        "=&v"(bor)
      : "v"(r[0]), "v"(r[1]), "v"(r[2]), "v"(r[3]), "v"(r[4]), "v"(r[5]), "v"(r[6]), "v"(r[7]), "v"(p0), "v"(p1),
        "v"(p2), "v"(p3), "v"(p4), "v"(p5), "v"(p6), "v"(p7), "v"(top)
      : "vcc");
  Fp out;
#pragma unroll
  for (int i = 0; i < 8; ++i) out.v[i] = (bor != 0) ? r[i] : s[i];
  return out;
}

// t >= c ? t - c : t for a 256-bit constant c given as limbs (c = p, 2p or 4p)
BN_DEV void cond_sub_const(u32 (&r)[8], u32 c0, u32 c1, u32 c2, u32 c3, u32 c4, u32 c5, u32 c6, u32 c7) {
  u32 s[8];
  u32 bor;
  asm("v_sub_co_u32 %0, vcc, %9, %17\n\t"
      "v_subb_co_u32 %1, vcc, %10, %18, vcc\n\t"
      "v_subb_co_u32 %2, vcc, %11, %19, vcc\n\t"
      "v_subb_co_u32 %3, vcc, %12, %20, vcc\n\t"
      "v_subb_co_u32 %4, vcc, %13, %21, vcc\n\t"
      "v_subb_co_u32 %5, vcc, %14, %22, vcc\n\t"
      "v_subb_co_u32 %6, vcc, %15, %23, vcc\n\t"
      "v_subb_co_u32 %7, vcc, %16, %24, vcc\n\t"
      "v_cndmask_b32 %8, 0, -1, vcc"
      : "=&v"(s[0]), "=&v"(s[1]), "=&v"(s[2]), "=&v"(s[3]), "=&v"(s[4]), "=&v"(s[5]), "=&v"(s[6]), "=&v"(s[7]),
        "=&v"(bor)
      : "v"(r[0]), "v"(r[1]), "v"(r[2]), "v"(r[3]), "v"(r[4]), "v"(r[5]), "v"(r[6]), "v"(r[7]), "v"(c0), "v"(c1),
        "v"(c2), "v"(c3), "v"(c4), "v"(c5), "v"(c6), "v"(c7)
      : "vcc");
#pragma unroll
  for (int i = 0; i < 8; ++i) r[i] = (bor != 0) ? r[i] : s[i];
}
// any 256-bit value -> canonical residue, like Fp::new (fp.rs:199-201) but without a Montgomery round
// trip: 2^256 < 6p, so conditional subtractions of 4p, 2p, p suffice (4p = 0xc19139cb... < 2^256)
BN_DEV Fp fp_reduce_plain(const Fp& x) {
  u32 r[8] = {x.v[0], x.v[1], x.v[2], x.v[3], x.v[4], x.v[5], x.v[6], x.v[7]};
  cond_sub_const(r, 0x61f3f51cu, 0xf082305bu, 0xa1c72a34u, 0x5e05aa45u, 0x06056176u, 0xe14116dau, 0x84c680a6u, 0xc19139cbu);  // 4p
  cond_sub_const(r, 0xb0f9fa8eu, 0x7841182du, 0xd0e3951au, 0x2f02d522u, 0x0302b0bbu, 0x70a08b6du, 0xc2634053u, 0x60c89ce5u);  // 2p
  cond_sub_const(r, BN_P0, BN_P1, BN_P2, BN_P3, BN_P4, BN_P5, BN_P6, BN_P7);
  return fp_from_limbs(r[0], r[1], r[2], r[3], r[4], r[5], r[6], r[7]);
}

// Montgomery product a*b/R mod p, inputs and output canonical (fp.rs:387-393).
// Out-of-line leaf: 8 + 8 argument registers is exactly what the AMDGPU C ABI passes in VGPRs
// (16), the result comes back in 8, and the body touches ~50 VGPRs, so callers keep ~200
// registers of live tower state across the call without spilling.
BN_DEV Fp fp_mul_inline(const Fp& a, const Fp& b) {
  const u32 p[8] = {BN_P0, BN_P1, BN_P2, BN_P3, BN_P4, BN_P5, BN_P6, BN_P7};
  u32 m[8];
  u32 r[8];
  u64 acc = 0;
  u32 ovf = 0;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
#pragma unroll
    for (int i = 0; i <= k; ++i) mac(acc, ovf, a.v[i], b.v[k - i]);
#pragma unroll
    for (int i = 0; i < k; ++i) mac_s(acc, ovf, m[i], p[k - i]);
    m[k] = (u32)acc * BN_PINV32;
    mac_s(acc, ovf, m[k], p[0]);
    acc = (acc >> 32) | ((u64)ovf << 32);
    ovf = 0;
  }
#pragma unroll
  for (int k = 8; k < 16; ++k) {
#pragma unroll
    for (int i = k - 7; i < 8; ++i) mac(acc, ovf, a.v[i], b.v[k - i]);
#pragma unroll
    for (int i = k - 7; i < 8; ++i) mac_s(acc, ovf, m[i], p[k - i]);
    r[k - 8] = (u32)acc;
    acc = (acc >> 32) | ((u64)ovf << 32);
    ovf = 0;
  }
  return fp_cond_sub_p(r, (u32)acc);
}

BN_NOINLINE Fp fp_mul(Fp a, Fp b) { return fp_mul_inline(a, b); }

// a*b mod p for plain (non-Montgomery) operands -- any 256-bit values -- by Barrett reduction (HAC 14.42, b = 2^32, k = 8,
// mu = floor(2^512 / p), 9 limbs): T = a*b (64 products), q3 = top limbs of (T >> 224) * mu with the partial
// products below column 7 dropped (53 products; the dropped mass is < 28 b^8, i.e. at most one unit of q3),
// r = (T - q3*p) mod 2^288 (43 products) < 4p, then conditional subtraction of 2p and p.  160 multiply-adds
// instead of the 256 of two Montgomery products: what the HBM-bound canonical-domain batch multiply uses.
BN_DEV Fp fp_mulmod_plain(const Fp& a, const Fp& b) {
  const u32 p[8] = {BN_P0, BN_P1, BN_P2, BN_P3, BN_P4, BN_P5, BN_P6, BN_P7};
  const u32 mu[9] = {0x9bf90e51u, 0xf3aed8a1u, 0x7cd4c086u, 0xe965e176u, 0x8073013au, 0xb074a586u, 0x23a04a7au, 0x4a474626u, 0x00000005u};
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
  // q1 = T[7..15] (9 limbs); q2 columns 7..17 of q1 * mu; q3 = columns 9..17 (9 limbs: operands may be any
  // 256-bit values, like Fp::new accepts, so q can reach 2^259)
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
  // low 9 limbs of q3 * p
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
  // r = T[0..8] - qp[0..8] mod 2^288 (the true remainder is < 4p < 2^256, so limb 8 ends up 0)
  u32 r[8];
  u32 r8;
  asm("v_sub_co_u32 %0, vcc, %9, %18\n\t"
      "v_subb_co_u32 %1, vcc, %10, %19, vcc\n\t"
      "v_subb_co_u32 %2, vcc, %11, %20, vcc\n\t"
      "v_subb_co_u32 %3, vcc, %12, %21, vcc\n\t"
      "v_subb_co_u32 %4, vcc, %13, %22, vcc\n\t"
      "v_subb_co_u32 %5, vcc, %14, %23, vcc\n\t"
      "v_subb_co_u32 %6, vcc, %15, %24, vcc\n\t"
      "v_subb_co_u32 %7, vcc, %16, %25, vcc\n\t"
      "v_subb_co_u32 %8, vcc, %17, %26, vcc"
      : "=&v"(r[0]), "=&v"(r[1]), "=&v"(r[2]), "=&v"(r[3]), "=&v"(r[4]), "=&v"(r[5]), "=&v"(r[6]), "=&v"(r[7]), "=&v"(r8)
      : "v"(T[0]), "v"(T[1]), "v"(T[2]), "v"(T[3]), "v"(T[4]), "v"(T[5]), "v"(T[6]), "v"(T[7]), "v"(T[8]),
        "v"(qp[0]), "v"(qp[1]), "v"(qp[2]), "v"(qp[3]), "v"(qp[4]), "v"(qp[5]), "v"(qp[6]), "v"(qp[7]), "v"(qp[8])
      : "vcc");
  (void)r8;
  cond_sub_const(r, 0xb0f9fa8eu, 0x7841182du, 0xd0e3951au, 0x2f02d522u, 0x0302b0bbu, 0x70a08b6du, 0xc2634053u, 0x60c89ce5u);  // 2p
  cond_sub_const(r, BN_P0, BN_P1, BN_P2, BN_P3, BN_P4, BN_P5, BN_P6, BN_P7);
  return fp_from_limbs(r[0], r[1], r[2], r[3], r[4], r[5], r[6], r[7]);
}

// (a*b + c*d) / R mod p in ONE column pass and ONE Montgomery reduction (lazy reduction at the Fp2 level):
// for inputs <= p the sum is <= 2p^2 < p*R, so the reduced value is < 2p^2/R + p < 1.4p and a single
// conditional subtraction is enough.  The 3-word column accumulator absorbs the extra products for free.
BN_DEV Fp fp_dot2_inline(const Fp& a, const Fp& b, const Fp& c, const Fp& d) {
  const u32 p[8] = {BN_P0, BN_P1, BN_P2, BN_P3, BN_P4, BN_P5, BN_P6, BN_P7};
  u32 m[8];
  u32 r[8];
  u64 acc = 0;
  u32 ovf = 0;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
#pragma unroll
    for (int i = 0; i <= k; ++i) { mac(acc, ovf, a.v[i], b.v[k - i]); mac(acc, ovf, c.v[i], d.v[k - i]); }
#pragma unroll
    for (int i = 0; i < k; ++i) mac_s(acc, ovf, m[i], p[k - i]);
    m[k] = (u32)acc * BN_PINV32;
    mac_s(acc, ovf, m[k], p[0]);
    acc = (acc >> 32) | ((u64)ovf << 32);
    ovf = 0;
  }
#pragma unroll
  for (int k = 8; k < 16; ++k) {
#pragma unroll
    for (int i = k - 7; i < 8; ++i) { mac(acc, ovf, a.v[i], b.v[k - i]); mac(acc, ovf, c.v[i], d.v[k - i]); }
#pragma unroll
    for (int i = k - 7; i < 8; ++i) mac_s(acc, ovf, m[i], p[k - i]);
    r[k - 8] = (u32)acc;
    acc = (acc >> 32) | ((u64)ovf << 32);
    ovf = 0;
  }
  return fp_cond_sub_p(r, (u32)acc);
}
// p - a for canonical a (in (0, p]; p itself stands for 0 and is fine as a fp_dot2 operand)
BN_DEV Fp fp_neg_lazy(const Fp& a) {
  Fp r;
  const u32 p0 = BN_P0, p1 = BN_P1, p2 = BN_P2, p3 = BN_P3, p4 = BN_P4, p5 = BN_P5, p6 = BN_P6, p7 = BN_P7;
  asm("v_sub_co_u32 %0, vcc, %8, %16\n\t"
      "v_subb_co_u32 %1, vcc, %9, %17, vcc\n\t"
      "v_subb_co_u32 %2, vcc, %10, %18, vcc\n\t"
      "v_subb_co_u32 %3, vcc, %11, %19, vcc\n\t"
      "v_subb_co_u32 %4, vcc, %12, %20, vcc\n\t"
      "v_subb_co_u32 %5, vcc, %13, %21, vcc\n\t"
      "v_subb_co_u32 %6, vcc, %14, %22, vcc\n\t"
      "v_subb_co_u32 %7, vcc, %15, %23, vcc"
      : "=&v"(r.v[0]), "=&v"(r.v[1]), "=&v"(r.v[2]), "=&v"(r.v[3]), "=&v"(r.v[4]), "=&v"(r.v[5]), "=&v"(r.v[6]), "=&v"(r.v[7])
      : "v"(p0), "v"(p1), "v"(p2), "v"(p3), "v"(p4), "v"(p5), "v"(p6), "v"(p7),
        "v"(a.v[0]), "v"(a.v[1]), "v"(a.v[2]), "v"(a.v[3]), "v"(a.v[4]), "v"(a.v[5]), "v"(a.v[6]), "v"(a.v[7])
      : "vcc");
  return r;
}
BN_DEV Fp fp_sqr(const Fp& a) { return fp_mul(a, a); }  // fp.rs:620-622

// (a + b) mod p  (fp.rs:304-310)
BN_DEV Fp fp_add(const Fp& a, const Fp& b) {
  u32 s[8];
  asm("v_add_co_u32 %0, vcc, %8, %16\n\t"
      "v_addc_co_u32 %1, vcc, %9, %17, vcc\n\t"
      "v_addc_co_u32 %2, vcc, %10, %18, vcc\n\t"
      "v_addc_co_u32 %3, vcc, %11, %19, vcc\n\t"
      "v_addc_co_u32 %4, vcc, %12, %20, vcc\n\t"
      "v_addc_co_u32 %5, vcc, %13, %21, vcc\n\t"
      "v_addc_co_u32 %6, vcc, %14, %22, vcc\n\t"
      "v_addc_co_u32 %7, vcc, %15, %23, vcc"
      : "=&v"(s[0]), "=&v"(s[1]), "=&v"(s[2]), "=&v"(s[3]), "=&v"(s[4]), "=&v"(s[5]), "=&v"(s[6]), "=&v"(s[7])
      : "v"(a.v[0]), "v"(a.v[1]), "v"(a.v[2]), "v"(a.v[3]), "v"(a.v[4]), "v"(a.v[5]), "v"(a.v[6]), "v"(a.v[7]),
        "v"(b.v[0]), "v"(b.v[1]), "v"(b.v[2]), "v"(b.v[3]), "v"(b.v[4]), "v"(b.v[5]), "v"(b.v[6]), "v"(b.v[7])
      : "vcc");
  return fp_cond_sub_p(s, 0);  // a + b < 2p < 2^255: no carry out of the top limb
}

// a + b WITHOUT the conditional subtraction: for canonical a, b the sum is < 2p < 2^255.  Only used as a
// direct operand of fp_mul: the Montgomery product of x < 2p and y < 2p is < 4p^2/R + p < 1.76p, which the
// multiplier's own conditional subtraction brings into [0, p) -- same residue, 16 instructions saved.
BN_DEV Fp fp_add_lazy(const Fp& a, const Fp& b) {
  Fp s;
  asm("v_add_co_u32 %0, vcc, %8, %16\n\t"
      "v_addc_co_u32 %1, vcc, %9, %17, vcc\n\t"
      "v_addc_co_u32 %2, vcc, %10, %18, vcc\n\t"
      "v_addc_co_u32 %3, vcc, %11, %19, vcc\n\t"
      "v_addc_co_u32 %4, vcc, %12, %20, vcc\n\t"
      "v_addc_co_u32 %5, vcc, %13, %21, vcc\n\t"
      "v_addc_co_u32 %6, vcc, %14, %22, vcc\n\t"
      "v_addc_co_u32 %7, vcc, %15, %23, vcc"
      : "=&v"(s.v[0]), "=&v"(s.v[1]), "=&v"(s.v[2]), "=&v"(s.v[3]), "=&v"(s.v[4]), "=&v"(s.v[5]), "=&v"(s.v[6]), "=&v"(s.v[7])
      : "v"(a.v[0]), "v"(a.v[1]), "v"(a.v[2]), "v"(a.v[3]), "v"(a.v[4]), "v"(a.v[5]), "v"(a.v[6]), "v"(a.v[7]),
        "v"(b.v[0]), "v"(b.v[1]), "v"(b.v[2]), "v"(b.v[3]), "v"(b.v[4]), "v"(b.v[5]), "v"(b.v[6]), "v"(b.v[7])
      : "vcc");
  return s;
}

// (9a + s*b) mod p for s = +-1: the two coordinates of x*(9+u) (fp2.rs:99-107) as ONE multiply-by-9 pass instead of
// five modular add/subs.  t = 9a + (s > 0 ? b : p - b) is a 9-word value < 10p; its quotient by p is estimated
// from the top bits (h = t >> 250 <= 120, q^ = 677 h >> 13 is q or q-1 -- checked exhaustively over every h
// boundary and every multiple of p in tools/check_mulxi_quotient.py), so r = t - q^ p < 2p and one conditional
// subtraction finishes.  ~70 instructions instead of 120.
template <bool ADD>
BN_DEV Fp fp_mul9_addsub(const Fp& a, const Fp& b) {
  const u32 p[8] = {BN_P0, BN_P1, BN_P2, BN_P3, BN_P4, BN_P5, BN_P6, BN_P7};
  // w = ADD ? b : p - b   (in (0, p])
  Fp w = ADD ? b : fp_neg_lazy(b);
  // t = 9a + w, 9 words
  u32 t[9];
  u64 acc = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    acc += (u64)a.v[i] * 9u + w.v[i];      // < 9*2^32 + 2^32 + carry: no overflow
    t[i] = (u32)acc;
    acc >>= 32;
  }
  t[8] = (u32)acc;                          // < 10
  // q^ from the top bits: h = t >> 250 = (t[8] << 6) | (t[7] >> 26)
  u32 h = (t[8] << 6) | (t[7] >> 26);
  u32 q = (h * 677u) >> 13;
  // r = t - q*p (fits 8 words: r < 2p)
  u32 r[8];
  u64 bor = 0;                              // running (q*p) carry and borrow folded: compute qp limb by limb
  u64 mp = 0;
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

// (a - b) mod p  (fp.rs:340-347)
BN_DEV Fp fp_sub(const Fp& a, const Fp& b) {
  u32 d[8];
  u32 mask;
  asm("v_sub_co_u32 %0, vcc, %9, %17\n\t"
      "v_subb_co_u32 %1, vcc, %10, %18, vcc\n\t"
      "v_subb_co_u32 %2, vcc, %11, %19, vcc\n\t"
      "v_subb_co_u32 %3, vcc, %12, %20, vcc\n\t"
      "v_subb_co_u32 %4, vcc, %13, %21, vcc\n\t"
      "v_subb_co_u32 %5, vcc, %14, %22, vcc\n\t"
      "v_subb_co_u32 %6, vcc, %15, %23, vcc\n\t"
      "v_subb_co_u32 %7, vcc, %16, %24, vcc\n\t"
      "v_cndmask_b32 %8, 0, -1, vcc"
      : "=&v"(d[0]), "=&v"(d[1]), "=&v"(d[2]), "=&v"(d[3]), "=&v"(d[4]), "=&v"(d[5]), "=&v"(d[6]), "=&v"(d[7]),
        "=&v"(mask)
      : "v"(a.v[0]), "v"(a.v[1]), "v"(a.v[2]), "v"(a.v[3]), "v"(a.v[4]), "v"(a.v[5]), "v"(a.v[6]), "v"(a.v[7]),
        "v"(b.v[0]), "v"(b.v[1]), "v"(b.v[2]), "v"(b.v[3]), "v"(b.v[4]), "v"(b.v[5]), "v"(b.v[6]), "v"(b.v[7])
      : "vcc");
  // add (p & mask)
  u32 q[8] = {BN_P0 & mask, BN_P1 & mask, BN_P2 & mask, BN_P3 & mask, BN_P4 & mask, BN_P5 & mask, BN_P6 & mask, BN_P7 & mask};
  Fp out;
  asm("v_add_co_u32 %0, vcc, %8, %16\n\t"
      "v_addc_co_u32 %1, vcc, %9, %17, vcc\n\t"
      "v_addc_co_u32 %2, vcc, %10, %18, vcc\n\t"
      "v_addc_co_u32 %3, vcc, %11, %19, vcc\n\t"
      "v_addc_co_u32 %4, vcc, %12, %20, vcc\n\t"
      "v_addc_co_u32 %5, vcc, %13, %21, vcc\n\t"
      "v_addc_co_u32 %6, vcc, %14, %22, vcc\n\t"
      "v_addc_co_u32 %7, vcc, %15, %23, vcc"
      : "=&v"(out.v[0]), "=&v"(out.v[1]), "=&v"(out.v[2]), "=&v"(out.v[3]), "=&v"(out.v[4]), "=&v"(out.v[5]),
        "=&v"(out.v[6]), "=&v"(out.v[7])
      : "v"(d[0]), "v"(d[1]), "v"(d[2]), "v"(d[3]), "v"(d[4]), "v"(d[5]), "v"(d[6]), "v"(d[7]), "v"(q[0]), "v"(q[1]),
        "v"(q[2]), "v"(q[3]), "v"(q[4]), "v"(q[5]), "v"(q[6]), "v"(q[7])
      : "vcc");
  return out;
}

BN_DEV Fp fp_neg(const Fp& a) { return fp_sub(fp_zero(), a); }  // fp.rs:442-449
BN_DEV Fp fp_dbl(const Fp& a) { return fp_add(a, a); }

BN_DEV bool fp_is_zero(const Fp& a) {
  return (a.v[0] | a.v[1] | a.v[2] | a.v[3] | a.v[4] | a.v[5] | a.v[6] | a.v[7]) == 0;
}
BN_DEV bool fp_eq(const Fp& a, const Fp& b) {
  u32 d = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) d |= a.v[i] ^ b.v[i];
  return d == 0;
}
// c ? b : a   (subtle::ConditionallySelectable, fp.rs:379-383)
BN_DEV Fp fp_select(const Fp& a, const Fp& b, bool c) {
  Fp r;
#pragma unroll
  for (int i = 0; i < 8; ++i) r.v[i] = c ? b.v[i] : a.v[i];
  return r;
}

// canonical integer -> Montgomery form.  Accepts ANY 256-bit value, like Fp::new (fp.rs:199-201):
// the Montgomery product with R^2 of x < 2^256 is < (2^256*p + p*2^256)/2^256 = 2p, and the
// final conditional subtraction brings it into [0,p).
BN_DEV Fp fp_to_mont(const Fp& plain) { return fp_mul(plain, fp_r2()); }
// Montgomery form -> canonical integer (Fp::value, fp.rs:232-234)
BN_DEV Fp fp_from_mont(const Fp& a) { return fp_mul(a, fp_from_limbs(1, 0, 0, 0, 0, 0, 0, 0)); }

// small constants in Montgomery form, built by additions from one (no tables)
BN_DEV Fp fp_small(int k) {
  Fp one = fp_one();
  Fp r = fp_zero();
  Fp pw = one;
#pragma unroll 1
  for (; k; k >>= 1) {
    if (k & 1) r = fp_add(r, pw);
    pw = fp_dbl(pw);
  }
  return r;
}

// a^e for a fixed 256-bit exponent given as 8 wave-uniform u32 words (inversion / sqrt / Legendre paths).
// Defined in bn254_f29.hpp: the chain runs on the carry-free core (dedicated squaring, 4-bit windows).
BN_NOINLINE Fp fp_pow_words(Fp a, u32 e0, u32 e1, u32 e2, u32 e3, u32 e4, u32 e5, u32 e6, u32 e7);
// a^-1, inv(0) = 0 like the reference (fp.rs:418-433, test fp.rs:1126-1132).  The reference inverts with crypto-bigint's
// Bernstein-Yang safegcd; so does this (bn254_f29.hpp: 600 branch-free divsteps on 30-bit limbs -- every lane runs the same
// instruction stream whatever its input, which is exactly what a wavefront wants) instead of the Fermat power a^(p-2).
BN_NOINLINE Fp fp_inv_safegcd(Fp a);
BN_DEV Fp fp_inv(const Fp& a) { return fp_inv_safegcd(a); }
// the Fermat twin, kept for the tests
BN_DEV Fp fp_inv_fermat(const Fp& a) {
  return fp_pow_words(a, BN_P0 - 2, BN_P1, BN_P2, BN_P3, BN_P4, BN_P5, BN_P6, BN_P7);
}

// a^((p-3)/4): one chain serves both the Legendre symbol (a * t^2 = a^((p-1)/2)) and the square-root candidate
// (a * t = a^((p+1)/4)) of the same element
BN_NOINLINE Fp fp_pow_pm3_quarter_chain(Fp a);       // bn254_f29.hpp: the precomputed sliding-window schedule of this exponent
BN_DEV Fp fp_pow_pm3_quarter(const Fp& a) { return fp_pow_pm3_quarter_chain(a); }

}  // namespace bn254
