// common.hpp -- device-side helpers shared by every translation unit of libsylow_hip.so: struct-of-arrays load / store,
// launch geometry, byte codecs for 32-byte big-endian field elements.  gfx950 only.
#pragma once
#include "../../include/sylow_hip.h"

#include <hip/hip_runtime.h>

#include "bn254_hash.hpp"
#include "bn254_fr.hpp"

using namespace bn254;

// ------------------------------------------------------------------ SoA load / store ----------
// word w of element i lives at base[w * n + i]: a wavefront reads 64 consecutive uint64 (512 B)
// per word -> fully coalesced, and the 4 words of an Fp are 4 independent loads in flight.
BN_DEV Fp load_plain(const u64* __restrict__ base, size_t n, size_t i, int w0) {
  Fp r;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    u64 w = base[(size_t)(w0 + k) * n + i];
    r.v[2 * k] = (u32)w;
    r.v[2 * k + 1] = (u32)(w >> 32);
  }
  return r;
}
BN_DEV void store_plain(u64* __restrict__ base, size_t n, size_t i, int w0, const Fp& a) {
#pragma unroll
  for (int k = 0; k < 4; ++k) base[(size_t)(w0 + k) * n + i] = (u64)a.v[2 * k] | ((u64)a.v[2 * k + 1] << 32);
}
BN_DEV Fp load_fp(const u64* base, size_t n, size_t i, int w0) { return fp_to_mont(load_plain(base, n, i, w0)); }
BN_DEV void store_fp(u64* base, size_t n, size_t i, int w0, const Fp& a) { store_plain(base, n, i, w0, fp_from_mont(a)); }
BN_DEV Fp2 load_fp2(const u64* base, size_t n, size_t i, int w0) { return Fp2{load_fp(base, n, i, w0), load_fp(base, n, i, w0 + 4)}; }
BN_DEV void store_fp2(u64* base, size_t n, size_t i, int w0, const Fp2& a) { store_fp(base, n, i, w0, a.c0); store_fp(base, n, i, w0 + 4, a.c1); }
BN_DEV void load_fp6(Fp6& r, const u64* base, size_t n, size_t i, int w0) {
  r.c0 = load_fp2(base, n, i, w0); r.c1 = load_fp2(base, n, i, w0 + 8); r.c2 = load_fp2(base, n, i, w0 + 16);
}
BN_DEV void store_fp6(u64* base, size_t n, size_t i, int w0, const Fp6& a) {
  store_fp2(base, n, i, w0, a.c0); store_fp2(base, n, i, w0 + 8, a.c1); store_fp2(base, n, i, w0 + 16, a.c2);
}
BN_DEV void load_fp12(Fp12& r, const u64* base, size_t n, size_t i) { load_fp6(r.c0, base, n, i, 0); load_fp6(r.c1, base, n, i, 24); }
BN_DEV void store_fp12(u64* base, size_t n, size_t i, const Fp12& a) { store_fp6(base, n, i, 0, a.c0); store_fp6(base, n, i, 24, a.c1); }

#define TID ((size_t)blockIdx.x * blockDim.x + threadIdx.x)
constexpr int BLOCK = 256;
// the heavy kernels keep an Fp12 working set per lane: ask for 2 waves per SIMD (<= 256 VGPRs),
// the occupancy at which v_mad_u64_u32 already reaches its peak issue rate (profiles/r01_issue_rate_ubench.txt)
#define HEAVY_BOUNDS __launch_bounds__(BLOCK, 2)

enum { OP_ADD = 0, OP_SUB = 1, OP_MUL = 2, OP_SQR = 3, OP_NEG = 4, OP_INV = 5 };

// ------------------------------------------------------------------ group kernels --------------
BN_DEV void load_scalar(u32 (&k)[8], const u64* base, size_t n, size_t i) {
  // scalars are Fp values: reduce like Fp::new so that k >= p behaves as in the reference
  Fp s = fp_from_mont(fp_to_mont(load_plain(base, n, i, 0)));
#pragma unroll
  for (int j = 0; j < 8; ++j) k[j] = s.v[j];
}

BN_DEV int wave_max(int v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) { int o = __shfl_xor(v, off); v = o > v ? o : v; }
  return __builtin_amdgcn_readfirstlane(v);
}

// ---- 32-byte big-endian field elements (Fp::from_be_bytes / to_be_bytes, fp.rs:686-737) ----------
// Two 16-byte loads whatever the alignment of b (the fixed-size copy compiles to global_load_dwordx4), then byte swaps.  clear_flag drops
// bit 7 of byte 0 (the identity flag of the point encodings) before the range check.
BN_DEV bool read_be_fp(Fp& out, const uint8_t* b, bool clear_flag = false) {         // returns false when the value is >= p
  u32 w[8];
  __builtin_memcpy(w, b, 32);
  Fp x;
#pragma unroll
  for (int j = 0; j < 8; ++j) x.v[j] = __builtin_bswap32(w[7 - j]);
  if (clear_flag) x.v[7] &= 0x7fffffffu;
  const u32 pl[8] = {BN_P0, BN_P1, BN_P2, BN_P3, BN_P4, BN_P5, BN_P6, BN_P7};
  bool lt = false, decided = false;
#pragma unroll
  for (int j = 7; j >= 0; --j) {
    if (!decided && x.v[j] != pl[j]) { lt = x.v[j] < pl[j]; decided = true; }
  }
  out = x;
  return lt;
}
BN_DEV void write_be_fp(uint8_t* b, const Fp& plain) {
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    u32 w = plain.v[j];
    uint8_t* q = b + 28 - 4 * j;
    q[0] = (uint8_t)(w >> 24); q[1] = (uint8_t)(w >> 16); q[2] = (uint8_t)(w >> 8); q[3] = (uint8_t)w;
  }
}

// ---- EVM alt_bn128 point / scalar codecs (examples/reth_bn128.rs:99-217), shared by g1.hip and sign_wide.hip -----------------------
// read_point + new_g1_point (reth_bn128.rs:107-128): Montgomery-form affine point or identity
BN_DEV uint8_t evm_read_g1(G1P& out, const uint8_t* b) {
  Fp x, y;
  bool okx = read_be_fp(x, b), oky = read_be_fp(y, b + 32);
  if (!(okx && oky)) { out = proj_zero<OpsFp>(); return SYLOW_HIP_ST_DECODE_ERROR; }
  if (fp_is_zero(x) && fp_is_zero(y)) { out = proj_zero<OpsFp>(); return SYLOW_HIP_ST_OK; }
  Fp xm = fp_to_mont(x), ym = fp_to_mont(y);
  out = G1P{xm, ym, fp_one()};
  return g1_on_curve_affine(xm, ym) ? SYLOW_HIP_ST_OK : SYLOW_HIP_ST_NOT_ON_CURVE;
}
// to_be_bytes_scrubbed (g1.rs:182-192): all-zero bytes for the identity
BN_DEV void evm_write_g1(uint8_t* b, const G1P& p) {
  Fp x, y; bool inf;
  g1_to_affine(x, y, inf, p);
  Fp zero = fp_zero();
  write_be_fp(b, inf ? zero : fp_from_mont(x));
  write_be_fp(b + 32, inf ? zero : fp_from_mont(y));
}
// EIP-196 accepts any 256-bit scalar; G1 has prime order r, so reduce mod r (2^256 < 6r).  (The reference adapter unwraps
// Fr::from_be_bytes and would panic for k >= r, reth_bn128.rs:144.)
BN_DEV void evm_read_scalar(u32 (&k)[8], const uint8_t* b) {
  Fp kx;
  read_be_fp(kx, b);
#pragma unroll
  for (int w = 0; w < 8; ++w) k[w] = kx.v[w];
  cond_sub_const(k, 0xc0000004u, 0x0f87d64fu, 0xe6e5c245u, 0xa0cfa121u, 0x06056174u, 0xe14116dau, 0x84c680a6u, 0xc19139cbu);  // 4r
  cond_sub_const(k, 0xe0000002u, 0x87c3eb27u, 0xf372e122u, 0x5067d090u, 0x0302b0bau, 0x70a08b6du, 0xc2634053u, 0x60c89ce5u);  // 2r
  cond_sub_const(k, 0xf0000001u, 0x43e1f593u, 0x79b97091u, 0x2833e848u, 0x8181585du, 0xb85045b6u, 0xe131a029u, 0x30644e72u);  // r
}

