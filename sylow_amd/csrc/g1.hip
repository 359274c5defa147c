// g1.hip -- one-element-per-lane G1 kernels: group law, scalar multiplication (GLV, carry-free core), weighted aggregation,
// hash-to-G1 (XMD-Keccak256 + SvdW), BLS signing, the G1 wire format and the EIP-196 ecAdd / ecMul byte adapters.
#include "host.hpp"

// An affine SoA point as projective coordinates.  A FLAGGED point is the identity whatever its coordinate words hold: it is loaded as the
// canonical (0 : 1 : 0) -- (x : y : 0) with x != 0 is not a point of the curve and the complete formulas owe it nothing
BN_DEV G1P load_g1_flagged(const u64* xy, const uint8_t* inf, size_t n, size_t i) {
  const bool z = inf && inf[i];
  return G1P{z ? fp_zero() : load_fp(xy, n, i, 0), z ? fp_one() : load_fp(xy, n, i, 4), z ? fp_zero() : fp_one()};
}
// tables: NULL = window tables in the stack frame, else a block of n * G1_TABLE_BYTES_PER_LANE bytes (lane i's table contiguous)
__global__ void HEAVY_BOUNDS k_g1_scalar_mul(const u64* pxy, const uint8_t* pinf, const u64* ks, u64* oxy, uint8_t* oinf, size_t n, uint8_t* tables) {
  size_t i = TID;
  if (i >= n) return;
  bool inf = pinf && pinf[i];
  G1P p{load_fp(pxy, n, i, 0), load_fp(pxy, n, i, 4), inf ? fp_zero() : fp_one()};
  u32 k[8];
  load_scalar(k, ks, n, i);
  G1P r = tables ? g1_scalar_mul_ws(p, k, tables + i * G1_TABLE_BYTES_PER_LANE) : g1_scalar_mul(p, k);
  Fp x, y; bool rinf;
  g1_to_affine(x, y, rinf, r);
  store_fp(oxy, n, i, 0, x); store_fp(oxy, n, i, 4, y);
  oinf[i] = rinf ? 1 : 0;
}
__global__ void __launch_bounds__(BLOCK) k_g1_add(const u64* axy, const uint8_t* ainf, const u64* bxy, const uint8_t* binf, u64* oxy, uint8_t* oinf, size_t n) {
  size_t i = TID;
  if (i >= n) return;
  G1P a = load_g1_flagged(axy, ainf, n, i), b = load_g1_flagged(bxy, binf, n, i);
  G1P r = g1_add(a, b);
  Fp x, y; bool rinf;
  g1_to_affine(x, y, rinf, r);
  store_fp(oxy, n, i, 0, x); store_fp(oxy, n, i, 4, y);
  oinf[i] = rinf ? 1 : 0;
}
// Sub for projective points (group.rs:614-624): self + (-other), affine SoA in / out like k_g1_add
__global__ void __launch_bounds__(BLOCK) k_g1_sub(const u64* axy, const uint8_t* ainf, const u64* bxy, const uint8_t* binf, u64* oxy, uint8_t* oinf, size_t n) {
  size_t i = TID;
  if (i >= n) return;
  G1P a = load_g1_flagged(axy, ainf, n, i), b = load_g1_flagged(bxy, binf, n, i);
  b.y = fp_neg(b.y);
  G1P r = g1_add(a, b);
  Fp x, y; bool rinf;
  g1_to_affine(x, y, rinf, r);
  store_fp(oxy, n, i, 0, x); store_fp(oxy, n, i, 4, y);
  oinf[i] = rinf ? 1 : 0;
}
// G1Projective::new([x, y, z]) (g1.rs:383-402): Y^2 Z == X^3 + 3 Z^3, or Z == 0
__global__ void __launch_bounds__(BLOCK) k_g1_projective_new(const u64* pxyz, uint8_t* status, size_t n) {
  size_t i = TID;
  if (i >= n) return;
  const Fp x = load_fp(pxyz, n, i, 0), y = load_fp(pxyz, n, i, 4), z = load_fp(pxyz, n, i, 8);
  const Fp lhs = fp_mul(fp_sqr(y), z);
  const Fp rhs = fp_add(fp_mul(fp_sqr(x), x), fp_mul(fp_mul(fp_sqr(z), z), fp_small(3)));
  status[i] = (fp_eq(lhs, rhs) || fp_is_zero(z)) ? SYLOW_HIP_ST_OK : SYLOW_HIP_ST_NOT_ON_CURVE;
}
// ConstantTimeEq for projective points (group.rs:426-447)
__global__ void __launch_bounds__(BLOCK) k_g1_ct_eq(const u64* a, const u64* b, uint8_t* eq, size_t n) {
  size_t i = TID;
  if (i >= n) return;
  const Fp ax = load_fp(a, n, i, 0), ay = load_fp(a, n, i, 4), az = load_fp(a, n, i, 8);
  const Fp bx = load_fp(b, n, i, 0), by = load_fp(b, n, i, 4), bz = load_fp(b, n, i, 8);
  const bool iz = fp_is_zero(az), yz = fp_is_zero(bz);
  const bool same = fp_eq(fp_mul(ax, bz), fp_mul(bx, az)) && fp_eq(fp_mul(ay, bz), fp_mul(by, az));
  eq[i] = ((iz && yz) || (!iz && !yz && same)) ? 1 : 0;
}
// out_j = sum_i k_{j,i} * P_{j,i}: the aggregation loop of examples/threshold_signing.rs:124-143 (Lagrange-weighted partial
// signatures), one job per lane, terms walked in order with the reference's own scalar multiplication and complete addition.
// Term-major layout: element (job j, term i) lives at index i * n_jobs + j, so a wave reads consecutive addresses.
__global__ void HEAVY_BOUNDS k_g1_lincomb(const u64* pxy, const uint8_t* pinf, const u64* ks, u64* oxy, uint8_t* oinf, size_t n_jobs, size_t n_terms) {
  size_t j = TID;
  if (j >= n_jobs) return;
  const size_t n = n_jobs * n_terms;
  G1P acc{fp_zero(), fp_one(), fp_zero()};               // G1Projective::default() = identity
#pragma unroll 1
  for (size_t t = 0; t < n_terms; ++t) {
    const size_t i = t * n_jobs + j;
    G1P p = load_g1_flagged(pxy, pinf, n, i);
    u32 k[8];
    load_scalar(k, ks, n, i);
    acc = g1_add(acc, g1_scalar_mul(p, k));
  }
  Fp x, y; bool rinf;
  g1_to_affine(x, y, rinf, acc);
  store_fp(oxy, n_jobs, j, 0, x); store_fp(oxy, n_jobs, j, 4, y);
  oinf[j] = rinf ? 1 : 0;
}
// G1Affine::new (g1.rs:111-132): y^2 - x^3 == 3, the identity flag passes
__global__ void __launch_bounds__(BLOCK) k_g1_on_curve(const u64* pxy, const uint8_t* pinf, uint8_t* status, size_t n) {
  size_t i = TID;
  if (i >= n) return;
  const bool ok = (pinf && pinf[i]) || g1_on_curve_affine(load_fp(pxy, n, i, 0), load_fp(pxy, n, i, 4));
  status[i] = ok ? SYLOW_HIP_ST_OK : SYLOW_HIP_ST_NOT_ON_CURVE;
}
__global__ void __launch_bounds__(BLOCK) k_g1_normalize(const u64* pxyz, u64* oxy, uint8_t* oinf, size_t n) {
  size_t i = TID;
  if (i >= n) return;
  G1P p{load_fp(pxyz, n, i, 0), load_fp(pxyz, n, i, 4), load_fp(pxyz, n, i, 8)};
  Fp x, y; bool rinf;
  g1_to_affine(x, y, rinf, p);
  store_fp(oxy, n, i, 0, x); store_fp(oxy, n, i, 4, y);
  oinf[i] = rinf ? 1 : 0;
}

// ------------------------------------------------------------------ k * G1gen with a fixed-base table ----------
// G1Projective::generator() * k for a batch of scalars (test data, GroupTrait::rand): k mod r as 32 signed 8-bit digits against a
// per-device table T[w][j] = j 256^w G (j = 1..128, affine, reduced carry-free digits, 32 x 128 x 18 words = 295 KB): 32 complete
// additions, no doublings.  The twin of plk_group.hip's G2 table.
constexpr int COMB_WIN = 32, COMB_ENT = 128;
constexpr size_t COMB1_WORDS = (size_t)COMB_WIN * COMB_ENT * 18;
__global__ void HEAVY_BOUNDS k_g1_comb_table(i32* table) {
  const size_t e = TID;
  if (e >= (size_t)COMB_WIN * COMB_ENT) return;
  const int w = (int)(e / COMB_ENT), j = (int)(e % COMB_ENT) + 1;
  if (w == COMB_WIN - 1 && j > 64) return;            // k mod r < 2^254: the top digit is at most 0x30 + 1, these entries are never read
  u32 k[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
  for (int q = 0; q < 8; ++q) if (q == (w >> 2)) k[q] = (u32)j << (8 * (w & 3));
  const G1P g{fp_one(), fp_small(2), fp_one()};
  const G1P r = g1_scalar_mul(g, k);                  // GLV window product (k < 2^256 is reduced mod r inside)
  Fp x, y; bool inf;
  g1_to_affine(x, y, inf, r);                         // never the identity: j 256^w < r
  const F29 fx = f29_from_fp_reduced(x), fy = f29_from_fp_reduced(y);
  i32* dst = table + e * 18;
#pragma unroll
  for (int q = 0; q < 9; ++q) { dst[q] = fx.v[q]; dst[9 + q] = fy.v[q]; }
}
__global__ void HEAVY_BOUNDS k_g1_generator_mul(const u64* ks, const i32* __restrict__ table, u64* oxy, uint8_t* oinf, size_t n) {
  const size_t i = TID;
  if (i >= n) return;
  u32 k[8];
  load_scalar(k, ks, n, i);
  cond_sub_const(k, 0xf0000001u, 0x43e1f593u, 0x79b97091u, 0x2833e848u, 0x8181585du, 0xb85045b6u, 0xe131a029u, 0x30644e72u);   // k mod r (k < p < 2r)
  G1W res = proj_zero<OpsF29>();
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
    const i32* src = table + ((size_t)w * COMB_ENT + (size_t)(mag ? mag - 1 : 0)) * 18;
    F29 ex, ey;
#pragma unroll
    for (int q = 0; q < 9; ++q) { ex.v[q] = src[q]; ey.v[q] = src[9 + q]; }
    const bool nz = mag != 0;
    G1W q1;                                           // digit 0 adds the identity (0 : 1 : 0): the formulas are complete
    q1.x = OpsF29::select(OpsF29::zero(), ex, nz);
    q1.y = OpsF29::select(OpsF29::one(), OpsF29::select(ey, OpsF29::neg(ey), d < 0), nz);
    q1.z = OpsF29::select(OpsF29::zero(), OpsF29::one(), nz);
    res = proj_add_lazy<OpsF29>(res, q1);                        // leaves out of line: inlined (OpsF29I) this loop measured 3.25 against 3.14 ms per 2^20
  }
  Fp x, y; bool rinf;
  g1_to_affine(x, y, rinf, G1P{f29_to_fp(res.x), f29_to_fp(res.y), f29_to_fp(res.z)});
  store_fp(oxy, n, i, 0, x); store_fp(oxy, n, i, 4, y);
  oinf[i] = rinf ? 1 : 0;
}

// ------------------------------------------------------------------ sum of a batch of G1 points ----------
// sum_i P_i: the G1 side of aggregate verification (prod_i e(sig_i, G2gen) = e(sum_i sig_i, G2gen); the fold of
// examples/verify_multiple_messages_same_signer.rs:41-60).  Stages of SERIAL accumulation: with L lanes, lane t adds up elements
// t, t + L, t + 2 L, ... (a wavefront reads consecutive elements at every step) with the complete addition on the carry-free core,
// accumulator in registers -- no intermediate leaves the lane -- and writes ONE projective partial to element t of acc [12][stride]
// (in place when the input is acc itself: lane t is the only reader of element t and has read it before it writes).  Every stage
// divides the count by SUM_FOLD; the last <= 2 * BLOCK partials go through the one-block tail (a level per barrier), which also
// converts to affine.  (The round-3 form -- a binary tree with one launch per level and every intermediate through HBM on the
// saturated core -- took 4-6 ms per 2^20 points, neither issue- nor bandwidth-bound; this one is a few hundred microseconds.)
constexpr size_t SUM_FOLD = 16;
BN_DEV G1W g1w_load_proj(const u64* a, size_t stride, size_t i) {
  return G1W{f29_from_fp_reduced(load_fp(a, stride, i, 0)), f29_from_fp_reduced(load_fp(a, stride, i, 4)), f29_from_fp_reduced(load_fp(a, stride, i, 8))};
}
BN_DEV void g1w_store_proj(u64* a, size_t stride, size_t i, const G1W& r) {
  store_fp(a, stride, i, 0, f29_to_fp(r.x)); store_fp(a, stride, i, 4, f29_to_fp(r.y)); store_fp(a, stride, i, 8, f29_to_fp(r.z));
}
// affine points + identity flags (stride n) -> L partial sums in acc (stride acc_stride)
__global__ void HEAVY_BOUNDS k_g1_sum_fold_affine(const u64* pxy, const uint8_t* pinf, size_t n, size_t L, u64* acc, size_t acc_stride) {
  const size_t t = TID;
  if (t >= L) return;
  G1W res = proj_zero<OpsF29>();
#pragma unroll 1
  for (size_t i = t; i < n; i += L) {
    const bool inf = pinf && pinf[i];
    G1W q;                                              // the identity joins as (0 : 1 : 0): the formulas are complete
    q.x = OpsF29::select(f29_from_fp_reduced(load_fp(pxy, n, i, 0)), OpsF29::zero(), inf);
    q.y = OpsF29::select(f29_from_fp_reduced(load_fp(pxy, n, i, 4)), OpsF29::one(), inf);
    q.z = OpsF29::select(OpsF29::one(), OpsF29::zero(), inf);
    res = proj_add_lazy<OpsF29>(res, q);
  }
  g1w_store_proj(acc, acc_stride, t, res);
}
// m projective points of acc (stride `stride`) -> L partial sums, in place
__global__ void HEAVY_BOUNDS k_g1_sum_fold_proj(u64* acc, size_t stride, size_t m, size_t L) {
  const size_t t = TID;
  if (t >= L) return;
  G1W res = g1w_load_proj(acc, stride, t);
#pragma unroll 1
  for (size_t i = t + L; i < m; i += L) res = proj_add_lazy<OpsF29>(res, g1w_load_proj(acc, stride, i));
  g1w_store_proj(acc, stride, t, res);
}
// the last levels (m <= 2 * BLOCK live elements) and the finish in ONE block: a level per barrier instead of a launch per level
__global__ void __launch_bounds__(BLOCK) k_g1_sum_tail(u64* acc, size_t n, size_t m, u64* oxy, uint8_t* oinf, size_t stride, size_t col, int negate) {
  const size_t t = threadIdx.x;
  while (m > 1) {
    const size_t h = (m + 1) / 2;
    if (t + h < m) {
      const G1P a{load_fp(acc, n, t, 0), load_fp(acc, n, t, 4), load_fp(acc, n, t, 8)};
      const G1P b{load_fp(acc, n, t + h, 0), load_fp(acc, n, t + h, 4), load_fp(acc, n, t + h, 8)};
      const G1P r = g1_add(a, b);
      store_fp(acc, n, t, 0, r.x); store_fp(acc, n, t, 4, r.y); store_fp(acc, n, t, 8, r.z);
    }
    __threadfence_block();
    __syncthreads();
    m = h;
  }
  if (t != 0) return;
  const G1P p = n ? G1P{load_fp(acc, n, 0, 0), load_fp(acc, n, 0, 4), load_fp(acc, n, 0, 8)} : G1P{fp_zero(), fp_one(), fp_zero()};
  Fp x, y; bool inf;
  g1_to_affine(x, y, inf, p);
  if (negate && !inf) y = fp_neg(y);
  store_fp(oxy, stride, col, 0, x); store_fp(oxy, stride, col, 4, y);
  oinf[col] = inf ? 1 : 0;
}

// ------------------------------------------------------------------ hash / BLS kernels ----------
// k_hash_to_g1 lives in hash.hip (a unit of its own: compiled for four wavefronts per SIMD)
// SvdW::unchecked_map_to_point (svdw.rs:180-262) on its own: u -> (x, y) on the curve; status CANNOT_HASH where the reference
// returns MapError (cannot happen on this curve)
__global__ void __launch_bounds__(BLOCK) k_svdw_map(const u64* u, u64* oxy, uint8_t* status, size_t n) {
  size_t i = TID;
  if (i >= n) return;
  // through svdw_map2 -- the form hash_to_g1 uses, two maps sharing one inversion -- with the neighbouring element as the second map
  // (its result is dropped: element i^1 computes it again as ITS first), so that the entry point's parity tests cover the shared
  // inversion, including the inv0 cases of either or both operands
  const size_t j = (i ^ 1) < n ? (i ^ 1) : i;
  Fp x, y, x2, y2;
  const bool ok = svdw_map2(x, y, x2, y2, load_fp(u, n, i, 0), load_fp(u, n, j, 0));
  store_fp(oxy, n, i, 0, x); store_fp(oxy, n, i, 4, y);
  if (status) status[i] = ok ? SYLOW_HIP_ST_OK : SYLOW_HIP_ST_CANNOT_HASH;
}
// Fp::compute_naf (fp.rs:653-662): the two 256-bit masks (np, nm) of the +1 / -1 digits of the RAW 256-bit value, digit_i = np_i - nm_i
__global__ void __launch_bounds__(BLOCK) k_compute_naf(const u64* k, u64* onp, u64* onm, size_t n) {
  size_t i = TID;
  if (i >= n) return;
  u64 x[4], xh[4], x3[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) x[j] = k[(size_t)j * n + i];
#pragma unroll
  for (int j = 0; j < 4; ++j) xh[j] = (x[j] >> 1) | (j < 3 ? (x[j + 1] << 63) : 0);
  u64 c = 0;
#pragma unroll
  for (int j = 0; j < 4; ++j) {                        // x + (x >> 1) modulo 2^256
    const u64 s = x[j] + xh[j];
    const u64 s2 = s + c;
    c = (s < x[j] || s2 < s) ? 1 : 0;
    x3[j] = s2;
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const u64 d = xh[j] ^ x3[j];
    onp[(size_t)j * n + i] = x3[j] & d;
    onm[(size_t)j * n + i] = xh[j] & d;
  }
}
// Expander::hash_to_field(msg, 2, 48) (hasher.rs:84-128) over XMDExpander<Keccak256>::expand_message (hasher.rs:201-250)
__global__ void __launch_bounds__(BLOCK) k_hash_to_field(const uint8_t* msgs, const u64* off, DstPrime dp, u64* out, size_t n) {
  size_t i = TID;
  if (i >= n) return;
  u64 em[12];
  expand_message_xmd96_words(em, msgs + off[i], (size_t)(off[i + 1] - off[i]), dp);
  const u64 a[6] = {em[0], em[1], em[2], em[3], em[4], em[5]}, b[6] = {em[6], em[7], em[8], em[9], em[10], em[11]};
  store_fp(out, n, i, 0, fp_from_be48_words(a));
  store_fp(out, n, i, 4, fp_from_be48_words(b));
}
// ------------------------------------------------------------------ EVM alt_bn128 adapter -------
// Byte-level batches of the three precompile shapes of examples/reth_bn128.rs:99-217 (EIP-196/197):
// 32-byte big-endian field elements (Fp::from_be_bytes rejects >= p, fp.rs:686-719), (0,0) encodes the
// identity, G1 points must be on the curve (G1Affine::new, g1.rs:111-132), G2 points on the twist AND in
// the r-torsion (G2Projective::new, g2.rs:460-525); G2 is encoded x.c1 | x.c0 | y.c1 | y.c0.
// status: OK, DECODE_ERROR (= Bn128FieldPointNotAMember), NOT_ON_CURVE / NOT_IN_SUBGROUP (= Bn128AffineGFailedToCreate).
__global__ void __launch_bounds__(BLOCK) k_evm_ecadd(const uint8_t* in, uint8_t* out, uint8_t* status, size_t n) {
  size_t i = TID;
  if (i >= n) return;
  G1P a, b;
  uint8_t sa = evm_read_g1(a, in + 128 * i), sb = evm_read_g1(b, in + 128 * i + 64);
  uint8_t st = sa ? sa : sb;
  status[i] = st;
  if (st) { __builtin_memset(out + 64 * i, 0, 64); return; }
  evm_write_g1(out + 64 * i, g1_add(a, b));
}
__global__ void HEAVY_BOUNDS k_evm_ecmul(const uint8_t* in, uint8_t* out, uint8_t* status, size_t n) {
  size_t i = TID;
  if (i >= n) return;
  G1P a;
  uint8_t st = evm_read_g1(a, in + 96 * i);
  status[i] = st;
  if (st) { __builtin_memset(out + 64 * i, 0, 64); return; }
  u32 k[8];
  evm_read_scalar(k, in + 96 * i + 64);
  evm_write_g1(out + 64 * i, g1_scalar_mul(a, k));
}

// ------------------------------------------------------------------ wire formats -----------------
// G1Affine::to_be_bytes / from_be_bytes (g1.rs:151-280): x | y big-endian, bit 7 of byte 0 = infinity flag,
// identity encoded as (0, 1) + flag; decoding masks the flag, rejects coordinates >= p (DECODE_ERROR),
// a set flag with (x, y) != (0, 1) (DECODE_ERROR) and off-curve points (NOT_ON_CURVE).
// G2: x.c1 | x.c0 | y.c1 | y.c0 (g2.rs:319-433); decoding also runs the subgroup check of G2Projective::new.
__global__ void __launch_bounds__(BLOCK) k_g1_to_bytes(const u64* xy, const uint8_t* inf, uint8_t* out, size_t n) {
  size_t i = TID;
  if (i >= n) return;
  bool z = inf && inf[i];
  Fp x = z ? fp_zero() : fp_reduce_plain(load_plain(xy, n, i, 0));
  Fp y = z ? fp_from_limbs(1, 0, 0, 0, 0, 0, 0, 0) : fp_reduce_plain(load_plain(xy, n, i, 4));
  write_be_fp(out + 64 * i, x);
  write_be_fp(out + 64 * i + 32, y);
  if (z) out[64 * i] |= 0x80;
}
__global__ void __launch_bounds__(BLOCK) k_g1_from_bytes(const uint8_t* in, u64* xy, uint8_t* inf, uint8_t* status, size_t n) {
  size_t i = TID;
  if (i >= n) return;
  const uint8_t* b = in + 64 * i;
  const bool flag = (b[0] >> 7) & 1;
  Fp x, y;
  bool ok = read_be_fp(x, b, true);
  ok = read_be_fp(y, b + 32) && ok;
  uint8_t st = SYLOW_HIP_ST_OK;
  bool is01 = fp_is_zero(x) && fp_eq(y, fp_from_limbs(1, 0, 0, 0, 0, 0, 0, 0));
  if (!ok) st = SYLOW_HIP_ST_DECODE_ERROR;
  else if (flag) st = is01 ? SYLOW_HIP_ST_OK : SYLOW_HIP_ST_DECODE_ERROR;
  else if (!g1_on_curve_affine(fp_to_mont(x), fp_to_mont(y))) st = SYLOW_HIP_ST_NOT_ON_CURVE;
  bool z = flag || st != SYLOW_HIP_ST_OK;
  store_plain(xy, n, i, 0, z ? fp_zero() : x);
  store_plain(xy, n, i, 4, z ? fp_from_limbs(1, 0, 0, 0, 0, 0, 0, 0) : y);
  inf[i] = z ? 1 : 0;
  status[i] = st;
}

// GroupProjective::double (group.rs:339-386) on affine inputs
__global__ void __launch_bounds__(BLOCK) k_g1_double(const u64* axy, const uint8_t* ainf, u64* oxy, uint8_t* oinf, size_t n) {
  size_t i = TID;
  if (i >= n) return;
  G1P a = load_g1_flagged(axy, ainf, n, i);
  G1P r = g1_double(a);
  Fp x, y; bool rinf;
  g1_to_affine(x, y, rinf, r);
  store_fp(oxy, n, i, 0, x); store_fp(oxy, n, i, 4, y);
  oinf[i] = rinf ? 1 : 0;
}

// ================================================================== C ABI ======================
namespace g1h {
size_t g1_comb_bytes() { return COMB1_WORDS * sizeof(bn254::i32); }
int32_t build_g1_comb(bn254::i32* table, void* stream) {
  k_g1_comb_table<<<GRID((size_t)COMB_WIN * COMB_ENT)>>>(table); LAUNCHED();
}
static size_t fold_lanes(size_t m) { return (m + SUM_FOLD - 1) / SUM_FOLD; }
int32_t sum(const uint64_t* p_xy, const uint8_t* p_inf, size_t n, uint64_t* acc, uint64_t* out_xy, uint8_t* out_inf, size_t stride, size_t col, int negate, void* stream) {
  if (!n) return sum_tree(acc, 0, out_xy, out_inf, stride, col, negate, stream);
  const size_t L = fold_lanes(n);
  k_g1_sum_fold_affine<<<GRID(L)>>>(p_xy, p_inf, n, L, acc, n);
  return sum_tree_strided(acc, n, L, out_xy, out_inf, stride, col, negate, stream);
}
int32_t sum_tree(uint64_t* acc, size_t n, uint64_t* out_xy, uint8_t* out_inf, size_t stride, size_t col, int negate, void* stream) {
  return sum_tree_strided(acc, n, n, out_xy, out_inf, stride, col, negate, stream);
}
// the first m of the projective points in acc [12][acc_stride] -> their sum (affine, column `col` of an SoA array of stride `stride`)
int32_t sum_tree_strided(uint64_t* acc, size_t acc_stride, size_t m, uint64_t* out_xy, uint8_t* out_inf, size_t stride, size_t col, int negate, void* stream) {
  while (m > 2 * BLOCK) {
    const size_t L = fold_lanes(m);
    k_g1_sum_fold_proj<<<GRID(L)>>>(acc, acc_stride, m, L);
    m = L;
  }
  k_g1_sum_tail<<<1, BLOCK, 0, (hipStream_t)stream>>>(acc, acc_stride, m, out_xy, out_inf, stride, col, negate); LAUNCHED();
}
}  // namespace g1h

// window tables of n lanes in a leased global block, one contiguous KB per lane (bn254_pairing.hpp: ProjTableGlobal);
// a failed lease keeps them in the stack frame (NULL)
static uint8_t* g1_window_tables(host::Lease& ws, size_t n, void* stream) {
  if (ws.acquire(n * G1_TABLE_BYTES_PER_LANE, (hipStream_t)stream) == SYLOW_HIP_OK) return (uint8_t*)ws.p;
  (void)hipGetLastError();
  return nullptr;
}

extern "C" {
int32_t sylow_hip_g1_sum_batch(const uint64_t* p_xy, const uint8_t* p_inf, size_t n, uint64_t* out_xy, uint8_t* out_inf, void* stream) {
  ARGCHK(out_xy && out_inf && (p_xy || !n));
  host::Lease ws;
  int32_t rc = ws.acquire(12 * (n ? n : 1) * sizeof(u64), (hipStream_t)stream);
  if (rc != SYLOW_HIP_OK) return rc;
  rc = g1h::sum(p_xy, p_inf, n, (uint64_t*)ws.p, out_xy, out_inf, 1, 0, 0, stream);
  const int32_t r2 = ws.release();
  return rc != SYLOW_HIP_OK ? rc : r2;
}
int32_t sylow_hip_g1_scalar_mul_batch(const uint64_t* p_xy, const uint8_t* p_inf, const uint64_t* k, uint64_t* out_xy, uint8_t* out_inf, size_t n, void* stream) {
  ARGCHK(p_xy && k && out_xy && out_inf); if (!n) return SYLOW_HIP_OK;
  // single calls and small batches: eight lanes per product (sign_wide.hip) -- one product 1.0 -> ~0.35 ms
  if (plkh::wide_batch_max() != 0 && n <= g1h::sign_wide_max()) return g1h::g1_scalar_mul_wide(p_xy, p_inf, k, out_xy, out_inf, n, stream);
  // window tables in a leased global block, one contiguous KB per lane (bn254_pairing.hpp: G1TableGlobal); a failed
  // lease keeps them in the stack frame
  host::Lease ws;
  uint8_t* tables = g1_window_tables(ws, n, stream);
  k_g1_scalar_mul<<<GRID(n)>>>(p_xy, p_inf, k, out_xy, out_inf, n, tables);
  const hipError_t e = hipGetLastError();
  const int32_t rc = ws.release();
  return e != hipSuccess ? host::fail(e, "kernel launch") : rc;
}
int32_t sylow_hip_g1_add_batch(const uint64_t* a_xy, const uint8_t* a_inf, const uint64_t* b_xy, const uint8_t* b_inf, uint64_t* out_xy, uint8_t* out_inf, size_t n, void* stream) {
  ARGCHK(a_xy && b_xy && out_xy && out_inf); if (!n) return SYLOW_HIP_OK;
  k_g1_add<<<GRID(n)>>>(a_xy, a_inf, b_xy, b_inf, out_xy, out_inf, n); LAUNCHED();
}
int32_t sylow_hip_g1_sub_batch(const uint64_t* a_xy, const uint8_t* a_inf, const uint64_t* b_xy, const uint8_t* b_inf, uint64_t* out_xy, uint8_t* out_inf, size_t n, void* stream) {
  ARGCHK(a_xy && b_xy && out_xy && out_inf); if (!n) return SYLOW_HIP_OK;
  k_g1_sub<<<GRID(n)>>>(a_xy, a_inf, b_xy, b_inf, out_xy, out_inf, n); LAUNCHED();
}
int32_t sylow_hip_g1_projective_new_batch(const uint64_t* p_xyz, uint8_t* status, size_t n, void* stream) {
  ARGCHK(p_xyz && status); if (!n) return SYLOW_HIP_OK; k_g1_projective_new<<<GRID(n)>>>(p_xyz, status, n); LAUNCHED();
}
int32_t sylow_hip_g1_ct_eq_batch(const uint64_t* a_xyz, const uint64_t* b_xyz, uint8_t* eq, size_t n, void* stream) {
  ARGCHK(a_xyz && b_xyz && eq); if (!n) return SYLOW_HIP_OK; k_g1_ct_eq<<<GRID(n)>>>(a_xyz, b_xyz, eq, n); LAUNCHED();
}
int32_t sylow_hip_g1_lincomb_batch(const uint64_t* p_xy, const uint8_t* p_inf, const uint64_t* k, uint64_t* out_xy, uint8_t* out_inf, size_t n_jobs, size_t n_terms, void* stream) {
  ARGCHK(out_xy && out_inf && (n_terms == 0 || (p_xy && k))); if (!n_jobs) return SYLOW_HIP_OK;
  k_g1_lincomb<<<GRID(n_jobs)>>>(p_xy, p_inf, k, out_xy, out_inf, n_jobs, n_terms); LAUNCHED();
}
int32_t sylow_hip_g1_on_curve_batch(const uint64_t* p_xy, const uint8_t* p_inf, uint8_t* status, size_t n, void* stream) {
  ARGCHK(p_xy && status); if (!n) return SYLOW_HIP_OK; k_g1_on_curve<<<GRID(n)>>>(p_xy, p_inf, status, n); LAUNCHED();
}
int32_t sylow_hip_g1_normalize_batch(const uint64_t* p_xyz, uint64_t* out_xy, uint8_t* out_inf, size_t n, void* stream) {
  ARGCHK(p_xyz && out_xy && out_inf); if (!n) return SYLOW_HIP_OK;
  k_g1_normalize<<<GRID(n)>>>(p_xyz, out_xy, out_inf, n); LAUNCHED();
}
int32_t sylow_hip_g1_generator_mul_batch(const uint64_t* k, uint64_t* out_xy, uint8_t* out_inf, size_t n, void* stream) {
  ARGCHK(k && out_xy && out_inf); if (!n) return SYLOW_HIP_OK;
  const bn254::i32* table = nullptr;
  int32_t rc = host::g1_gen_comb(&table, (hipStream_t)stream);
  if (rc != SYLOW_HIP_OK) return rc;
  k_g1_generator_mul<<<GRID(n)>>>(k, table, out_xy, out_inf, n); LAUNCHED();
}
int32_t sylow_hip_hash_to_g1_batch(const uint8_t* msgs, const uint64_t* msg_offsets, const uint8_t* dst_host, size_t dst_len,
                                   uint64_t* out_xy, uint8_t* out_inf, size_t n, void* stream) {
  ARGCHK(msgs && msg_offsets && out_xy && out_inf); if (!n) return SYLOW_HIP_OK;
  DstPrime dp; host::dst_arg(dp, dst_host, dst_len);
  return g1h::hash_to_g1_dst(msgs, msg_offsets, dp, out_xy, out_inf, n, 0, stream);
}
int32_t sylow_hip_svdw_map_batch(const uint64_t* u, uint64_t* out_xy, uint8_t* status, size_t n, void* stream) {
  ARGCHK(u && out_xy); if (!n) return SYLOW_HIP_OK;
  k_svdw_map<<<GRID(n)>>>(u, out_xy, status, n); LAUNCHED();
}
int32_t sylow_hip_fp_compute_naf_batch(const uint64_t* k, uint64_t* out_np, uint64_t* out_nm, size_t n, void* stream) {
  ARGCHK(k && out_np && out_nm); if (!n) return SYLOW_HIP_OK;
  k_compute_naf<<<GRID(n)>>>(k, out_np, out_nm, n); LAUNCHED();
}
int32_t sylow_hip_hash_to_field_batch(const uint8_t* msgs, const uint64_t* msg_offsets, const uint8_t* dst_host, size_t dst_len,
                                      uint64_t* out_u, size_t n, void* stream) {
  ARGCHK(msgs && msg_offsets && out_u); if (!n) return SYLOW_HIP_OK;
  DstPrime dp; host::dst_arg(dp, dst_host, dst_len);
  k_hash_to_field<<<GRID(n)>>>(msgs, msg_offsets, dp, out_u, n); LAUNCHED();
}
int32_t sylow_hip_evm_ecadd_batch(const uint8_t* in, uint8_t* out, uint8_t* status, size_t n, void* stream) {
  ARGCHK(in && out && status); if (!n) return SYLOW_HIP_OK;
  k_evm_ecadd<<<GRID(n)>>>(in, out, status, n); LAUNCHED();
}
int32_t sylow_hip_evm_ecmul_batch(const uint8_t* in, uint8_t* out, uint8_t* status, size_t n, void* stream) {
  ARGCHK(in && out && status); if (!n) return SYLOW_HIP_OK;
  // single calls and small batches: eight lanes per product (sign_wide.hip) -- one ecMul 1.1 -> ~0.36 ms
  if (plkh::wide_batch_max() != 0 && n <= g1h::sign_wide_max()) return g1h::evm_ecmul_wide(in, out, status, n, stream);
  k_evm_ecmul<<<GRID(n)>>>(in, out, status, n); LAUNCHED();
}
int32_t sylow_hip_g1_to_be_bytes_batch(const uint64_t* p_xy, const uint8_t* p_inf, uint8_t* out, size_t n, void* stream) {
  ARGCHK(p_xy && out); if (!n) return SYLOW_HIP_OK; k_g1_to_bytes<<<GRID(n)>>>(p_xy, p_inf, out, n); LAUNCHED();
}
int32_t sylow_hip_g1_from_be_bytes_batch(const uint8_t* in, uint64_t* out_xy, uint8_t* out_inf, uint8_t* status, size_t n, void* stream) {
  ARGCHK(in && out_xy && out_inf && status); if (!n) return SYLOW_HIP_OK; k_g1_from_bytes<<<GRID(n)>>>(in, out_xy, out_inf, status, n); LAUNCHED();
}
int32_t sylow_hip_g1_double_batch(const uint64_t* a_xy, const uint8_t* a_inf, uint64_t* out_xy, uint8_t* out_inf, size_t n, void* stream) {
  ARGCHK(a_xy && out_xy && out_inf); if (!n) return SYLOW_HIP_OK; k_g1_double<<<GRID(n)>>>(a_xy, a_inf, out_xy, out_inf, n); LAUNCHED();
}

}  // extern "C"
