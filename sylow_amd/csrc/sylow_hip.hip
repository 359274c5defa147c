// libsylow_hip.so: kernels + the C ABI declared in include/sylow_hip.h.
// gfx950 only.  No CPU fallback: every entry point launches HIP kernels or fails.
#include "../../include/sylow_hip.h"

#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <mutex>
#include <cstring>

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

// ------------------------------------------------------------------ Fp kernels ----------------
enum { OP_ADD = 0, OP_SUB = 1, OP_MUL = 2, OP_SQR = 3, OP_NEG = 4, OP_INV = 5 };

// HBM-bound kernels (96 B per element): each lane handles TWO adjacent elements so that every limb
// plane is read / written with one 16-byte access per lane (1 KiB per wavefront instruction).
// Inputs are reduced like Fp::new by conditional subtraction (no multiplications); a*b is then ONE Barrett
// multiplication in the canonical domain (fp_mulmod_plain) -- no round trip through Montgomery form.
BN_DEV void load_plain2(Fp& e0, Fp& e1, const u64* __restrict__ base, size_t n, size_t i) {
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const ulonglong2 w = *reinterpret_cast<const ulonglong2*>(base + (size_t)k * n + i);
    e0.v[2 * k] = (u32)w.x; e0.v[2 * k + 1] = (u32)(w.x >> 32);
    e1.v[2 * k] = (u32)w.y; e1.v[2 * k + 1] = (u32)(w.y >> 32);
  }
}
BN_DEV void store_plain2(u64* __restrict__ base, size_t n, size_t i, const Fp& e0, const Fp& e1) {
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    ulonglong2 w;
    w.x = (u64)e0.v[2 * k] | ((u64)e0.v[2 * k + 1] << 32);
    w.y = (u64)e1.v[2 * k] | ((u64)e1.v[2 * k + 1] << 32);
    *reinterpret_cast<ulonglong2*>(base + (size_t)k * n + i) = w;
  }
}
template <int OP, int FR>
BN_DEV Fp fp_binop_one(const Fp& x, const Fp& y) {
  if (FR) {                                            // the scalar field, same macro-generated API (fp.rs:556-565)
    if (OP == OP_MUL) return fr_mulmod_inline(x, y);
    Fp xr = fr_reduce_plain(x), yr = fr_reduce_plain(y);
    return (OP == OP_ADD) ? fr_add(xr, yr) : fr_sub(xr, yr);
  }
  if (OP == OP_MUL) return fp_mulmod_plain(x, y);      // Barrett takes any 256-bit operands
  Fp xr = fp_reduce_plain(x), yr = fp_reduce_plain(y);
  return (OP == OP_ADD) ? fp_add(xr, yr) : fp_sub(xr, yr);
}
template <int OP, int FR>
__global__ void __launch_bounds__(BLOCK) k_fp_binop(const u64* __restrict__ a, const u64* __restrict__ b, u64* __restrict__ out, size_t n) {
  size_t i = 2 * TID;
  if (i >= n) return;
  const bool vec = ((n & 1) == 0);          // planes stay 16-byte aligned only for even n
  if (vec) {
    Fp x0, x1, y0, y1;
    load_plain2(x0, x1, a, n, i);
    load_plain2(y0, y1, b, n, i);
    store_plain2(out, n, i, fp_binop_one<OP, FR>(x0, y0), fp_binop_one<OP, FR>(x1, y1));
  } else {
    for (size_t j = i; j < n && j < i + 2; ++j)
      store_plain(out, n, j, 0, fp_binop_one<OP, FR>(load_plain(a, n, j, 0), load_plain(b, n, j, 0)));
  }
}
template <int OP, int FR>
BN_DEV Fp fp_unop_one(const Fp& x) {
  if (FR) {
    if (OP == OP_SQR) return fr_mulmod_inline(x, x);
    if (OP == OP_NEG) return fr_neg(fr_reduce_plain(x));
    return fr_inv(fr_reduce_plain(x));
  }
  if (OP == OP_SQR) return fp_mulmod_plain(x, x);
  if (OP == OP_NEG) return fp_neg(fp_reduce_plain(x));
  return fp_from_mont(fp_inv(fp_to_mont(x)));
}
template <int OP, int FR>
__global__ void __launch_bounds__(BLOCK) k_fp_unop(const u64* __restrict__ a, u64* __restrict__ out, size_t n) {
  size_t i = 2 * TID;
  if (i >= n) return;
  const bool vec = ((n & 1) == 0);
  if (vec) {
    Fp x0, x1;
    load_plain2(x0, x1, a, n, i);
    store_plain2(out, n, i, fp_unop_one<OP, FR>(x0), fp_unop_one<OP, FR>(x1));
  } else {
    for (size_t j = i; j < n && j < i + 2; ++j) store_plain(out, n, j, 0, fp_unop_one<OP, FR>(load_plain(a, n, j, 0)));
  }
}

// Fp::pow(U256) with a per-element exponent (fp.rs:451-457): uniform 256-step square-and-multiply (a lane whose bit is
// clear multiplies by one); sqrt (fp.rs:611-616); is_square (fp.rs:625-631, as a Jacobi symbol); sgn0 is bit 0 of the value
__global__ void __launch_bounds__(BLOCK) k_fp_pow(const u64* a, const u64* e, u64* out, size_t n) {
  size_t i = TID;
  if (i >= n) return;
  const Fp x = load_fp(a, n, i, 0), one = fp_one();
  const Fp ev = load_plain(e, n, i, 0);
  Fp r = one;
#pragma unroll 1
  for (int b = 255; b >= 0; --b) {
    r = fp_mul(r, r);
    r = fp_mul(r, fp_select(one, x, (ev.v[b >> 5] >> (b & 31)) & 1));
  }
  store_fp(out, n, i, 0, r);
}
__global__ void __launch_bounds__(BLOCK) k_fp_sqrt(const u64* a, u64* out, uint8_t* ok, uint8_t* sq, size_t n) {
  size_t i = TID;
  if (i >= n) return;
  const Fp x = load_fp(a, n, i, 0);
  if (out) {
    const Fp r = fp_mul(x, fp_pow_pm3_quarter(x));            // x^((p+1)/4)
    store_fp(out, n, i, 0, r);
    if (ok) ok[i] = fp_eq(fp_mul(r, r), x) ? 1 : 0;
  }
  if (sq) sq[i] = fp_is_square(x) ? 1 : 0;
}

// ------------------------------------------------------------------ tower test hooks ----------
__global__ void __launch_bounds__(BLOCK) k_fp2_op(int op, const u64* a, const u64* b, u64* out, size_t n) {
  size_t i = TID;
  if (i >= n) return;
  Fp2 x = load_fp2(a, n, i, 0), r;
  if (op == OP_MUL) r = fp2_mul(x, load_fp2(b, n, i, 0));
  else if (op == OP_SQR) r = fp2_sqr(x);
  else r = fp2_inv(x);
  store_fp2(out, n, i, 0, r);
}
__global__ void HEAVY_BOUNDS k_fp6_op(int op, const u64* a, const u64* b, u64* out, size_t n) {
  size_t i = TID;
  if (i >= n) return;
  Fp6 x, y, r;
  load_fp6(x, a, n, i, 0);
  if (op == OP_MUL) { load_fp6(y, b, n, i, 0); fp6_mul(r, x, y); }
  else fp6_inv(r, x);
  store_fp6(out, n, i, 0, r);
}
enum { OP12_MUL = 0, OP12_SQR = 1, OP12_INV = 2, OP12_FROB1 = 3, OP12_FROB2 = 4, OP12_FROB3 = 5, OP12_SPARSE = 6, OP12_CYCSQR = 7,
       OP12_U_MUL = 8, OP12_U_CYCSQR = 9, OP12_EXPZ = 10, OP12_EXPZ_SAT = 11 };
__global__ void HEAVY_BOUNDS k_fp12_op(int op, const u64* a, const u64* b, u64* out, size_t n) {
  size_t i = TID;
  if (i >= n) return;
  Fp12 x, y, r;
  load_fp12(x, a, n, i);
  switch (op) {
    case OP12_MUL: load_fp12(y, b, n, i); fp12_mul(r, x, y); break;
    case OP12_SQR: fp12_sqr(r, x); break;
    case OP12_INV: fp12_inv(r, x); break;
    case OP12_FROB1: fp12_frobenius<1>(r, x); break;
    case OP12_FROB2: fp12_frobenius<2>(r, x); break;
    case OP12_FROB3: fp12_frobenius<3>(r, x); break;
    case OP12_CYCSQR: cyclotomic_sqr(r, x); break;
    case OP12_U_MUL: { load_fp12(y, b, n, i); U12 ux, uy, ur; u12_from_fp12(ux, x); u12_reduce(ux); u12_from_fp12(uy, y); u12_reduce(uy); u12_mul(ur, ux, uy); u12_to_fp12(r, ur); break; }
    case OP12_U_CYCSQR: { U12 ux, ur; u12_from_fp12(ux, x); u12_reduce(ux); u12_cyclotomic_sqr(ur, ux); u12_to_fp12(r, ur); break; }
    case OP12_EXPZ: exp_by_neg_z(r, x); break;
    case OP12_EXPZ_SAT: exp_by_neg_z_sat(r, x); break;
    default: {
      Fp2 l0 = load_fp2(b, n, i, 0), lvw = load_fp2(b, n, i, 8), lvv = load_fp2(b, n, i, 16);
      fp12_sparse_mul(r, x, l0, lvw, lvv);
    }
  }
  store_fp12(out, n, i, r);
}

// ------------------------------------------------------------------ group kernels --------------
BN_DEV void load_scalar(u32 (&k)[8], const u64* base, size_t n, size_t i) {
  // scalars are Fp values: reduce like Fp::new so that k >= p behaves as in the reference
  Fp s = fp_from_mont(fp_to_mont(load_plain(base, n, i, 0)));
#pragma unroll
  for (int j = 0; j < 8; ++j) k[j] = s.v[j];
}
__global__ void HEAVY_BOUNDS k_g1_scalar_mul(const u64* pxy, const uint8_t* pinf, const u64* ks, u64* oxy, uint8_t* oinf, size_t n) {
  size_t i = TID;
  if (i >= n) return;
  bool inf = pinf && pinf[i];
  G1P p{load_fp(pxy, n, i, 0), load_fp(pxy, n, i, 4), inf ? fp_zero() : fp_one()};
  u32 k[8];
  load_scalar(k, ks, n, i);
  G1P r = g1_scalar_mul(p, k);
  Fp x, y; bool rinf;
  g1_to_affine(x, y, rinf, r);
  store_fp(oxy, n, i, 0, x); store_fp(oxy, n, i, 4, y);
  oinf[i] = rinf ? 1 : 0;
}
__global__ void HEAVY_BOUNDS k_g2_scalar_mul(const u64* pxy, const uint8_t* pinf, const u64* ks, u64* oxy, uint8_t* oinf, size_t n) {
  size_t i = TID;
  if (i >= n) return;
  bool inf = pinf && pinf[i];
  G2P p{load_fp2(pxy, n, i, 0), load_fp2(pxy, n, i, 8), inf ? fp2_zero() : fp2_one()};
  u32 k[8];
  load_scalar(k, ks, n, i);
  G2P r;
  g2_scalar_mul(r, p, k);
  Fp2 x, y; bool rinf;
  g2_to_affine(x, y, rinf, r);
  store_fp2(oxy, n, i, 0, x); store_fp2(oxy, n, i, 8, y);
  oinf[i] = rinf ? 1 : 0;
}
__global__ void __launch_bounds__(BLOCK) k_g1_add(const u64* axy, const uint8_t* ainf, const u64* bxy, const uint8_t* binf, u64* oxy, uint8_t* oinf, size_t n) {
  size_t i = TID;
  if (i >= n) return;
  G1P a{load_fp(axy, n, i, 0), load_fp(axy, n, i, 4), (ainf && ainf[i]) ? fp_zero() : fp_one()};
  G1P b{load_fp(bxy, n, i, 0), load_fp(bxy, n, i, 4), (binf && binf[i]) ? fp_zero() : fp_one()};
  G1P r = g1_add(a, b);
  Fp x, y; bool rinf;
  g1_to_affine(x, y, rinf, r);
  store_fp(oxy, n, i, 0, x); store_fp(oxy, n, i, 4, y);
  oinf[i] = rinf ? 1 : 0;
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
    G1P p{load_fp(pxy, n, i, 0), load_fp(pxy, n, i, 4), (pinf && pinf[i]) ? fp_zero() : fp_one()};
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
__global__ void __launch_bounds__(BLOCK) k_g2_normalize(const u64* pxyz, u64* oxy, uint8_t* oinf, size_t n) {
  size_t i = TID;
  if (i >= n) return;
  G2P p{load_fp2(pxyz, n, i, 0), load_fp2(pxyz, n, i, 8), load_fp2(pxyz, n, i, 16)};
  Fp2 x, y; bool rinf;
  g2_to_affine(x, y, rinf, p);
  store_fp2(oxy, n, i, 0, x); store_fp2(oxy, n, i, 8, y);
  oinf[i] = rinf ? 1 : 0;
}
// g2.rs:460-525 on an affine input: on-curve, then (x+1)Q + psi(xQ) + psi^2(xQ) == psi^3(2xQ)
__global__ void HEAVY_BOUNDS k_g2_subgroup_check(const u64* qxy, const uint8_t* qinf, uint8_t* status, size_t n) {
  size_t i = TID;
  if (i >= n) return;
  if (qinf && qinf[i]) { status[i] = SYLOW_HIP_ST_OK; return; }   // Z == 0 passes both tests (g2.rs:469,510)
  Fp2 x = load_fp2(qxy, n, i, 0), y = load_fp2(qxy, n, i, 8);
  if (!g2_on_curve_affine(x, y)) { status[i] = SYLOW_HIP_ST_NOT_ON_CURVE; return; }
  G2P q{x, y, fp2_one()};
  const u32 bx[8] = {(u32)BN_BLS_X, (u32)(BN_BLS_X >> 32), 0, 0, 0, 0, 0, 0};
  G2P a;
  g2_scalar_mul(a, q, bx);                       // xQ
  // psi on projective coordinates: conj is a field automorphism, so psi(X:Y:Z) = (eps0 conj X : eps1 conj Y : conj Z)
  auto psi = [](G2P& r, const G2P& p) {
    r.x = fp2_mul(fp2_const(C_EPS_EXP0), fp2_conj(p.x));
    r.y = fp2_mul(fp2_const(C_EPS_EXP1), fp2_conj(p.y));
    r.z = fp2_conj(p.z);
  };
  G2P b, c, l, r;
  psi(b, a);                                      // psi(xQ)
  g2_add(a, a, q);                                // (x+1)Q
  psi(c, b);                                      // psi^2(xQ)
  g2_add(l, c, b);
  g2_add(l, l, a);                                // lhs
  psi(r, c);
  g2_double(r, r);                                // psi^3(2xQ)
  G2P nl = proj_neg<OpsFp2>(l);
  g2_add(r, r, nl);
  status[i] = fp2_is_zero(r.z) ? SYLOW_HIP_ST_OK : SYLOW_HIP_ST_NOT_IN_SUBGROUP;
}

// ------------------------------------------------------------------ pairing kernels -------------
__global__ void HEAVY_BOUNDS k_miller_loop(const u64* pxy, const u64* qxy, u64* fout, size_t n) {
  size_t i = TID;
  if (i >= n) return;
  Fp px = load_fp(pxy, n, i, 0), py = load_fp(pxy, n, i, 4);
  Fp2 qx = load_fp2(qxy, n, i, 0), qy = load_fp2(qxy, n, i, 8);
  Fp12 f;
  miller_loop(f, px, py, qx, qy);
  store_fp12(fout, n, i, f);
}
__global__ void HEAVY_BOUNDS k_final_exp(const u64* fin, u64* gout, size_t n) {
  size_t i = TID;
  if (i >= n) return;
  Fp12 f, g;
  load_fp12(f, fin, n, i);
  final_exponentiation(g, f);
  store_fp12(gout, n, i, g);
}
// pairing.rs:870-893
__global__ void HEAVY_BOUNDS k_pairing(const u64* pxy, const uint8_t* pinf, const u64* qxy, const uint8_t* qinf, u64* gout, size_t n) {
  size_t i = TID;
  if (i >= n) return;
  bool either_zero = (pinf && pinf[i]) || (qinf && qinf[i]);
  Fp12 f, g;
  if (either_zero) {
    fp12_set_one(g);   // Miller value forced to one; final_exponentiation(1) == 1
  } else {
    Fp px = load_fp(pxy, n, i, 0), py = load_fp(pxy, n, i, 4);
    Fp2 qx = load_fp2(qxy, n, i, 0), qy = load_fp2(qxy, n, i, 8);
    miller_loop(f, px, py, qx, qy);
    final_exponentiation(g, f);
  }
  store_fp12(gout, n, i, g);
}


// ------------------------------------------------------------------ multi-pairing ---------------
// glued_miller_loop + final_exponentiation (pairing.rs:970-1037): one job per lane, the job's pairs
// share each squaring of the accumulator.  Pairs are processed KMAX at a time; the product of the
// chunk accumulators equals the reference's single accumulator exactly (Fp12 multiplication is
// exact and commutative, and (prod f_c)^2 * prod lines is the same recurrence).
//
// The schedule is WAVE-UNIFORM: every lane walks the same number of chunks and the same number of
// pair slots per chunk (the wavefront maximum); a lane whose job has fewer pairs steps a dummy point
// and multiplies its accumulator by the unit line (1, 0, 0), which leaves it bit-identical.  The
// shared digit schedule is uniform anyway, so the whole Miller loop runs without divergence.
constexpr int KMAX = 4;
struct PairState { G2P r; Fp2 qx, qy; Fp px, py; bool qinf; bool live; };

BN_DEV int wave_max(int v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) { int o = __shfl_xor(v, off); v = o > v ? o : v; }
  return __builtin_amdgcn_readfirstlane(v);
}

__global__ void HEAVY_BOUNDS k_multi_pairing(const u64* pxy, const uint8_t* pinf, const u64* qxy, const uint8_t* qinf,
                                             const u64* offsets, size_t n_jobs, size_t n_pairs, int skip_infinity,
                                             u64* gout, uint8_t* is_one) {
  size_t job = TID;
  const bool active = job < n_jobs;           // no early return: every lane takes part in the wave reductions
  size_t next = active ? offsets[job] : 0, hi = active ? offsets[job + 1] : 0;
  Fp12 acc;
  fp12_set_one(acc);
  PairState st[KMAX];
  const u64 nz = BN_ATE_NAF_NZ, ng = BN_ATE_NAF_NEG;
  const Fp2 gx{fp_const(C_G2_GEN[0]), fp_const(C_G2_GEN[1])}, gy{fp_const(C_G2_GEN[2]), fp_const(C_G2_GEN[3])};
#pragma unroll 1
  while (wave_max(next < hi ? 1 : 0)) {
    // gather up to KMAX pairs of this lane's job; dead slots hold the generator (any curve point does)
    int k = 0;
#pragma unroll 1
    for (int slot = 0; slot < KMAX; ++slot) {
      bool have = false;
      size_t idx = 0;
      // advance to the next pair this lane actually multiplies in (bounded scan: uniform trip count not needed, no calls inside)
      while (next < hi) {
        bool pi = pinf && pinf[next], qi = qinf && qinf[next];
        idx = next++;
        if (!(skip_infinity && (pi || qi))) { have = true; break; }   // EIP-197: identity pairs contribute 1
      }
      PairState& s = st[slot];
      s.live = have;
      size_t src = have ? idx : 0;
      bool qi = have && qinf && qinf[src];
      bool ld = have && n_pairs != 0;
      s.px = ld ? load_fp(pxy, n_pairs, src, 0) : fp_one();
      s.py = ld ? load_fp(pxy, n_pairs, src, 4) : fp_one();
      s.qx = ld ? load_fp2(qxy, n_pairs, src, 0) : gx;
      s.qy = ld ? load_fp2(qxy, n_pairs, src, 8) : gy;
      s.qinf = qi;
      // G2Projective::from(&G2Affine): Z = infinity ? 0 : 1 (group.rs:506-517); the reference's glued
      // loop never looks at the flag again, neither do we in replay mode
      s.r = G2P{s.qx, s.qy, qi ? fp2_zero() : fp2_one()};
      if (have) k = slot + 1;
    }
    const int kw = wave_max(k);
    if (kw == 0) continue;
    Acc12 f;
    f.set_one();
    Fp2 l0, l1, l2;
    const Fp2 u0 = fp2_one(), u1 = fp2_zero();
    auto apply = [&](PairState& s) {          // f *= line, or *= 1 for a dead slot
      bool lv = s.live;
      f.sparse(fp2_select(u0, l0, lv), fp2_select(u1, fp2_scale(l1, s.py), lv), fp2_select(u1, fp2_scale(l2, s.px), lv));
    };
#pragma unroll 1
    for (int i = 0; i < 64; ++i) {
      f.square();
#pragma unroll 1
      for (int j = 0; j < kw; ++j) { g2_doubling_step(st[j].r, l0, l1, l2); apply(st[j]); }
      if ((nz >> (63 - i)) & 1) {
        bool neg = (ng >> (63 - i)) & 1;
#pragma unroll 1
        for (int j = 0; j < kw; ++j) {
          Fp2 by = neg ? fp2_neg(st[j].qy) : st[j].qy;
          g2_addition_step(st[j].r, st[j].qx, by, l0, l1, l2);
          apply(st[j]);
        }
      }
    }
    // the two Frobenius additions; endomorphism() returns self for the identity (g2.rs:141-143)
#pragma unroll 1
    for (int step = 0; step < 2; ++step) {
#pragma unroll 1
      for (int j = 0; j < kw; ++j) {
        Fp2 q1x, q1y, q2x, q2y;
        g2_psi_affine(q1x, q1y, st[j].qx, st[j].qy);
        g2_psi_affine(q2x, q2y, q1x, q1y);
        bool qi = st[j].qinf;
        q1x = fp2_select(q1x, st[j].qx, qi); q1y = fp2_select(q1y, st[j].qy, qi);
        q2x = fp2_select(q2x, st[j].qx, qi); q2y = fp2_select(q2y, st[j].qy, qi);
        if (step == 0) g2_addition_step(st[j].r, q1x, q1y, l0, l1, l2);
        else g2_addition_step(st[j].r, q2x, fp2_neg(q2y), l0, l1, l2);
        apply(st[j]);
      }
    }
    fp12_mul(acc, acc, f.get());
  }
  Fp12 g;
  final_exponentiation(g, acc);
  if (active) {
    if (gout) store_fp12(gout, n_jobs, job, g);
    if (is_one) {
      Fp12 one;
      fp12_set_one(one);
      is_one[job] = fp12_eq(g, one) ? 1 : 0;
    }
  }
}

// ------------------------------------------------------------------ hash / BLS kernels ----------
__global__ void __launch_bounds__(BLOCK) k_hash_to_g1(const uint8_t* msgs, const u64* off, DstPrime dp, u64* oxy, uint8_t* oinf, uint8_t* status, size_t n) {
  size_t i = TID;
  if (i >= n) return;
  G1P h;
  bool ok = hash_to_g1(h, msgs + off[i], (size_t)(off[i + 1] - off[i]), dp);
  Fp x, y; bool inf;
  g1_to_affine(x, y, inf, h);
  store_fp(oxy, n, i, 0, x); store_fp(oxy, n, i, 4, y);
  oinf[i] = inf ? 1 : 0;
  if (status) status[i] = ok ? SYLOW_HIP_ST_OK : SYLOW_HIP_ST_CANNOT_HASH;
}
// Expander::hash_to_field(msg, 2, 48) (hasher.rs:84-128) over XMDExpander<Keccak256>::expand_message (hasher.rs:201-250)
__global__ void __launch_bounds__(BLOCK) k_hash_to_field(const uint8_t* msgs, const u64* off, DstPrime dp, u64* out, size_t n) {
  size_t i = TID;
  if (i >= n) return;
  uint8_t em[96];
  expand_message_xmd96(em, msgs + off[i], (size_t)(off[i + 1] - off[i]), dp);
  store_fp(out, n, i, 0, fp_from_be48(em));
  store_fp(out, n, i, 4, fp_from_be48(em + 48));
}
// lib.rs:179-187
__global__ void HEAVY_BOUNDS k_bls_sign(const u64* sk, const uint8_t* msgs, const u64* off, DstPrime dp, u64* oxy, uint8_t* oinf, size_t n) {
  size_t i = TID;
  if (i >= n) return;
  G1P h;
  hash_to_g1(h, msgs + off[i], (size_t)(off[i + 1] - off[i]), dp);
  u32 k[8];
  load_scalar(k, sk, n, i);
  G1P s = g1_scalar_mul(h, k);
  Fp x, y; bool inf;
  g1_to_affine(x, y, inf, s);
  store_fp(oxy, n, i, 0, x); store_fp(oxy, n, i, 4, y);
  oinf[i] = inf ? 1 : 0;
}
// ------------------------------------------------------------------ fused BLS verify ------------
// The batch-verify shape sylow's own examples recommend (examples/verify_multiple_messages_same_signer.rs:41-60,
// threshold_signing.rs:92-121): e(sig, G2gen) * e(-H(msg), pk) == 1 with ONE shared-squaring Miller loop
// and ONE final exponentiation.  Pairs with an identity contribute 1, exactly as pairing() treats them
// (pairing.rs:876-886), so for points in G1 x G2 the boolean equals verify()'s (lib.rs:223-236).
// The G2 generator is fixed, so its 87 line-coefficient triples (pairing.rs:676-708) are computed once
// per device into g_g2gen_lines and read with wave-uniform (scalar) loads.
__device__ u32 g_g2gen_lines[87 * 48];
// the 87 line-coefficient triples of G2Affine::precompute (pairing.rs:676-708) for one affine point, Montgomery form
template <class PUT>
BN_DEV void g2_line_table(const Fp2& qx, const Fp2& qy, PUT put) {
  G2P r{qx, qy, fp2_one()};
  const Fp2 nqy = fp2_neg(qy);
  Fp2 l0, l1, l2;
  int idx = 0;
  const u64 nz = BN_ATE_NAF_NZ, ng = BN_ATE_NAF_NEG;
#pragma unroll 1
  for (int i = 0; i < 64; ++i) {
    g2_doubling_step(r, l0, l1, l2); put(idx++, l0, l1, l2);
    if ((nz >> (63 - i)) & 1) { g2_addition_step(r, qx, ((ng >> (63 - i)) & 1) ? nqy : qy, l0, l1, l2); put(idx++, l0, l1, l2); }
  }
  Fp2 q1x, q1y, q2x, q2y;
  g2_psi_affine(q1x, q1y, qx, qy);
  g2_psi_affine(q2x, q2y, q1x, q1y);
  g2_addition_step(r, q1x, q1y, l0, l1, l2); put(idx++, l0, l1, l2);
  g2_addition_step(r, q2x, fp2_neg(q2y), l0, l1, l2); put(idx++, l0, l1, l2);
}
// one lane builds the Montgomery-form u32 table [87][48] of a single point: the generator (qxy == nullptr) or
// element `idx` of an SoA G2 array
__global__ void k_g2_lines(const u64* qxy, size_t n, size_t idx, u32* table) {
  if (TID != 0) return;
  Fp2 qx{fp_const(C_G2_GEN[0]), fp_const(C_G2_GEN[1])}, qy{fp_const(C_G2_GEN[2]), fp_const(C_G2_GEN[3])};
  if (qxy) { qx = load_fp2(qxy, n, idx, 0); qy = load_fp2(qxy, n, idx, 8); }
  u32* t = table ? table : g_g2gen_lines;
  g2_line_table(qx, qy, [&](int at, const Fp2& l0, const Fp2& l1, const Fp2& l2) {
    const Fp* src[6] = {&l0.c0, &l0.c1, &l1.c0, &l1.c1, &l2.c0, &l2.c1};
    for (int c = 0; c < 6; ++c) for (int j = 0; j < 8; ++j) t[at * 48 + c * 8 + j] = src[c]->v[j];
  });
}
// G2Affine::precompute for a batch (pairing.rs:676-708): canonical words, SoA [87*24][n], triple t of point i at
// words 24t .. 24t+23 = (ell.0, ell.1, ell.2) as Fp2 each -- the reference's [Ell; 87] in its own order
__global__ void HEAVY_BOUNDS k_g2_precompute(const u64* qxy, u64* coeffs, size_t n) {
  size_t i = TID;
  if (i >= n) return;
  Fp2 qx = load_fp2(qxy, n, i, 0), qy = load_fp2(qxy, n, i, 8);
  g2_line_table(qx, qy, [&](int at, const Fp2& l0, const Fp2& l1, const Fp2& l2) {
    store_fp2(coeffs, n, i, 24 * at, l0); store_fp2(coeffs, n, i, 24 * at + 8, l1); store_fp2(coeffs, n, i, 24 * at + 16, l2);
  });
}
BN_DEV Fp2 table_fp2(const u32* table, int at, int c) {
  const u32* t = table + at * 48 + c * 16;
  return Fp2{fp_from_limbs(t[0], t[1], t[2], t[3], t[4], t[5], t[6], t[7]), fp_from_limbs(t[8], t[9], t[10], t[11], t[12], t[13], t[14], t[15])};
}

// G2PreComputed::miller_loop (pairing.rs:590-619) against a precomputed line table (Montgomery u32 [87][48])
BN_NOINLINE void miller_loop_table(Fp12& fout, const Fp& px, const Fp& py, const u32* table) {
  Acc12 f;
  f.set_one();
  const u64 nz = BN_ATE_NAF_NZ;
  int idx = 0;
#pragma unroll 1
  for (int i = 0; i < 64; ++i) {
    f.square();
    f.line(table_fp2(table, idx, 0), table_fp2(table, idx, 1), table_fp2(table, idx, 2), px, py);
    ++idx;
    if ((nz >> (63 - i)) & 1) {
      f.line(table_fp2(table, idx, 0), table_fp2(table, idx, 1), table_fp2(table, idx, 2), px, py);
      ++idx;
    }
  }
  f.line(table_fp2(table, idx, 0), table_fp2(table, idx, 1), table_fp2(table, idx, 2), px, py);
  ++idx;
  f.line(table_fp2(table, idx, 0), table_fp2(table, idx, 1), table_fp2(table, idx, 2), px, py);
  fout = f.get();
}

// lib.rs:223-236: ok = pairing(sig, G2gen) == pairing(H(msg), pk), two full pairings as written
// (the generator's line coefficients are read from the precomputed table: same G2PreComputed values)
__global__ void HEAVY_BOUNDS k_bls_verify(const u64* pkxy, const uint8_t* pkinf, const uint8_t* msgs, const u64* off, DstPrime dp,
                                          const u64* sigxy, const uint8_t* siginf, uint8_t* okout, size_t n) {
  size_t i = TID;
  if (i >= n) return;
  G1P h;
  hash_to_g1(h, msgs + off[i], (size_t)(off[i + 1] - off[i]), dp);
  Fp hx, hy; bool hinf;
  g1_to_affine(hx, hy, hinf, h);
  Fp12 f, lhs, rhs;
  // lhs = pairing(sig, G2gen)
  if (siginf && siginf[i]) {
    fp12_set_one(lhs);
  } else {
    Fp sx = load_fp(sigxy, n, i, 0), sy = load_fp(sigxy, n, i, 4);
    miller_loop_table(f, sx, sy, g_g2gen_lines);     // Q = G2 generator: its [Ell; 87] comes from the per-device table
    final_exponentiation(lhs, f);
  }
  // rhs = pairing(H, pk)
  if (hinf || (pkinf && pkinf[i])) {
    fp12_set_one(rhs);
  } else {
    Fp2 qx = load_fp2(pkxy, n, i, 0), qy = load_fp2(pkxy, n, i, 8);
    miller_loop(f, hx, hy, qx, qy);
    final_exponentiation(rhs, f);
  }
  okout[i] = fp12_eq(lhs, rhs) ? 1 : 0;
}



// PK_TABLE: every element is checked against ONE public key whose line table was precomputed (the same-signer
// shape of examples/verify_multiple_messages_same_signer.rs): both pairs read wave-uniform tables and the loop
// contains no G2 arithmetic at all.
template <bool PK_TABLE>
__global__ void HEAVY_BOUNDS k_bls_verify_fused(const u64* pkxy, const uint8_t* pkinf, const u32* pk_table, const uint8_t* msgs, const u64* off, DstPrime dp,
                                                const u64* sigxy, const uint8_t* siginf, uint8_t* okout, size_t n) {
  size_t i = TID;
  const bool active = i < n;
  const size_t ii = active ? i : 0;          // out-of-range lanes recompute element 0 (uniform control flow), store nothing
  G1P h;
  hash_to_g1(h, msgs + off[ii], (size_t)(off[ii + 1] - off[ii]), dp);
  Fp hx, hy; bool hinf;
  g1_to_affine(hx, hy, hinf, h);
  hy = fp_neg(hy);                            // pair B is (-H, pk)
  const bool liveA = !(siginf && siginf[ii]);
  const bool liveB = !(hinf || (pkinf && pkinf[PK_TABLE ? 0 : ii]));
  const Fp sx = load_fp(sigxy, n, ii, 0), sy = load_fp(sigxy, n, ii, 4);
  const Fp2 gx{fp_const(C_G2_GEN[0]), fp_const(C_G2_GEN[1])}, gy{fp_const(C_G2_GEN[2]), fp_const(C_G2_GEN[3])};
  // a dead pair B steps the generator instead (any curve point keeps the arithmetic defined) and multiplies by the unit line
  const Fp2 qx = PK_TABLE ? gx : fp2_select(gx, load_fp2(pkxy, n, ii, 0), liveB), qy = PK_TABLE ? gy : fp2_select(gy, load_fp2(pkxy, n, ii, 8), liveB);
  const Fp2 nqy = fp2_neg(qy);
  G2P r{qx, qy, fp2_one()};
  Acc12 f;
  f.set_one();
  Fp2 l0, l1, l2;
  const Fp2 u0 = fp2_one(), u1 = fp2_zero();
  auto lineA = [&](int at) {
    Fp2 a0 = table_fp2(g_g2gen_lines, at, 0), a1 = fp2_scale(table_fp2(g_g2gen_lines, at, 1), sy), a2 = fp2_scale(table_fp2(g_g2gen_lines, at, 2), sx);
    f.sparse(fp2_select(u0, a0, liveA), fp2_select(u1, a1, liveA), fp2_select(u1, a2, liveA));
  };
  auto lineB = [&](int at) {
    if (PK_TABLE) { l0 = table_fp2(pk_table, at, 0); l1 = table_fp2(pk_table, at, 1); l2 = table_fp2(pk_table, at, 2); }
    f.sparse(fp2_select(u0, l0, liveB), fp2_select(u1, fp2_scale(l1, hy), liveB), fp2_select(u1, fp2_scale(l2, hx), liveB));
  };
  const u64 nz = BN_ATE_NAF_NZ, ng = BN_ATE_NAF_NEG;
  int idx = 0;
#pragma unroll 1
  for (int it = 0; it < 64; ++it) {
    f.square();
    lineA(idx);
    if (!PK_TABLE) g2_doubling_step(r, l0, l1, l2);
    lineB(idx);
    ++idx;
    if ((nz >> (63 - it)) & 1) {
      lineA(idx);
      if (!PK_TABLE) g2_addition_step(r, qx, ((ng >> (63 - it)) & 1) ? nqy : qy, l0, l1, l2);
      lineB(idx);
      ++idx;
    }
  }
  Fp2 q1x, q1y, q2x, q2y;
  if (!PK_TABLE) { g2_psi_affine(q1x, q1y, qx, qy); g2_psi_affine(q2x, q2y, q1x, q1y); }
  lineA(idx);
  if (!PK_TABLE) g2_addition_step(r, q1x, q1y, l0, l1, l2);
  lineB(idx);
  ++idx;
  lineA(idx);
  if (!PK_TABLE) g2_addition_step(r, q2x, fp2_neg(q2y), l0, l1, l2);
  lineB(idx);
  Fp12 g, one;
  final_exponentiation(g, f.get());
  fp12_set_one(one);
  if (active) okout[i] = fp12_eq(g, one) ? 1 : 0;
}


#include "pair_kernels.hpp"

// ------------------------------------------------------------------ EVM alt_bn128 adapter -------
// Byte-level batches of the three precompile shapes of examples/reth_bn128.rs:99-217 (EIP-196/197):
// 32-byte big-endian field elements (Fp::from_be_bytes rejects >= p, fp.rs:686-719), (0,0) encodes the
// identity, G1 points must be on the curve (G1Affine::new, g1.rs:111-132), G2 points on the twist AND in
// the r-torsion (G2Projective::new, g2.rs:460-525); G2 is encoded x.c1 | x.c0 | y.c1 | y.c0.
// status: OK, DECODE_ERROR (= Bn128FieldPointNotAMember), NOT_ON_CURVE / NOT_IN_SUBGROUP (= Bn128AffineGFailedToCreate).
BN_DEV bool read_be_fp(Fp& out, const uint8_t* b) {         // returns false when the value is >= p
  Fp x;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const uint8_t* q = b + 28 - 4 * j;
    x.v[j] = ((u32)q[0] << 24) | ((u32)q[1] << 16) | ((u32)q[2] << 8) | (u32)q[3];
  }
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
__global__ void __launch_bounds__(BLOCK) k_evm_ecadd(const uint8_t* in, uint8_t* out, uint8_t* status, size_t n) {
  size_t i = TID;
  if (i >= n) return;
  G1P a, b;
  uint8_t sa = evm_read_g1(a, in + 128 * i), sb = evm_read_g1(b, in + 128 * i + 64);
  uint8_t st = sa ? sa : sb;
  status[i] = st;
  if (st) { for (int k = 0; k < 64; ++k) out[64 * i + k] = 0; return; }
  evm_write_g1(out + 64 * i, g1_add(a, b));
}
__global__ void HEAVY_BOUNDS k_evm_ecmul(const uint8_t* in, uint8_t* out, uint8_t* status, size_t n) {
  size_t i = TID;
  if (i >= n) return;
  G1P a;
  uint8_t st = evm_read_g1(a, in + 96 * i);
  status[i] = st;
  if (st) { for (int k = 0; k < 64; ++k) out[64 * i + k] = 0; return; }
  Fp kx;
  read_be_fp(kx, in + 96 * i + 64);
  // EIP-196 accepts any 256-bit scalar; G1 has prime order r, so reduce mod r (2^256 < 6r).  (The reference
  // adapter unwraps Fr::from_be_bytes and would panic for k >= r, reth_bn128.rs:144.)
  u32 k[8] = {kx.v[0], kx.v[1], kx.v[2], kx.v[3], kx.v[4], kx.v[5], kx.v[6], kx.v[7]};
  cond_sub_const(k, 0xc0000004u, 0x0f87d64fu, 0xe6e5c245u, 0xa0cfa121u, 0x06056174u, 0xe14116dau, 0x84c680a6u, 0xc19139cbu);  // 4r
  cond_sub_const(k, 0xe0000002u, 0x87c3eb27u, 0xf372e122u, 0x5067d090u, 0x0302b0bau, 0x70a08b6du, 0xc2634053u, 0x60c89ce5u);  // 2r
  cond_sub_const(k, 0xf0000001u, 0x43e1f593u, 0x79b97091u, 0x2833e848u, 0x8181585du, 0xb85045b6u, 0xe131a029u, 0x30644e72u);  // r
  evm_write_g1(out + 64 * i, g1_scalar_mul(a, k));
}
// one LANE PAIR per 192-byte pair: decode + validate into the SoA arrays the multi-pairing kernel consumes.  Both lanes decode
// the six field elements; the G2 checks (twist equation, subgroup) run on the lane-pair Fp2 (pair_kernels.hpp).
__global__ void HEAVY_BOUNDS k_evm_decode_pairs(const uint8_t* in, size_t n_pairs, u64* pxy, uint8_t* pinf, u64* qxy, uint8_t* qinf, uint8_t* pst) {
  const size_t t = TID, i = t >> 1;
  const bool odd = (t & 1) != 0;
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
__global__ void __launch_bounds__(BLOCK) k_evm_pair_finalize(const uint8_t* pst, const u64* offsets, size_t n_jobs, const uint8_t* is_one, uint8_t* result, uint8_t* status) {
  size_t j = TID;
  if (j >= n_jobs) return;
  uint8_t st = SYLOW_HIP_ST_OK;
  for (u64 k = offsets[j]; k < offsets[j + 1]; ++k) if (!st && pst[k]) st = pst[k];     // first failing pair, like the `?` in run_pair
  status[j] = st;
  result[j] = st ? 0 : is_one[j];
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
  uint8_t b[64];
  for (int k = 0; k < 64; ++k) b[k] = in[64 * i + k];
  bool flag = (b[0] >> 7) & 1;
  b[0] &= 0x7f;
  Fp x, y;
  bool ok = read_be_fp(x, b);
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
  const size_t t = TID, i = t >> 1;
  const bool odd = (t & 1) != 0;
  if (i >= n) return;
  uint8_t b[128];
  for (int k = 0; k < 128; ++k) b[k] = in[128 * i + k];
  bool flag = (b[0] >> 7) & 1;
  b[0] &= 0x7f;
  Fp xc1, xc0, yc1, yc0;
  bool ok = read_be_fp(xc1, b);
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


// test hook for the carry-free core (bn254_f29.hpp): op 0: to_fp(from_fp(a)) (must be a); 1: product through
// f29_mul; 2: a*b + b*a through f29_dot2; 3: lazy (a + b) - b + a normalised then * 1 ... all compared with the
// saturated core by tests/test_gpu_fields.py
__global__ void __launch_bounds__(BLOCK) k_f29_hook(int op, const u64* a, const u64* b, u64* out, size_t n) {
  size_t i = TID;
  if (i >= n) return;
  Fp x = load_fp(a, n, i, 0), y = load_fp(b, n, i, 0), r;
  F29 fx = f29_from_fp(x), fy = f29_from_fp(y);
  if (op == 0) r = f29_to_fp(fx);
  else if (op == 1) r = f29_to_fp(f29_mul(fx, fy));
  else if (op == 2) r = f29_to_fp(f29_dot2(fx, fy, fy, fx));
  else if (op == 4) r = fp_inv(x);                       // safegcd
  else if (op == 5) r = fp_inv_fermat(x);                // x^(p-2) on the carry-free exponentiation chain
  else if (op == 6) r = f29_to_fp(f29_sqr(f29_reduce_from([&](int k) { return (i64)fx.v[k]; })));
  else {
    F29 t = f29_norm(f29_sub(f29_add(f29_add(fx, fy), fx), fy));     // 2x as a lazy value (L <= 3), normalised
    r = f29_to_fp(f29_mul(t, f29_sub(fy, fx)));                        // 2x * (y - x)
  }
  store_fp(out, n, i, 0, r);
}


// ------------------------------------------------------------------ f4: Gt * Fr, G2 add, doublings ---------
// Mul<&Fr> for &Gt (gt.rs:161-187): the reference's own algorithm -- 256-step signed-digit square-and-multiply
// on generic Fp12 squares (Gt::double = Fp12::square, gt.rs:268-270), negative digits multiply by the conjugate
// -- so the value matches even for inputs outside the cyclotomic subgroup.  The scalar is the Fr VALUE.
__global__ void HEAVY_BOUNDS k_gt_pow(const u64* g, const u64* ks, u64* out, size_t n) {
  size_t i = TID;
  if (i >= n) return;
  Fp12 a, na, buf[2];
  load_fp12(a, g, n, i);
  fp12_conj(na, a);
  // digits of fp.rs:653-662 on the raw 256-bit scalar (Fr values are < r < p: no reduction involved)
  u32 k[8], xh[8], x3[8], np[8], nm[8];
  {
    Fp kp = load_plain(ks, n, i, 0);
#pragma unroll
    for (int j = 0; j < 8; ++j) k[j] = kp.v[j];
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) xh[j] = (k[j] >> 1) | (j < 7 ? (k[j + 1] << 31) : 0);
  u64 c = 0;
#pragma unroll
  for (int j = 0; j < 8; ++j) { c += (u64)k[j] + xh[j]; x3[j] = (u32)c; c >>= 32; }
#pragma unroll
  for (int j = 0; j < 8; ++j) { u32 cc = xh[j] ^ x3[j]; np[j] = x3[j] & cc; nm[j] = xh[j] & cc; }
  int cur = 0;
  fp12_set_one(buf[0]);
  // wave-uniform schedule: every lane squares and multiplies each step; lanes whose digit is zero multiply by one
  Fp12 one;
  fp12_set_one(one);
#pragma unroll 1
  for (int b = 255; b >= 0; --b) {
    fp12_sqr(buf[cur ^ 1], buf[cur]); cur ^= 1;
    const bool bp = (np[b >> 5] >> (b & 31)) & 1, bm = (nm[b >> 5] >> (b & 31)) & 1;
    if (__any(bp || bm)) {
      Fp12 m;
      const Fp12& src = bp ? a : na;
      // select per lane: a, conj(a) or one
      m.c0 = (bp || bm) ? src.c0 : one.c0;
      m.c1 = (bp || bm) ? src.c1 : one.c1;
      fp12_mul(buf[cur ^ 1], buf[cur], m); cur ^= 1;
    }
  }
  store_fp12(out, n, i, buf[cur]);
}
__global__ void HEAVY_BOUNDS k_g2_add(const u64* axy, const uint8_t* ainf, const u64* bxy, const uint8_t* binf, u64* oxy, uint8_t* oinf, size_t n) {
  size_t i = TID;
  if (i >= n) return;
  G2P a{load_fp2(axy, n, i, 0), load_fp2(axy, n, i, 8), (ainf && ainf[i]) ? fp2_zero() : fp2_one()};
  G2P b{load_fp2(bxy, n, i, 0), load_fp2(bxy, n, i, 8), (binf && binf[i]) ? fp2_zero() : fp2_one()};
  G2P r;
  g2_add(r, a, b);
  Fp2 x, y; bool rinf;
  g2_to_affine(x, y, rinf, r);
  store_fp2(oxy, n, i, 0, x); store_fp2(oxy, n, i, 8, y);
  oinf[i] = rinf ? 1 : 0;
}
// GroupProjective::double (group.rs:339-386) on affine inputs
__global__ void __launch_bounds__(BLOCK) k_g1_double(const u64* axy, const uint8_t* ainf, u64* oxy, uint8_t* oinf, size_t n) {
  size_t i = TID;
  if (i >= n) return;
  G1P a{load_fp(axy, n, i, 0), load_fp(axy, n, i, 4), (ainf && ainf[i]) ? fp_zero() : fp_one()};
  G1P r = g1_double(a);
  Fp x, y; bool rinf;
  g1_to_affine(x, y, rinf, r);
  store_fp(oxy, n, i, 0, x); store_fp(oxy, n, i, 4, y);
  oinf[i] = rinf ? 1 : 0;
}
__global__ void HEAVY_BOUNDS k_g2_double(const u64* axy, const uint8_t* ainf, u64* oxy, uint8_t* oinf, size_t n) {
  size_t i = TID;
  if (i >= n) return;
  G2P a{load_fp2(axy, n, i, 0), load_fp2(axy, n, i, 8), (ainf && ainf[i]) ? fp2_zero() : fp2_one()};
  G2P r;
  g2_double(r, a);
  Fp2 x, y; bool rinf;
  g2_to_affine(x, y, rinf, r);
  store_fp2(oxy, n, i, 0, x); store_fp2(oxy, n, i, 8, y);
  oinf[i] = rinf ? 1 : 0;
}

// ------------------------------------------------------------------ layout helpers --------------
__global__ void __launch_bounds__(BLOCK) k_aos_to_soa(const u64* __restrict__ aos, u64* __restrict__ soa, size_t words, size_t n) {
  size_t t = TID;
  if (t >= words * n) return;
  size_t w = t / n, i = t % n;
  soa[t] = aos[i * words + w];
}
__global__ void __launch_bounds__(BLOCK) k_soa_to_aos(const u64* __restrict__ soa, u64* __restrict__ aos, size_t words, size_t n) {
  size_t t = TID;
  if (t >= words * n) return;
  size_t w = t / n, i = t % n;
  aos[i * words + w] = soa[t];
}
__global__ void __launch_bounds__(BLOCK) k_flags_all(const uint8_t* flags, size_t n, int32_t* out) {
  // out pre-set to 1; any zero flag clears it
  size_t i = TID;
  bool bad = (i < n) && (flags[i] == 0);
  if (__ballot(bad) != 0 && (threadIdx.x & 63) == 0) atomicAnd(out, 0);
}

// ================================================================== C ABI ======================
static thread_local char g_err[256] = "";
static int32_t fail(hipError_t e, const char* what) {
  snprintf(g_err, sizeof(g_err), "%s: %s", what, hipGetErrorString(e));
  return SYLOW_HIP_E_HIP;
}
#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return fail(e_, #x); } while (0)
#define ARGCHK(c) do { if (!(c)) { snprintf(g_err, sizeof(g_err), "bad argument: %s", #c); return SYLOW_HIP_E_ARG; } } while (0)
#define GRID(n) dim3((unsigned)(((n) + BLOCK - 1) / BLOCK)), dim3(BLOCK), 0, (hipStream_t)stream
static bool single_lane();
static int32_t workspace(void** out, size_t bytes, hipStream_t st);
#define LAUNCHED() do { hipError_t e_ = hipGetLastError(); if (e_ != hipSuccess) return fail(e_, "kernel launch"); return SYLOW_HIP_OK; } while (0)

extern "C" {

const char* sylow_hip_last_error(void) { return g_err; }
int32_t sylow_hip_device_count(void) {
  int c = 0;
  if (hipGetDeviceCount(&c) != hipSuccess) return 0;
  return c;
}
int32_t sylow_hip_init(int32_t device) {
  int c = 0;
  if (hipGetDeviceCount(&c) != hipSuccess || c == 0) { snprintf(g_err, sizeof(g_err), "no HIP device"); return SYLOW_HIP_E_NO_DEVICE; }
  ARGCHK(device >= 0 && device < c);
  HIPCHK(hipSetDevice(device));
  hipDeviceProp_t prop;
  HIPCHK(hipGetDeviceProperties(&prop, device));
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
    snprintf(g_err, sizeof(g_err), "device %d is %s; this library is built for gfx950 only", device, prop.gcnArchName);
    return SYLOW_HIP_E_NO_DEVICE;
  }
  return SYLOW_HIP_OK;
}
int32_t sylow_hip_malloc(void** dptr, size_t bytes) { ARGCHK(dptr); HIPCHK(hipMalloc(dptr, bytes ? bytes : 1)); return SYLOW_HIP_OK; }
int32_t sylow_hip_free(void* dptr) { HIPCHK(hipFree(dptr)); return SYLOW_HIP_OK; }
int32_t sylow_hip_memcpy_h2d(void* dst, const void* src, size_t bytes, void* stream) {
  HIPCHK(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, (hipStream_t)stream)); return SYLOW_HIP_OK;
}
int32_t sylow_hip_memcpy_d2h(void* dst, const void* src, size_t bytes, void* stream) {
  HIPCHK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, (hipStream_t)stream)); return SYLOW_HIP_OK;
}
int32_t sylow_hip_stream_sync(void* stream) { HIPCHK(hipStreamSynchronize((hipStream_t)stream)); return SYLOW_HIP_OK; }
int32_t sylow_hip_aos_to_soa(const uint64_t* aos, uint64_t* soa, size_t words, size_t n, void* stream) {
  ARGCHK(aos && soa); if (!n || !words) return SYLOW_HIP_OK;
  k_aos_to_soa<<<GRID(words * n)>>>(aos, soa, words, n); LAUNCHED();
}
int32_t sylow_hip_soa_to_aos(const uint64_t* soa, uint64_t* aos, size_t words, size_t n, void* stream) {
  ARGCHK(aos && soa); if (!n || !words) return SYLOW_HIP_OK;
  k_soa_to_aos<<<GRID(words * n)>>>(soa, aos, words, n); LAUNCHED();
}

#define FP_BIN(field, FR, name, OP)                                                                                 \
  int32_t sylow_hip_##field##_##name##_batch(const uint64_t* a, const uint64_t* b, uint64_t* out, size_t n, void* stream) { \
    ARGCHK(a && b && out); if (!n) return SYLOW_HIP_OK;                                                           \
    k_fp_binop<OP, FR><<<GRID((n + 1) / 2)>>>(a, b, out, n); LAUNCHED();                                             \
  }
FP_BIN(fp, 0, add, OP_ADD) FP_BIN(fp, 0, sub, OP_SUB) FP_BIN(fp, 0, mul, OP_MUL)
FP_BIN(fr, 1, add, OP_ADD) FP_BIN(fr, 1, sub, OP_SUB) FP_BIN(fr, 1, mul, OP_MUL)
#define FP_UN(field, FR, name, OP)                                                                          \
  int32_t sylow_hip_##field##_##name##_batch(const uint64_t* a, uint64_t* out, size_t n, void* stream) {      \
    ARGCHK(a && out); if (!n) return SYLOW_HIP_OK;                                                        \
    k_fp_unop<OP, FR><<<GRID((n + 1) / 2)>>>(a, out, n); LAUNCHED();                                         \
  }
FP_UN(fp, 0, sqr, OP_SQR) FP_UN(fp, 0, neg, OP_NEG) FP_UN(fp, 0, inv, OP_INV)
FP_UN(fr, 1, sqr, OP_SQR) FP_UN(fr, 1, neg, OP_NEG) FP_UN(fr, 1, inv, OP_INV)

int32_t sylow_hip_fp_pow_batch(const uint64_t* a, const uint64_t* e, uint64_t* out, size_t n, void* stream) {
  ARGCHK(a && e && out); if (!n) return SYLOW_HIP_OK; k_fp_pow<<<GRID(n)>>>(a, e, out, n); LAUNCHED();
}
int32_t sylow_hip_fp_sqrt_batch(const uint64_t* a, uint64_t* out, uint8_t* is_some, size_t n, void* stream) {
  ARGCHK(a && out && is_some); if (!n) return SYLOW_HIP_OK; k_fp_sqrt<<<GRID(n)>>>(a, out, is_some, nullptr, n); LAUNCHED();
}
int32_t sylow_hip_fp_is_square_batch(const uint64_t* a, uint8_t* flags, size_t n, void* stream) {
  ARGCHK(a && flags); if (!n) return SYLOW_HIP_OK; k_fp_sqrt<<<GRID(n)>>>(a, nullptr, nullptr, flags, n); LAUNCHED();
}
int32_t sylow_hip_fp2_mul_batch(const uint64_t* a, const uint64_t* b, uint64_t* out, size_t n, void* stream) {
  ARGCHK(a && b && out); if (!n) return SYLOW_HIP_OK; k_fp2_op<<<GRID(n)>>>(OP_MUL, a, b, out, n); LAUNCHED();
}
int32_t sylow_hip_fp2_sqr_batch(const uint64_t* a, uint64_t* out, size_t n, void* stream) {
  ARGCHK(a && out); if (!n) return SYLOW_HIP_OK; k_fp2_op<<<GRID(n)>>>(OP_SQR, a, nullptr, out, n); LAUNCHED();
}
int32_t sylow_hip_fp2_inv_batch(const uint64_t* a, uint64_t* out, size_t n, void* stream) {
  ARGCHK(a && out); if (!n) return SYLOW_HIP_OK; k_fp2_op<<<GRID(n)>>>(OP_INV, a, nullptr, out, n); LAUNCHED();
}
int32_t sylow_hip_fp6_mul_batch(const uint64_t* a, const uint64_t* b, uint64_t* out, size_t n, void* stream) {
  ARGCHK(a && b && out); if (!n) return SYLOW_HIP_OK; k_fp6_op<<<GRID(n)>>>(OP_MUL, a, b, out, n); LAUNCHED();
}
int32_t sylow_hip_fp6_inv_batch(const uint64_t* a, uint64_t* out, size_t n, void* stream) {
  ARGCHK(a && out); if (!n) return SYLOW_HIP_OK; k_fp6_op<<<GRID(n)>>>(OP_INV, a, nullptr, out, n); LAUNCHED();
}
int32_t sylow_hip_fp12_mul_batch(const uint64_t* a, const uint64_t* b, uint64_t* out, size_t n, void* stream) {
  ARGCHK(a && b && out); if (!n) return SYLOW_HIP_OK; k_fp12_op<<<GRID(n)>>>(OP12_MUL, a, b, out, n); LAUNCHED();
}
int32_t sylow_hip_fp12_sqr_batch(const uint64_t* a, uint64_t* out, size_t n, void* stream) {
  ARGCHK(a && out); if (!n) return SYLOW_HIP_OK; k_fp12_op<<<GRID(n)>>>(OP12_SQR, a, nullptr, out, n); LAUNCHED();
}
int32_t sylow_hip_fp12_inv_batch(const uint64_t* a, uint64_t* out, size_t n, void* stream) {
  ARGCHK(a && out); if (!n) return SYLOW_HIP_OK; k_fp12_op<<<GRID(n)>>>(OP12_INV, a, nullptr, out, n); LAUNCHED();
}
int32_t sylow_hip_fp12_frobenius_batch(const uint64_t* a, int32_t exponent, uint64_t* out, size_t n, void* stream) {
  ARGCHK(a && out && exponent >= 1 && exponent <= 3); if (!n) return SYLOW_HIP_OK;
  k_fp12_op<<<GRID(n)>>>(OP12_FROB1 + exponent - 1, a, nullptr, out, n); LAUNCHED();
}
int32_t sylow_hip_fp12_sparse_mul_batch(const uint64_t* f, const uint64_t* ell, uint64_t* out, size_t n, void* stream) {
  ARGCHK(f && ell && out); if (!n) return SYLOW_HIP_OK; k_fp12_op<<<GRID(n)>>>(OP12_SPARSE, f, ell, out, n); LAUNCHED();
}
// test hook (not in the public header's stable surface): Granger-Scott cyclotomic square
int32_t sylow_hip_fp12_cyclotomic_sqr_batch(const uint64_t* a, uint64_t* out, size_t n, void* stream) {
  ARGCHK(a && out); if (!n) return SYLOW_HIP_OK; k_fp12_op<<<GRID(n)>>>(OP12_CYCSQR, a, nullptr, out, n); LAUNCHED();
}
// test hook: raw k_fp12_op selector (8: product on the carry-free core, 9: cyclotomic square on it,
// 10 / 11: exp_by_neg_z on the carry-free / saturated core); 16..28: the lane-pair Fp12 layer (plk::k_w12_op)
int32_t sylow_hip_fp12_hook_batch(int32_t op, const uint64_t* a, const uint64_t* b, uint64_t* out, size_t n, void* stream) {
  ARGCHK(a && out && op >= 0 && (op <= 11 || (op >= 16 && op <= plk::OPW_LAST))); if (!n) return SYLOW_HIP_OK;
  if (op >= 16) { plk::k_w12_op<<<GRID(2 * n)>>>(op, a, b, out, n); LAUNCHED(); }
  k_fp12_op<<<GRID(n)>>>(op, a, b, out, n); LAUNCHED();
}

int32_t sylow_hip_g1_scalar_mul_batch(const uint64_t* p_xy, const uint8_t* p_inf, const uint64_t* k, uint64_t* out_xy, uint8_t* out_inf, size_t n, void* stream) {
  ARGCHK(p_xy && k && out_xy && out_inf); if (!n) return SYLOW_HIP_OK;
  k_g1_scalar_mul<<<GRID(n)>>>(p_xy, p_inf, k, out_xy, out_inf, n); LAUNCHED();
}
int32_t sylow_hip_g2_scalar_mul_batch(const uint64_t* p_xy, const uint8_t* p_inf, const uint64_t* k, uint64_t* out_xy, uint8_t* out_inf, size_t n, void* stream) {
  ARGCHK(p_xy && k && out_xy && out_inf); if (!n) return SYLOW_HIP_OK;
  if (single_lane()) { k_g2_scalar_mul<<<GRID(n)>>>(p_xy, p_inf, k, out_xy, out_inf, n); LAUNCHED(); }
  plk::k_g2_scalar_mul<<<GRID(2 * n)>>>(p_xy, p_inf, k, out_xy, out_inf, n); LAUNCHED();
}
int32_t sylow_hip_g1_add_batch(const uint64_t* a_xy, const uint8_t* a_inf, const uint64_t* b_xy, const uint8_t* b_inf, uint64_t* out_xy, uint8_t* out_inf, size_t n, void* stream) {
  ARGCHK(a_xy && b_xy && out_xy && out_inf); if (!n) return SYLOW_HIP_OK;
  k_g1_add<<<GRID(n)>>>(a_xy, a_inf, b_xy, b_inf, out_xy, out_inf, n); LAUNCHED();
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
int32_t sylow_hip_g2_normalize_batch(const uint64_t* p_xyz, uint64_t* out_xy, uint8_t* out_inf, size_t n, void* stream) {
  ARGCHK(p_xyz && out_xy && out_inf); if (!n) return SYLOW_HIP_OK;
  if (single_lane()) { k_g2_normalize<<<GRID(n)>>>(p_xyz, out_xy, out_inf, n); LAUNCHED(); }
  plk::k_g2_normalize<<<GRID(2 * n)>>>(p_xyz, out_xy, out_inf, n); LAUNCHED();
}
int32_t sylow_hip_g2_psi_batch(const uint64_t* q_xy, const uint8_t* q_inf, uint64_t* out_xy, uint8_t* out_inf, uint8_t* status, size_t n, void* stream) {
  ARGCHK(q_xy && out_xy && out_inf); if (!n) return SYLOW_HIP_OK;
  plk::k_g2_psi<<<GRID(2 * n)>>>(q_xy, q_inf, out_xy, out_inf, status, n); LAUNCHED();
}
int32_t sylow_hip_g2_subgroup_check_batch(const uint64_t* q_xy, const uint8_t* q_inf, uint8_t* status, size_t n, void* stream) {
  ARGCHK(q_xy && status); if (!n) return SYLOW_HIP_OK;
  if (single_lane()) { k_g2_subgroup_check<<<GRID(n)>>>(q_xy, q_inf, status, n); LAUNCHED(); }
  plk::k_g2_subgroup_check<<<GRID(2 * n)>>>(q_xy, q_inf, status, n); LAUNCHED();
}
int32_t sylow_hip_miller_loop_batch(const uint64_t* p_xy, const uint64_t* q_xy, uint64_t* f_out, size_t n, void* stream) {
  ARGCHK(p_xy && q_xy && f_out); if (!n) return SYLOW_HIP_OK;
  if (single_lane()) { k_miller_loop<<<GRID(n)>>>(p_xy, q_xy, f_out, n); LAUNCHED(); }
  plk::k_miller_loop<<<GRID(2 * n)>>>(p_xy, q_xy, f_out, n); LAUNCHED();
}
int32_t sylow_hip_final_exp_batch(const uint64_t* f, uint64_t* gt_out, size_t n, void* stream) {
  ARGCHK(f && gt_out); if (!n) return SYLOW_HIP_OK;
  if (single_lane()) { k_final_exp<<<GRID(n)>>>(f, gt_out, n); LAUNCHED(); }
  plk::k_final_exp<<<GRID(2 * n)>>>(f, gt_out, n); LAUNCHED();
}
int32_t sylow_hip_pairing_batch(const uint64_t* p_xy, const uint8_t* p_inf, const uint64_t* q_xy, const uint8_t* q_inf, uint64_t* gt_out, size_t n, void* stream) {
  ARGCHK(p_xy && q_xy && gt_out); if (!n) return SYLOW_HIP_OK;
  if (single_lane()) { k_pairing<<<GRID(n)>>>(p_xy, p_inf, q_xy, q_inf, gt_out, n); LAUNCHED(); }
  plk::k_pairing<<<GRID(2 * n)>>>(p_xy, p_inf, q_xy, q_inf, gt_out, n); LAUNCHED();
}

int32_t sylow_hip_multi_pairing_batch(const uint64_t* p_xy, const uint8_t* p_inf, const uint64_t* q_xy, const uint8_t* q_inf,
                                      const uint64_t* pair_offsets, size_t n_jobs, size_t n_pairs, int32_t skip_infinity,
                                      uint64_t* gt_out, uint8_t* is_one, void* stream) {
  ARGCHK(pair_offsets && (gt_out || is_one) && (n_pairs == 0 || (p_xy && q_xy))); if (!n_jobs) return SYLOW_HIP_OK;
  if (single_lane()) { k_multi_pairing<<<GRID(n_jobs)>>>(p_xy, p_inf, q_xy, q_inf, pair_offsets, n_jobs, n_pairs, skip_infinity, gt_out, is_one); LAUNCHED(); }
  plk::k_multi_pairing<plk::KMAXW><<<GRID(2 * n_jobs)>>>(p_xy, p_inf, q_xy, q_inf, pair_offsets, n_jobs, n_pairs, skip_infinity, gt_out, is_one, 0); LAUNCHED();
}
int32_t sylow_hip_glued_miller_loop_batch(const uint64_t* p_xy, const uint64_t* q_xy, const uint64_t* pair_offsets, size_t n_jobs, size_t n_pairs,
                                           uint64_t* f_out, void* stream) {
  ARGCHK(pair_offsets && f_out && (n_pairs == 0 || (p_xy && q_xy))); if (!n_jobs) return SYLOW_HIP_OK;
  plk::k_multi_pairing<plk::KMAXW><<<GRID(2 * n_jobs)>>>(p_xy, nullptr, q_xy, nullptr, pair_offsets, n_jobs, n_pairs, 0, f_out, nullptr, 1); LAUNCHED();
}
int32_t sylow_hip_pairing_product_batch(const uint64_t* p_xy, const uint8_t* p_inf, const uint64_t* q_xy, const uint8_t* q_inf,
                                        size_t n_pairs, int32_t skip_infinity, uint64_t* gt_out, uint8_t* is_one, void* stream) {
  ARGCHK((gt_out || is_one) && (n_pairs == 0 || (p_xy && q_xy)));
  hipStream_t st = (hipStream_t)stream;
  const size_t n_jobs = (n_pairs + plk::KPROD - 1) / plk::KPROD;
  if (n_jobs == 0) {
    plk::k_final_exp_flag<<<1, 64, 0, st>>>(nullptr, 0, gt_out, is_one); LAUNCHED();
  }
  // workspace: chunk offsets + two ping-pong buffers of Fp12 values
  const size_t n_off = (n_jobs + 2) & ~(size_t)1, n_a = 48 * n_jobs, n_b = 48 * ((n_jobs + 1) / 2);
  void* base = nullptr;
  int32_t rc = workspace(&base, (n_off + n_a + n_b) * sizeof(u64), st);
  if (rc != SYLOW_HIP_OK) return rc;
  u64 *off = (u64*)base, *bufa = off + n_off, *bufb = bufa + n_a;
  plk::k_chunk_offsets<<<GRID(n_jobs + 1)>>>(off, n_jobs, n_pairs);
  plk::k_multi_pairing<plk::KPROD><<<GRID(2 * n_jobs)>>>(p_xy, p_inf, q_xy, q_inf, off, n_jobs, n_pairs, skip_infinity, bufa, nullptr, 1);
  u64 *cur = bufa, *nxt = bufb;
  size_t m = n_jobs;
  while (m > 1) {
    const size_t h = (m + 1) / 2;
    plk::k_fp12_tree_level<<<GRID(2 * h)>>>(cur, m, nxt, h);
    u64* tmp = cur; cur = nxt; nxt = tmp;
    m = h;
  }
  plk::k_final_exp_flag<<<1, 64, 0, st>>>(cur, 1, gt_out, is_one);
  LAUNCHED();
}
static const uint8_t SYLOW_DST[] = "WARLOCK-CHAOS-V01-CS01-SHA-256";   // lib.rs:90 (30 bytes)
static void dst_arg(DstPrime& dp, const uint8_t* dst, size_t len) {
  if (!dst) { dst = SYLOW_DST; len = 30; }
  make_dst_prime(dp, dst, len);
}
int32_t sylow_hip_hash_to_g1_batch(const uint8_t* msgs, const uint64_t* msg_offsets, const uint8_t* dst_host, size_t dst_len,
                                   uint64_t* out_xy, uint8_t* out_inf, size_t n, void* stream) {
  ARGCHK(msgs && msg_offsets && out_xy && out_inf); if (!n) return SYLOW_HIP_OK;
  DstPrime dp; dst_arg(dp, dst_host, dst_len);
  k_hash_to_g1<<<GRID(n)>>>(msgs, msg_offsets, dp, out_xy, out_inf, nullptr, n); LAUNCHED();
}
int32_t sylow_hip_hash_to_field_batch(const uint8_t* msgs, const uint64_t* msg_offsets, const uint8_t* dst_host, size_t dst_len,
                                      uint64_t* out_u, size_t n, void* stream) {
  ARGCHK(msgs && msg_offsets && out_u); if (!n) return SYLOW_HIP_OK;
  DstPrime dp; dst_arg(dp, dst_host, dst_len);
  k_hash_to_field<<<GRID(n)>>>(msgs, msg_offsets, dp, out_u, n); LAUNCHED();
}
int32_t sylow_hip_bls_sign_batch(const uint64_t* sk, const uint8_t* msgs, const uint64_t* msg_offsets,
                                 uint64_t* sig_xy, uint8_t* sig_inf, size_t n, void* stream) {
  ARGCHK(sk && msgs && msg_offsets && sig_xy && sig_inf); if (!n) return SYLOW_HIP_OK;
  DstPrime dp; dst_arg(dp, nullptr, 0);
  k_bls_sign<<<GRID(n)>>>(sk, msgs, msg_offsets, dp, sig_xy, sig_inf, n); LAUNCHED();
}
// Per-device scratch workspace for the few entry points that need temporaries (line table of one key, product tree).
// Grow-only, plain hipMalloc (the stream-ordered allocator on the legacy default stream produced intermittently wrong
// products here -- measured, see tools/dbg_prod.py -- so it is not used).  Calls that use the workspace are serialised
// against each other: a call on another stream first waits for the previous user's stream.
static int32_t workspace(void** out, size_t bytes, hipStream_t st) {
  struct Ws { void* p; size_t cap; hipStream_t last; bool used; };
  static Ws ws[64] = {};
  static std::mutex mu;
  std::lock_guard<std::mutex> lock(mu);     // bookkeeping only: two host threads must still not run workspace users concurrently
  int dev = 0;
  HIPCHK(hipGetDevice(&dev));
  if (dev < 0 || dev >= 64) { snprintf(g_err, sizeof(g_err), "device index out of range"); return SYLOW_HIP_E_ARG; }
  Ws& w = ws[dev];
  if (w.used && w.last != st) HIPCHK(hipStreamSynchronize(w.last));
  if (bytes > w.cap) {
    if (w.p) { HIPCHK(hipDeviceSynchronize()); HIPCHK(hipFree(w.p)); w.p = nullptr; w.cap = 0; }
    const size_t cap = bytes + (bytes >> 2) + 4096;
    HIPCHK(hipMalloc(&w.p, cap));
    w.cap = cap;
  }
  w.last = st; w.used = true;
  *out = w.p;
  return SYLOW_HIP_OK;
}
// The pairing-based entry points run on lane pairs (pair_kernels.hpp).  SYLOW_HIP_SINGLE_LANE=1 selects the one-element-per-lane
// kernels instead: the slower twin kept for A/B measurements and as a second implementation for the parity tests.
static bool single_lane() {
  static int v = -1;
  if (v < 0) { const char* e = getenv("SYLOW_HIP_SINGLE_LANE"); v = (e && e[0] == '1') ? 1 : 0; }
  return v == 1;
}
static int32_t ensure_g2gen_lines29(void* stream) {
  static bool ready[64] = {false};
  int dev = 0;
  HIPCHK(hipGetDevice(&dev));
  if (dev < 0 || dev >= 64) { snprintf(g_err, sizeof(g_err), "device index out of range"); return SYLOW_HIP_E_ARG; }
  if (!ready[dev]) {
    plk::k_g2_lines29<<<1, 64, 0, (hipStream_t)stream>>>(nullptr, 0, 0, nullptr);
    hipError_t e_ = hipGetLastError();
    if (e_ != hipSuccess) return fail(e_, "k_g2_lines29 launch");
    HIPCHK(hipStreamSynchronize((hipStream_t)stream));
    ready[dev] = true;
  }
  return SYLOW_HIP_OK;
}
static int32_t gen_table29(const bn254::i32** out) {
  void* p = nullptr;
  HIPCHK(hipGetSymbolAddress(&p, HIP_SYMBOL(plk::g_g2gen_lines29)));
  *out = (const bn254::i32*)p;
  return SYLOW_HIP_OK;
}
// one-time (per device) construction of the G2-generator line table used by the fused verifier
static int32_t ensure_g2gen_lines(void* stream) {
  static bool ready[64] = {false};
  int dev = 0;
  HIPCHK(hipGetDevice(&dev));
  if (dev < 0 || dev >= 64) { snprintf(g_err, sizeof(g_err), "device index out of range"); return SYLOW_HIP_E_ARG; }
  if (!ready[dev]) {
    k_g2_lines<<<1, 64, 0, (hipStream_t)stream>>>(nullptr, 0, 0, nullptr);
    hipError_t e_ = hipGetLastError();
    if (e_ != hipSuccess) return fail(e_, "k_g2_lines launch");
    // one-time: later calls may run on other streams, so the table must be complete before we report it ready
    HIPCHK(hipStreamSynchronize((hipStream_t)stream));
    ready[dev] = true;
  }
  return SYLOW_HIP_OK;
}
int32_t sylow_hip_bls_verify_batch(const uint64_t* pk_xy, const uint8_t* pk_inf, const uint8_t* msgs, const uint64_t* msg_offsets,
                                   const uint64_t* sig_xy, const uint8_t* sig_inf, uint8_t* ok, size_t n, void* stream) {
  ARGCHK(pk_xy && msgs && msg_offsets && sig_xy && ok); if (!n) return SYLOW_HIP_OK;
  DstPrime dp; dst_arg(dp, nullptr, 0);
  if (single_lane()) {
    int32_t rc = ensure_g2gen_lines(stream);
    if (rc != SYLOW_HIP_OK) return rc;
    k_bls_verify<<<GRID(n)>>>(pk_xy, pk_inf, msgs, msg_offsets, dp, sig_xy, sig_inf, ok, n); LAUNCHED();
  }
  int32_t rc = ensure_g2gen_lines29(stream);
  if (rc != SYLOW_HIP_OK) return rc;
  const bn254::i32* gen = nullptr;
  if ((rc = gen_table29(&gen)) != SYLOW_HIP_OK) return rc;
  plk::k_bls_verify<<<GRID(2 * n)>>>(pk_xy, pk_inf, msgs, msg_offsets, dp, sig_xy, sig_inf, gen, ok, n); LAUNCHED();
}

int32_t sylow_hip_bls_verify_fused_batch(const uint64_t* pk_xy, const uint8_t* pk_inf, const uint8_t* msgs, const uint64_t* msg_offsets,
                                         const uint64_t* sig_xy, const uint8_t* sig_inf, uint8_t* ok, size_t n, void* stream) {
  ARGCHK(pk_xy && msgs && msg_offsets && sig_xy && ok); if (!n) return SYLOW_HIP_OK;
  DstPrime dp; dst_arg(dp, nullptr, 0);
  if (single_lane()) {
    int32_t rc = ensure_g2gen_lines(stream);
    if (rc != SYLOW_HIP_OK) return rc;
    k_bls_verify_fused<false><<<GRID(n)>>>(pk_xy, pk_inf, nullptr, msgs, msg_offsets, dp, sig_xy, sig_inf, ok, n); LAUNCHED();
  }
  int32_t rc = ensure_g2gen_lines29(stream);
  if (rc != SYLOW_HIP_OK) return rc;
  const bn254::i32* gen = nullptr;
  if ((rc = gen_table29(&gen)) != SYLOW_HIP_OK) return rc;
  plk::k_bls_verify_fused<false><<<GRID(2 * n)>>>(pk_xy, pk_inf, nullptr, msgs, msg_offsets, dp, sig_xy, sig_inf, gen, ok, n); LAUNCHED();
}
int32_t sylow_hip_bls_verify_same_signer_batch(const uint64_t* pk_xy, const uint8_t* pk_inf, const uint8_t* msgs, const uint64_t* msg_offsets,
                                               const uint64_t* sig_xy, const uint8_t* sig_inf, uint8_t* ok, size_t n, void* stream) {
  ARGCHK(pk_xy && msgs && msg_offsets && sig_xy && ok); if (!n) return SYLOW_HIP_OK;
  hipStream_t st = (hipStream_t)stream;
  DstPrime dp; dst_arg(dp, nullptr, 0);
  if (single_lane()) {
    int32_t rc = ensure_g2gen_lines(stream);
    if (rc != SYLOW_HIP_OK) return rc;
    void* wsp = nullptr;
    if ((rc = workspace(&wsp, 87 * 48 * sizeof(u32), st)) != SYLOW_HIP_OK) return rc;
    u32* table = (u32*)wsp;
    k_g2_lines<<<1, 64, 0, st>>>(pk_xy, 1, 0, table);          // the key is a 1-element SoA array
    k_bls_verify_fused<true><<<GRID(n)>>>(pk_xy, pk_inf, table, msgs, msg_offsets, dp, sig_xy, sig_inf, ok, n);
    LAUNCHED();
  }
  int32_t rc = ensure_g2gen_lines29(stream);
  if (rc != SYLOW_HIP_OK) return rc;
  const bn254::i32* gen = nullptr;
  if ((rc = gen_table29(&gen)) != SYLOW_HIP_OK) return rc;
  void* wsp = nullptr;
  if ((rc = workspace(&wsp, plk::LINE_TABLE_WORDS * sizeof(bn254::i32), st)) != SYLOW_HIP_OK) return rc;
  bn254::i32* table = (bn254::i32*)wsp;
  plk::k_g2_lines29<<<1, 64, 0, st>>>(pk_xy, 1, 0, table);     // the key is a 1-element SoA array
  plk::k_bls_verify_fused<true><<<GRID(2 * n)>>>(pk_xy, pk_inf, table, msgs, msg_offsets, dp, sig_xy, sig_inf, gen, ok, n);
  LAUNCHED();
}
int32_t sylow_hip_g2_precompute_batch(const uint64_t* q_xy, uint64_t* coeffs, size_t n, void* stream) {
  ARGCHK(q_xy && coeffs); if (!n) return SYLOW_HIP_OK;
  k_g2_precompute<<<GRID(n)>>>(q_xy, coeffs, n); LAUNCHED();
}

int32_t sylow_hip_evm_ecadd_batch(const uint8_t* in, uint8_t* out, uint8_t* status, size_t n, void* stream) {
  ARGCHK(in && out && status); if (!n) return SYLOW_HIP_OK;
  k_evm_ecadd<<<GRID(n)>>>(in, out, status, n); LAUNCHED();
}
int32_t sylow_hip_evm_ecmul_batch(const uint8_t* in, uint8_t* out, uint8_t* status, size_t n, void* stream) {
  ARGCHK(in && out && status); if (!n) return SYLOW_HIP_OK;
  k_evm_ecmul<<<GRID(n)>>>(in, out, status, n); LAUNCHED();
}
int32_t sylow_hip_evm_ecpairing_batch(const uint8_t* in, const uint64_t* pair_offsets, size_t n_jobs, size_t n_pairs,
                                      uint8_t* result, uint8_t* status, void* stream) {
  ARGCHK(pair_offsets && result && status && (in || !n_pairs)); if (!n_jobs) return SYLOW_HIP_OK;
  hipStream_t st = (hipStream_t)stream;
  const size_t np = n_pairs ? n_pairs : 1;
  // workspace: decoded SoA points, flags, per-pair status, per-job product flag
  const size_t bytes = np * (8 + 16) * 8 + 3 * np + n_jobs + 64;
  void* wsp = nullptr;
  int32_t rc = workspace(&wsp, bytes, st);
  if (rc != SYLOW_HIP_OK) return rc;
  uint8_t* ws = (uint8_t*)wsp;
  u64* pxy = (u64*)ws;
  u64* qxy = pxy + 8 * np;
  uint8_t* pinf = (uint8_t*)(qxy + 16 * np);
  uint8_t* qinf = pinf + np;
  uint8_t* pst = qinf + np;
  uint8_t* isone = pst + np;
  if (n_pairs) k_evm_decode_pairs<<<dim3((unsigned)((2 * n_pairs + BLOCK - 1) / BLOCK)), dim3(BLOCK), 0, st>>>(in, n_pairs, pxy, pinf, qxy, qinf, pst);
  if (single_lane()) k_multi_pairing<<<GRID(n_jobs)>>>(pxy, pinf, qxy, qinf, pair_offsets, n_jobs, n_pairs, /*skip_infinity=*/1, nullptr, isone);
  else plk::k_multi_pairing<plk::KMAXW><<<GRID(2 * n_jobs)>>>(pxy, pinf, qxy, qinf, pair_offsets, n_jobs, n_pairs, /*skip_infinity=*/1, nullptr, isone, 0);
  k_evm_pair_finalize<<<GRID(n_jobs)>>>(pst, pair_offsets, n_jobs, isone, result, status);
  LAUNCHED();
}

int32_t sylow_hip_g1_to_be_bytes_batch(const uint64_t* p_xy, const uint8_t* p_inf, uint8_t* out, size_t n, void* stream) {
  ARGCHK(p_xy && out); if (!n) return SYLOW_HIP_OK; k_g1_to_bytes<<<GRID(n)>>>(p_xy, p_inf, out, n); LAUNCHED();
}
int32_t sylow_hip_g1_from_be_bytes_batch(const uint8_t* in, uint64_t* out_xy, uint8_t* out_inf, uint8_t* status, size_t n, void* stream) {
  ARGCHK(in && out_xy && out_inf && status); if (!n) return SYLOW_HIP_OK; k_g1_from_bytes<<<GRID(n)>>>(in, out_xy, out_inf, status, n); LAUNCHED();
}
int32_t sylow_hip_g2_to_be_bytes_batch(const uint64_t* p_xy, const uint8_t* p_inf, uint8_t* out, size_t n, void* stream) {
  ARGCHK(p_xy && out); if (!n) return SYLOW_HIP_OK; k_g2_to_bytes<<<GRID(n)>>>(p_xy, p_inf, out, n); LAUNCHED();
}
int32_t sylow_hip_g2_from_be_bytes_batch(const uint8_t* in, uint64_t* out_xy, uint8_t* out_inf, uint8_t* status, size_t n, void* stream) {
  ARGCHK(in && out_xy && out_inf && status); if (!n) return SYLOW_HIP_OK; k_g2_from_bytes<<<GRID(2 * n)>>>(in, out_xy, out_inf, status, n); LAUNCHED();
}

// test hook (see k_f29_hook)
int32_t sylow_hip_f29_hook_batch(int32_t op, const uint64_t* a, const uint64_t* b, uint64_t* out, size_t n, void* stream) {
  ARGCHK(a && b && out); if (!n) return SYLOW_HIP_OK;
  k_f29_hook<<<GRID(n)>>>(op, a, b, out, n); LAUNCHED();
}

int32_t sylow_hip_gt_pow_batch(const uint64_t* gt, const uint64_t* k, uint64_t* out, size_t n, void* stream) {
  ARGCHK(gt && k && out); if (!n) return SYLOW_HIP_OK;
  if (single_lane()) { k_gt_pow<<<GRID(n)>>>(gt, k, out, n); LAUNCHED(); }
  plk::k_gt_pow<<<GRID(2 * n)>>>(gt, k, out, n); LAUNCHED();
}
int32_t sylow_hip_g2_add_batch(const uint64_t* a_xy, const uint8_t* a_inf, const uint64_t* b_xy, const uint8_t* b_inf, uint64_t* out_xy, uint8_t* out_inf, size_t n, void* stream) {
  ARGCHK(a_xy && b_xy && out_xy && out_inf); if (!n) return SYLOW_HIP_OK;
  if (single_lane()) { k_g2_add<<<GRID(n)>>>(a_xy, a_inf, b_xy, b_inf, out_xy, out_inf, n); LAUNCHED(); }
  plk::k_g2_add<<<GRID(2 * n)>>>(a_xy, a_inf, b_xy, b_inf, out_xy, out_inf, n); LAUNCHED();
}
int32_t sylow_hip_g1_double_batch(const uint64_t* a_xy, const uint8_t* a_inf, uint64_t* out_xy, uint8_t* out_inf, size_t n, void* stream) {
  ARGCHK(a_xy && out_xy && out_inf); if (!n) return SYLOW_HIP_OK; k_g1_double<<<GRID(n)>>>(a_xy, a_inf, out_xy, out_inf, n); LAUNCHED();
}
int32_t sylow_hip_g2_double_batch(const uint64_t* a_xy, const uint8_t* a_inf, uint64_t* out_xy, uint8_t* out_inf, size_t n, void* stream) {
  ARGCHK(a_xy && out_xy && out_inf); if (!n) return SYLOW_HIP_OK; if (single_lane()) { k_g2_double<<<GRID(n)>>>(a_xy, a_inf, out_xy, out_inf, n); LAUNCHED(); } plk::k_g2_double<<<GRID(2 * n)>>>(a_xy, a_inf, out_xy, out_inf, n); LAUNCHED();
}
int32_t sylow_hip_flags_all(const uint8_t* flags, size_t n, int32_t* out_dev, void* stream) {
  ARGCHK(out_dev && (flags || !n));
  HIPCHK(hipMemsetD32Async((hipDeviceptr_t)out_dev, 1, 1, (hipStream_t)stream));
  if (!n) return SYLOW_HIP_OK;
  k_flags_all<<<GRID(n)>>>(flags, n, out_dev); LAUNCHED();
}

}  // extern "C"
