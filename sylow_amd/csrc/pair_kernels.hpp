// pair_kernels.hpp -- the pairing / verification kernels on LANE PAIRS (bn254_pair.hpp, bn254_pair29.hpp).
// Included by sylow_hip.hip after the SoA helpers.  Thread t handles coordinate (t & 1) of element t >> 1, so a launch
// covers 2 n threads and a wavefront carries 32 elements.  Both lanes of a pair always take the same branches.
#pragma once
#include "bn254_pair29.hpp"

namespace plk {
using namespace bn254;
using namespace bn254::pl;

BN_DEV S2 load_s2(const u64* base, size_t n, size_t i, int w0, int odd) { return S2{load_fp(base, n, i, w0 + 4 * odd)}; }
BN_DEV void store_s2(u64* base, size_t n, size_t i, int w0, int odd, const S2& a) { store_fp(base, n, i, w0 + 4 * odd, a.c); }
BN_DEV void load_s12(S12& r, const u64* base, size_t n, size_t i, int odd) {
  r.c0.c0 = load_s2(base, n, i, 0, odd); r.c0.c1 = load_s2(base, n, i, 8, odd); r.c0.c2 = load_s2(base, n, i, 16, odd);
  r.c1.c0 = load_s2(base, n, i, 24, odd); r.c1.c1 = load_s2(base, n, i, 32, odd); r.c1.c2 = load_s2(base, n, i, 40, odd);
}
BN_DEV void store_s12(u64* base, size_t n, size_t i, int odd, const S12& a) {
  store_s2(base, n, i, 0, odd, a.c0.c0); store_s2(base, n, i, 8, odd, a.c0.c1); store_s2(base, n, i, 16, odd, a.c0.c2);
  store_s2(base, n, i, 24, odd, a.c1.c0); store_s2(base, n, i, 32, odd, a.c1.c1); store_s2(base, n, i, 40, odd, a.c1.c2);
}
BN_DEV S2 s2_g2gen_x() { return S2{sel(lane_odd(), fp_const(C_G2_GEN[0]), fp_const(C_G2_GEN[1]))}; }
BN_DEV S2 s2_g2gen_y() { return S2{sel(lane_odd(), fp_const(C_G2_GEN[2]), fp_const(C_G2_GEN[3]))}; }
BN_DEV W2 w2_select(const W2& a, const W2& b, bool c) { return W2{sel9(c, a.c, b.c)}; }     // c ? b : a

// ------------------------------------------------------------------ pairing(), Miller loop, final exponentiation ------
__global__ void HEAVY_BOUNDS k_miller_loop(const u64* pxy, const u64* qxy, u64* fout, size_t n) {
  const size_t t = TID, i = t >> 1;
  const int odd = (int)(t & 1);
  if (i >= n) return;
  const Fp px = load_fp(pxy, n, i, 0), py = load_fp(pxy, n, i, 4);
  const S2 qx = load_s2(qxy, n, i, 0, odd), qy = load_s2(qxy, n, i, 8, odd);
  S12 f;
  miller_loop29g(f, px, py, qx, qy);
  store_s12(fout, n, i, odd, f);
}
__global__ void HEAVY_BOUNDS k_final_exp(const u64* fin, u64* gout, size_t n) {
  const size_t t = TID, i = t >> 1;
  const int odd = (int)(t & 1);
  if (i >= n) return;
  S12 f, g;
  load_s12(f, fin, n, i, odd);
  final_exponentiation29(g, f);
  store_s12(gout, n, i, odd, g);
}
// pairing.rs:870-893
__global__ void HEAVY_BOUNDS k_pairing(const u64* pxy, const uint8_t* pinf, const u64* qxy, const uint8_t* qinf, u64* gout, size_t n) {
  const size_t t = TID, i = t >> 1;
  const int odd = (int)(t & 1);
  if (i >= n) return;
  const bool either_zero = (pinf && pinf[i]) || (qinf && qinf[i]);
  S12 g;
  if (either_zero) {
    g = s12_one();                 // Miller value forced to one; final_exponentiation(1) == 1
  } else {
    const Fp px = load_fp(pxy, n, i, 0), py = load_fp(pxy, n, i, 4);
    const S2 qx = load_s2(qxy, n, i, 0, odd), qy = load_s2(qxy, n, i, 8, odd);
    S12 f;
    miller_loop29g(f, px, py, qx, qy);
    final_exponentiation29(g, f);
  }
  store_s12(gout, n, i, odd, g);
}

// test hook: the lane-pair Fp12 layer one operation at a time (ops 16.. of sylow_hip_fp12_hook_batch); `b` carries the second
// operand, or the three line coefficients (ell_0, ell_vw, ell_vv) in its first 24 words for the sparse product
enum { OPW_MUL = 16, OPW_SQR = 17, OPW_SPARSE = 18, OPW_CYCSQR = 19, OPW_FROB1 = 20, OPW_FROB2 = 21, OPW_FROB3 = 22, OPW_EXPZ = 23,
       OPW_S_MUL = 24, OPW_S_SQR = 25, OPW_S_INV = 26, OPW_S_CYCSQR = 27, OPW_CONJ = 28, OPW_LAST = 28 };
__global__ void HEAVY_BOUNDS k_w12_op(int op, const u64* a, const u64* b, u64* out, size_t n) {
  const size_t t = TID, i = t >> 1;
  const int odd = (int)(t & 1);
  if (i >= n) return;
  S12 sx, sy, sr;
  load_s12(sx, a, n, i, odd);
  if (b) load_s12(sy, b, n, i, odd);
  if (op >= OPW_S_MUL && op <= OPW_S_CYCSQR) {       // saturated lane-pair layer (bn254_pair.hpp)
    if (op == OPW_S_MUL) sr = s12_mul(sx, sy);
    else if (op == OPW_S_SQR) sr = s12_sqr(sx);
    else if (op == OPW_S_INV) sr = s12_inv(sx);
    else sr = cyclotomic_sqr(sx);
  } else {
    W12 x, y, r;
    w12_from_s12(x, sx);
    if (b) w12_from_s12(y, sy);
    switch (op) {
      case OPW_MUL: w12_mul_nl(r, x, y); break;
      case OPW_SQR: r = w12_sqr(x); break;
      case OPW_SPARSE: r = w12_sparse_mul(x, y.c0.c0, y.c0.c1, y.c0.c2); break;
      case OPW_CYCSQR: w12_cyclotomic_sqr_nl(r, x); break;
      case OPW_FROB1: w12_frobenius_nl<1>(r, x); break;
      case OPW_FROB2: w12_frobenius_nl<2>(r, x); break;
      case OPW_FROB3: w12_frobenius_nl<3>(r, x); break;
      case OPW_CONJ: r = w12_conj(x); break;
      default: exp_by_neg_z29(r, x); break;
    }
    w12_to_s12(sr, r);
  }
  store_s12(out, n, i, odd, sr);
}

// ------------------------------------------------------------------ glued pairing ----------------------------------------
// Same wave-uniform schedule as the single-lane k_multi_pairing (sylow_hip.hip): chunks of KMAX pairs share the squarings of
// one accumulator, a lane pair whose job has fewer pairs multiplies by the unit line.  Pair states live in the stack frame
// (they are touched once per loop iteration); the accumulator and the working point stay in registers.
struct PairStateW { G2W r; W2 qx, qy; S2 qxs, qys; F29 px, py; bool qinf, live; };
constexpr int KMAXW = 4;        // ecPairing / glued jobs (a few pairs each)
constexpr int KPROD = 8;        // batch-wide product: more pairs per shared squaring

template <int KMAX>
__global__ void HEAVY_BOUNDS k_multi_pairing(const u64* pxy, const uint8_t* pinf, const u64* qxy, const uint8_t* qinf,
                                             const u64* offsets, size_t n_jobs, size_t n_pairs, int skip_infinity,
                                             u64* gout, uint8_t* is_one, int raw_miller) {
  const size_t t = TID, job = t >> 1;
  const int odd = (int)(t & 1);
  const bool active = job < n_jobs;           // no early return: every lane takes part in the wave reductions
  size_t next = active ? offsets[job] : 0, hi = active ? offsets[job + 1] : 0;
  const W2 w_one = w2_from_s2(s2_one()), w_zero = W2{F29{{0, 0, 0, 0, 0, 0, 0, 0, 0}}};
  const W2 twist_b = w2_const(C_TWIST_B);
  W12 acc;
  {
    S12 one = s12_one();
    w12_from_s12(acc, one);
  }
  PairStateW st[KMAX];
  const u64 nz = BN_ATE_NAF_NZ, ng = BN_ATE_NAF_NEG;
#pragma unroll 1
  while (wave_max(next < hi ? 1 : 0)) {
    int k = 0;
#pragma unroll 1
    for (int slot = 0; slot < KMAX; ++slot) {
      bool have = false;
      size_t idx = 0;
      while (next < hi) {
        bool pi = pinf && pinf[next], qi = qinf && qinf[next];
        idx = next++;
        if (!(skip_infinity && (pi || qi))) { have = true; break; }   // EIP-197: identity pairs contribute 1
      }
      PairStateW& s = st[slot];
      s.live = have;
      const size_t src = have ? idx : 0;
      const bool qi = have && qinf && qinf[src];
      const bool ld = have && n_pairs != 0;
      s.px = f29_reduce(f29_from_fp(ld ? load_fp(pxy, n_pairs, src, 0) : fp_one()));
      s.py = f29_reduce(f29_from_fp(ld ? load_fp(pxy, n_pairs, src, 4) : fp_one()));
      s.qxs = ld ? load_s2(qxy, n_pairs, src, 0, odd) : s2_g2gen_x();
      s.qys = ld ? load_s2(qxy, n_pairs, src, 8, odd) : s2_g2gen_y();
      s.qx = w2_from_s2(s.qxs);
      s.qy = w2_from_s2(s.qys);
      s.qinf = qi;
      // G2Projective::from(&G2Affine): Z = infinity ? 0 : 1 (group.rs:506-517); the glued loop never looks at the flag
      // again in replay mode
      s.r = G2W{s.qx, s.qy, qi ? w_zero : w_one};
      if (have) k = slot + 1;
    }
    const int kw = wave_max(k);
    if (kw == 0) continue;
    W12 f;
    {
      S12 one = s12_one();
      w12_from_s12(f, one);
    }
    W2 l0, l1, l2;
    auto apply = [&](const PairStateW& s) {          // f *= line, or *= 1 for a dead slot
      const bool lv = s.live;
      f = w12_sparse_mul(f, w2_select(w_one, l0, lv), w2_select(w_zero, w2_scale(l1, s.py), lv), w2_select(w_zero, w2_scale(l2, s.px), lv));
    };
#pragma unroll 1
    for (int i = 0; i < 64; ++i) {
      f = w12_sqr(f);
#pragma unroll 1
      for (int j = 0; j < kw; ++j) { g2_doubling_step29(st[j].r, l0, l1, l2, twist_b); apply(st[j]); }
      if ((nz >> (63 - i)) & 1) {
        const bool neg = (ng >> (63 - i)) & 1;
#pragma unroll 1
        for (int j = 0; j < kw; ++j) {
          const W2 by = neg ? w2_neg(st[j].qy) : st[j].qy;
          g2_addition_step29(st[j].r, st[j].qx, by, l0, l1, l2);
          apply(st[j]);
        }
      }
    }
    // the two Frobenius additions; endomorphism() returns self for the identity (g2.rs:141-143)
#pragma unroll 1
    for (int step = 0; step < 2; ++step) {
#pragma unroll 1
      for (int j = 0; j < kw; ++j) {
        S2 q1x, q1y, q2x, q2y;
        g2_psi_affine(q1x, q1y, st[j].qxs, st[j].qys);
        g2_psi_affine(q2x, q2y, q1x, q1y);
        const bool qi = st[j].qinf;
        q1x = s2_select(q1x, st[j].qxs, qi); q1y = s2_select(q1y, st[j].qys, qi);
        q2x = s2_select(q2x, st[j].qxs, qi); q2y = s2_select(q2y, st[j].qys, qi);
        if (step == 0) g2_addition_step29(st[j].r, w2_from_s2(q1x), w2_from_s2(q1y), l0, l1, l2);
        else g2_addition_step29(st[j].r, w2_from_s2(q2x), w2_from_s2(s2_neg(q2y)), l0, l1, l2);
        apply(st[j]);
      }
    }
    w12_mul_nl(acc, acc, f);
  }
  S12 fin, g;
  w12_to_s12(fin, acc);
  if (raw_miller) g = fin;                     // partial Miller product for the batch-wide reduction below
  else final_exponentiation29(g, fin);
  if (active) {
    if (gout) store_s12(gout, n_jobs, job, odd, g);
    const bool one = s12_is_one(g);
    if (is_one && !odd) is_one[job] = one ? 1 : 0;
  }
}

// ------------------------------------------------------------------ Gt * Fr ---------------------------------------------------
// Mul<&Fr> for &Gt (gt.rs:161-187): the reference's 256-step signed-digit square-and-multiply on generic Fp12 squares and
// products (exact for any input), negative digits multiply by the conjugate.  Wave-uniform: every step squares; a step
// multiplies when any lane of the wavefront has a non-zero digit, lanes with a zero digit multiply by one.
__global__ void HEAVY_BOUNDS k_gt_pow(const u64* g, const u64* ks, u64* out, size_t n) {
  const size_t t = TID, i = t >> 1;
  const int odd = (int)(t & 1);
  const bool active = i < n;
  const size_t ii = active ? i : 0;          // out-of-range lanes recompute element 0 (uniform control flow), store nothing
  S12 sa;
  load_s12(sa, g, n, ii, odd);
  W12 a, na, one, res;
  w12_from_s12(a, sa);
  na = w12_conj(a);
  {
    S12 so = s12_one();
    w12_from_s12(one, so);
  }
  res = one;
  // digits of fp.rs:653-662 on the raw 256-bit scalar
  u32 k[8], xh[8], x3[8], np[8], nm[8];
  {
    const Fp kp = load_plain(ks, n, ii, 0);
#pragma unroll
    for (int j = 0; j < 8; ++j) k[j] = kp.v[j];
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) xh[j] = (k[j] >> 1) | (j < 7 ? (k[j + 1] << 31) : 0);
  u64 c = 0;
#pragma unroll
  for (int j = 0; j < 8; ++j) { c += (u64)k[j] + xh[j]; x3[j] = (u32)c; c >>= 32; }
#pragma unroll
  for (int j = 0; j < 8; ++j) { const u32 cc = xh[j] ^ x3[j]; np[j] = x3[j] & cc; nm[j] = xh[j] & cc; }
#pragma unroll 1
  for (int b = 255; b >= 0; --b) {
    res = w12_sqr(res);
    const bool bp = (np[b >> 5] >> (b & 31)) & 1, bm = (nm[b >> 5] >> (b & 31)) & 1;
    if (__any(bp || bm)) {
      W12 m;
      W2* mc[6] = {&m.c0.c0, &m.c0.c1, &m.c0.c2, &m.c1.c0, &m.c1.c1, &m.c1.c2};
      const W2* ac[6] = {&a.c0.c0, &a.c0.c1, &a.c0.c2, &a.c1.c0, &a.c1.c1, &a.c1.c2};
      const W2* nc[6] = {&na.c0.c0, &na.c0.c1, &na.c0.c2, &na.c1.c0, &na.c1.c1, &na.c1.c2};
      const W2* oc[6] = {&one.c0.c0, &one.c0.c1, &one.c0.c2, &one.c1.c0, &one.c1.c1, &one.c1.c2};
#pragma unroll
      for (int q = 0; q < 6; ++q) *mc[q] = w2_select(w2_select(*oc[q], *nc[q], bm), *ac[q], bp);
      w12_mul_nl(res, res, m);
    }
  }
  S12 sr;
  w12_to_s12(sr, res);
  if (active) store_s12(out, n, i, odd, sr);
}

// ------------------------------------------------------------------ one product over a whole batch ---------------------------
// glued_pairing over n pairs as ONE Gt (examples/verify_multiple_messages_same_signer.rs:41-60: 2n pairs, one final
// exponentiation, == identity).  The shared-squaring Miller value of a set of pairs is exactly the product of the per-pair Miller
// values ((prod f_i)^2 = prod f_i^2), so the batch is cut into chunks of KPROD pairs per lane pair (k_multi_pairing with
// raw_miller = 1), the chunk values are multiplied together by a log-depth tree of Fp12 products, and one lane pair runs the
// final exponentiation.
__global__ void k_chunk_offsets(u64* off, size_t n_jobs, size_t n_pairs) {
  const size_t j = TID;
  if (j > n_jobs) return;
  const size_t v = j * (size_t)KPROD;
  off[j] = v < n_pairs ? v : n_pairs;
}
// out[i] = in[2 i] * in[2 i + 1] (the odd tail is copied), SoA strides n_in / n_out
__global__ void HEAVY_BOUNDS k_fp12_tree_level(const u64* in, size_t n_in, u64* out, size_t n_out) {
  const size_t t = TID, i = t >> 1;
  const int odd = (int)(t & 1);
  if (i >= n_out) return;
  S12 a;
  load_s12(a, in, n_in, 2 * i, odd);
  if (2 * i + 1 < n_in) {
    S12 b;
    load_s12(b, in, n_in, 2 * i + 1, odd);
    W12 x, y, r;
    w12_from_s12(x, a);
    w12_from_s12(y, b);
    w12_mul_nl(r, x, y);
    w12_to_s12(a, r);
  }
  store_s12(out, n_out, i, odd, a);
}
__global__ void HEAVY_BOUNDS k_final_exp_flag(const u64* fin, size_t n_in, u64* gout, uint8_t* is_one) {
  const size_t t = TID;
  const int odd = (int)(t & 1);
  if (t >= 2) return;
  S12 f, g;
  if (n_in) load_s12(f, fin, n_in, 0, odd); else f = s12_one();      // empty product = identity (pairing.rs:1218-1219)
  final_exponentiation29(g, f);
  if (gout) store_s12(gout, 1, 0, odd, g);
  const bool one = s12_is_one(g);
  if (is_one && !odd) is_one[0] = one ? 1 : 0;
}

// ------------------------------------------------------------------ G2 group law on lane pairs -------------------------------
// The complete RCB'15 formulas of bn254_pairing.hpp (proj_double / proj_add, generic over the coordinate field like
// group.rs) instantiated over the lane-pair Fp2 on the carry-free core: a projective G2 point is 27 VGPRs per lane.
// Class invariant and value bounds as for OpsF29 (bn254_pairing.hpp): coordinates N-class, additions carry-normalised,
// products |V| < 2 VaVb/169 + 1; with inputs |V| <= 7 proj_double returns |V| <= 2.3, proj_add (inputs <= 2.3) <= 2.3.
struct OpsW2 {
  typedef W2 F;
  static BN_DEV F add(const F& a, const F& b) { return w2_norm(w2_add(a, b)); }
  static BN_DEV F sub(const F& a, const F& b) { return w2_norm(w2_sub(a, b)); }
  static BN_DEV F neg(const F& a) { return w2_norm(w2_neg(a)); }
  static BN_DEV F mul(const F& a, const F& b) { return w2_mul(a, b); }
  static BN_DEV F zero() { return W2{OpsF29::zero()}; }
  static BN_DEV F one() { return W2{sel9(lane_odd(), OpsF29::one(), OpsF29::zero())}; }
  static BN_DEV bool is_zero(const F& a) { return s2_is_zero(w2_to_s2(a)); }
  static BN_DEV F select(const F& a, const F& b, bool c) { return w2_select(a, b, c); }
  static BN_DEV F mul_b3(const F& a) { return w2_mul(a, w2_const(C_TWIST_B3)); }
};
typedef Proj<W2> G2Q;
BN_NOINLINE void g2q_double(G2Q& r, const G2Q& p) { r = proj_double<OpsW2>(p); }
BN_NOINLINE void g2q_add(G2Q& r, const G2Q& p, const G2Q& q) { r = proj_add<OpsW2>(p, q); }
BN_NOINLINE void g2q_scalar_mul(G2Q& out, const G2Q& p, const u32 (&k)[8], int nwin = 64) {
  out = scalar_mul_window<OpsW2>(p, k, [](const G2Q& a) { G2Q r; g2q_double(r, a); return r; },
                                 [](const G2Q& a, const G2Q& b) { G2Q r; g2q_add(r, a, b); return r; }, nwin);
}
// group.rs:475-495 (through the saturated core: one Fp2 inversion)
BN_DEV void g2q_to_affine(S2& x, S2& y, bool& inf, const G2Q& p) {
  const S2 zi = s2_inv(w2_to_s2(p.z));
  inf = s2_is_zero(zi);
  x = s2_select(s2_mul(w2_to_s2(p.x), zi), s2_zero(), inf);
  y = s2_select(s2_mul(w2_to_s2(p.y), zi), s2_one(), inf);
}
BN_DEV bool g2q_on_curve_affine(const S2& x, const S2& y) {      // g2.rs:279-297
  return s2_eq(s2_sub(s2_sqr(y), s2_mul(s2_sqr(x), x)), s2_const(C_TWIST_B));
}
// g2.rs:488-513: (x+1)Q + psi(xQ) + psi^2(xQ) == psi^3(2xQ) for Q on the twist, affine
BN_NOINLINE bool g2q_in_subgroup(const S2& x, const S2& y) {
  const G2Q q{w2_from_s2(x), w2_from_s2(y), OpsW2::one()};
  const u32 bx[8] = {(u32)BN_BLS_X, (u32)(BN_BLS_X >> 32), 0, 0, 0, 0, 0, 0};
  G2Q a;
  g2q_scalar_mul(a, q, bx, 17);                   // x < 2^63: 16 digits + the recoding carry
  // psi on projective coordinates: conj is a field automorphism, so psi(X:Y:Z) = (eps0 conj X : eps1 conj Y : conj Z)
  const W2 e0 = w2_const(C_EPS_EXP0), e1 = w2_const(C_EPS_EXP1);
  auto psi = [&](G2Q& r, const G2Q& p) {
    r.x = w2_mul(e0, w2_conj(p.x));
    r.y = w2_mul(e1, w2_conj(p.y));
    r.z = w2_conj(p.z);
  };
  G2Q b, c, l, r;
  psi(b, a);
  g2q_add(a, a, q);
  psi(c, b);
  g2q_add(l, c, b);
  g2q_add(l, l, a);
  psi(r, c);
  g2q_double(r, r);
  const G2Q nl = proj_neg<OpsW2>(l);
  g2q_add(r, r, nl);
  return OpsW2::is_zero(r.z);
}
BN_DEV G2Q load_g2q(const u64* xy, const uint8_t* inf, size_t n, size_t i, int odd) {
  return G2Q{w2_from_s2(load_s2(xy, n, i, 0, odd)), w2_from_s2(load_s2(xy, n, i, 8, odd)), (inf && inf[i]) ? OpsW2::zero() : OpsW2::one()};
}
BN_DEV void store_g2q_affine(u64* oxy, uint8_t* oinf, size_t n, size_t i, int odd, const G2Q& r) {
  S2 x, y; bool rinf;
  g2q_to_affine(x, y, rinf, r);
  store_s2(oxy, n, i, 0, odd, x); store_s2(oxy, n, i, 8, odd, y);
  if (!odd) oinf[i] = rinf ? 1 : 0;
}
__global__ void HEAVY_BOUNDS k_g2_scalar_mul(const u64* pxy, const uint8_t* pinf, const u64* ks, u64* oxy, uint8_t* oinf, size_t n) {
  const size_t t = TID, i = t >> 1;
  const int odd = (int)(t & 1);
  if (i >= n) return;
  u32 k[8];
  load_scalar(k, ks, n, i);
  G2Q r;
  g2q_scalar_mul(r, load_g2q(pxy, pinf, n, i, odd), k);
  store_g2q_affine(oxy, oinf, n, i, odd, r);
}
__global__ void HEAVY_BOUNDS k_g2_add(const u64* axy, const uint8_t* ainf, const u64* bxy, const uint8_t* binf, u64* oxy, uint8_t* oinf, size_t n) {
  const size_t t = TID, i = t >> 1;
  const int odd = (int)(t & 1);
  if (i >= n) return;
  G2Q r;
  g2q_add(r, load_g2q(axy, ainf, n, i, odd), load_g2q(bxy, binf, n, i, odd));
  store_g2q_affine(oxy, oinf, n, i, odd, r);
}
__global__ void HEAVY_BOUNDS k_g2_double(const u64* axy, const uint8_t* ainf, u64* oxy, uint8_t* oinf, size_t n) {
  const size_t t = TID, i = t >> 1;
  const int odd = (int)(t & 1);
  if (i >= n) return;
  G2Q r;
  g2q_double(r, load_g2q(axy, ainf, n, i, odd));
  store_g2q_affine(oxy, oinf, n, i, odd, r);
}
__global__ void HEAVY_BOUNDS k_g2_normalize(const u64* pxyz, u64* oxy, uint8_t* oinf, size_t n) {
  const size_t t = TID, i = t >> 1;
  const int odd = (int)(t & 1);
  if (i >= n) return;
  const G2Q p{w2_from_s2(load_s2(pxyz, n, i, 0, odd)), w2_from_s2(load_s2(pxyz, n, i, 8, odd)), w2_from_s2(load_s2(pxyz, n, i, 16, odd))};
  store_g2q_affine(oxy, oinf, n, i, odd, p);
}
// G2Affine::endomorphism (g2.rs:140-152): psi(x, y) = (eps0 conj x, eps1 conj y), psi(identity) = identity; status reports the
// on-curve re-check the reference performs on the result (it panics there; here NOT_ON_CURVE)
__global__ void __launch_bounds__(BLOCK) k_g2_psi(const u64* qxy, const uint8_t* qinf, u64* oxy, uint8_t* oinf, uint8_t* status, size_t n) {
  const size_t t = TID, i = t >> 1;
  const int odd = (int)(t & 1);
  if (i >= n) return;
  const bool inf = qinf && qinf[i];
  S2 x = load_s2(qxy, n, i, 0, odd), y = load_s2(qxy, n, i, 8, odd), px, py;
  g2_psi_affine(px, py, x, y);
  const bool on = inf || g2q_on_curve_affine(px, py);
  store_s2(oxy, n, i, 0, odd, inf ? x : px);
  store_s2(oxy, n, i, 8, odd, inf ? y : py);
  if (!odd) { oinf[i] = inf ? 1 : 0; if (status) status[i] = on ? SYLOW_HIP_ST_OK : SYLOW_HIP_ST_NOT_ON_CURVE; }
}
// g2.rs:460-525 on an affine input
__global__ void HEAVY_BOUNDS k_g2_subgroup_check(const u64* qxy, const uint8_t* qinf, uint8_t* status, size_t n) {
  const size_t t = TID, i = t >> 1;
  const int odd = (int)(t & 1);
  if (i >= n) return;
  uint8_t st = SYLOW_HIP_ST_OK;
  if (!(qinf && qinf[i])) {                       // Z == 0 passes both tests (g2.rs:469,510)
    const S2 x = load_s2(qxy, n, i, 0, odd), y = load_s2(qxy, n, i, 8, odd);
    if (!g2q_on_curve_affine(x, y)) st = SYLOW_HIP_ST_NOT_ON_CURVE;
    else if (!g2q_in_subgroup(x, y)) st = SYLOW_HIP_ST_NOT_IN_SUBGROUP;
  }
  if (!odd) status[i] = st;
}

// ------------------------------------------------------------------ hash to G1 on a lane pair ---------------------------
// g1.rs:307-331: map(u0) + map(u1).  The even lane maps u0, the odd lane u1 (the two SvdW maps are independent), the points
// are exchanged and both lanes finish with the same complete addition and affine normalisation.
BN_DEV bool hash_to_g1_pair(Fp& hx, Fp& hy, bool& hinf, const uint8_t* msg, size_t msg_len, const DstPrime& dp) {
  const bool odd = lane_odd();
  uint8_t em[96];
  expand_message_xmd96(em, msg, msg_len, dp);
  const Fp u = fp_from_be48(em + (odd ? 48 : 0));
  Fp x, y;
  u32 ok = svdw_map(x, y, u) ? 1u : 0u;
  ok &= swap_u32(ok);
  const Fp ox = xchg(x), oy = xchg(y);
  const G1P a{sel(odd, x, ox), sel(odd, y, oy), fp_one()}, b{sel(odd, ox, x), sel(odd, oy, y), fp_one()};
  const G1P h = g1_add(a, b);
  g1_to_affine(hx, hy, hinf, h);
  return ok != 0;
}

// ------------------------------------------------------------------ G2 line tables on the carry-free core ---------------
// [87][3 coefficients][2 coordinates][9 limbs] int32, R-class: G2Affine::precompute (pairing.rs:676-708) of one point
constexpr int LINE_TABLE_WORDS = 87 * 54;
__device__ i32 g_g2gen_lines29[LINE_TABLE_WORDS];
// launched with ONE lane pair: the generator (qxy == nullptr) or element idx of an SoA G2 array
__global__ void k_g2_lines29(const u64* qxy, size_t n, size_t idx, i32* table) {
  if (TID >= 2) return;
  const int odd = (int)(TID & 1);
  S2 qxs = s2_g2gen_x(), qys = s2_g2gen_y();
  if (qxy) { qxs = load_s2(qxy, n, idx, 0, odd); qys = load_s2(qxy, n, idx, 8, odd); }
  i32* tb = table ? table : g_g2gen_lines29;
  const W2 qx = w2_from_s2(qxs), qy = w2_from_s2(qys), nqy = w2_from_s2(s2_neg(qys));
  const W2 twist_b = w2_const(C_TWIST_B);
  G2W r{qx, qy, w2_from_s2(s2_one())};
  W2 l0, l1, l2;
  int at = 0;
  auto put = [&]() {
    const W2 c[3] = {l0, w2_reduce(l1), w2_reduce(l2)};
    for (int k = 0; k < 3; ++k) for (int j = 0; j < 9; ++j) tb[((at * 3 + k) * 2 + odd) * 9 + j] = c[k].c.v[j];
    ++at;
  };
  const u64 nz = BN_ATE_NAF_NZ, ng = BN_ATE_NAF_NEG;
#pragma unroll 1
  for (int i = 0; i < 64; ++i) {
    g2_doubling_step29(r, l0, l1, l2, twist_b); put();
    if ((nz >> (63 - i)) & 1) { g2_addition_step29(r, qx, ((ng >> (63 - i)) & 1) ? nqy : qy, l0, l1, l2); put(); }
  }
  S2 q1x, q1y, q2x, q2y;
  g2_psi_affine(q1x, q1y, qxs, qys);
  g2_psi_affine(q2x, q2y, q1x, q1y);
  g2_addition_step29(r, w2_from_s2(q1x), w2_from_s2(q1y), l0, l1, l2); put();
  g2_addition_step29(r, w2_from_s2(q2x), w2_from_s2(s2_neg(q2y)), l0, l1, l2); put();
}
// block-cooperative copy of a line table into LDS
BN_DEV void stage_table(i32* lds, const i32* src) {
  for (int k = threadIdx.x; k < LINE_TABLE_WORDS; k += blockDim.x) lds[k] = src[k];
}
BN_DEV W2 table_w2(const i32* tab, int at, int c, int odd) {
  const i32* t = tab + ((at * 3 + c) * 2 + odd) * 9;
  return W2{F29{{t[0], t[1], t[2], t[3], t[4], t[5], t[6], t[7], t[8]}}};
}

// ------------------------------------------------------------------ BLS verification ----------------------------------------
// lib.rs:223-236 as written: pairing(sig, G2gen) == pairing(H(msg), pk), two Miller loops and two final exponentiations
__global__ void HEAVY_BOUNDS k_bls_verify(const u64* pkxy, const uint8_t* pkinf, const uint8_t* msgs, const u64* off, DstPrime dp,
                                          const u64* sigxy, const uint8_t* siginf, const i32* gen_table, uint8_t* okout, size_t n) {
  __shared__ i32 tabA[LINE_TABLE_WORDS];
  stage_table(tabA, gen_table);
  __syncthreads();
  const size_t t = TID, i = t >> 1;
  const int odd = (int)(t & 1);
  if (i >= n) return;
  Fp hx, hy; bool hinf;
  hash_to_g1_pair(hx, hy, hinf, msgs + off[i], (size_t)(off[i + 1] - off[i]), dp);
  S12 lhs, rhs;
  if (siginf && siginf[i]) {
    lhs = s12_one();
  } else {
    // G2PreComputed::miller_loop (pairing.rs:590-619) against the generator's line table
    const F29 sx = f29_reduce(f29_from_fp(load_fp(sigxy, n, i, 0))), sy = f29_reduce(f29_from_fp(load_fp(sigxy, n, i, 4)));
    W12 f;
    {
      S12 one = s12_one();
      w12_from_s12(f, one);
    }
    const u64 nz = BN_ATE_NAF_NZ;
    int idx = 0;
    auto line = [&]() {
      f = w12_sparse_mul(f, table_w2(tabA, idx, 0, odd), w2_scale(table_w2(tabA, idx, 1, odd), sy), w2_scale(table_w2(tabA, idx, 2, odd), sx));
      ++idx;
    };
#pragma unroll 1
    for (int it = 0; it < 64; ++it) {
      f = w12_sqr(f);
      line();
      if ((nz >> (63 - it)) & 1) line();
    }
    line();
    line();
    S12 fs;
    w12_to_s12(fs, f);
    final_exponentiation29(lhs, fs);
  }
  if (hinf || (pkinf && pkinf[i])) {
    rhs = s12_one();
  } else {
    const S2 qx = load_s2(pkxy, n, i, 0, odd), qy = load_s2(pkxy, n, i, 8, odd);
    S12 f;
    miller_loop29g(f, hx, hy, qx, qy);
    final_exponentiation29(rhs, f);
  }
  bool eq = s2_eq(lhs.c0.c0, rhs.c0.c0) && s2_eq(lhs.c0.c1, rhs.c0.c1) && s2_eq(lhs.c0.c2, rhs.c0.c2) &&
            s2_eq(lhs.c1.c0, rhs.c1.c0) && s2_eq(lhs.c1.c1, rhs.c1.c1) && s2_eq(lhs.c1.c2, rhs.c1.c2);
  if (!odd) okout[i] = eq ? 1 : 0;
}

// e(sig, G2gen) * e(-H(msg), pk) == 1 with one shared-squaring Miller loop and one final exponentiation (see the single-lane
// k_bls_verify_fused in sylow_hip.hip for the contract).  PK_TABLE: one public key for the whole batch, its lines precomputed.
template <bool PK_TABLE>
__global__ void HEAVY_BOUNDS k_bls_verify_fused(const u64* pkxy, const uint8_t* pkinf, const i32* pk_table, const uint8_t* msgs, const u64* off, DstPrime dp,
                                                const u64* sigxy, const uint8_t* siginf, const i32* gen_table, uint8_t* okout, size_t n) {
  __shared__ i32 tabA[LINE_TABLE_WORDS];
  __shared__ i32 tabB[PK_TABLE ? LINE_TABLE_WORDS : 1];
  stage_table(tabA, gen_table);
  if (PK_TABLE) stage_table(tabB, pk_table);
  __syncthreads();
  const size_t t = TID, i = t >> 1;
  const int odd = (int)(t & 1);
  const bool active = i < n;
  const size_t ii = active ? i : 0;          // out-of-range lanes recompute element 0 (uniform control flow), store nothing
  Fp hxs, hys; bool hinf;
  hash_to_g1_pair(hxs, hys, hinf, msgs + off[ii], (size_t)(off[ii + 1] - off[ii]), dp);
  const F29 hx = f29_reduce(f29_from_fp(hxs)), hy = f29_reduce(f29_from_fp(fp_neg(hys)));   // pair B is (-H, pk)
  const bool liveA = !(siginf && siginf[ii]);
  const bool liveB = !(hinf || (pkinf && pkinf[PK_TABLE ? 0 : ii]));
  const F29 sx = f29_reduce(f29_from_fp(load_fp(sigxy, n, ii, 0))), sy = f29_reduce(f29_from_fp(load_fp(sigxy, n, ii, 4)));
  // a dead pair B steps the generator instead (any curve point keeps the arithmetic defined) and multiplies by the unit line
  const S2 qxs = (PK_TABLE || !liveB) ? s2_g2gen_x() : load_s2(pkxy, n, ii, 0, odd);
  const S2 qys = (PK_TABLE || !liveB) ? s2_g2gen_y() : load_s2(pkxy, n, ii, 8, odd);
  const W2 qx = w2_from_s2(qxs), qy = w2_from_s2(qys), nqy = w2_from_s2(s2_neg(qys));
  const W2 w_one = w2_from_s2(s2_one()), w_zero = W2{F29{{0, 0, 0, 0, 0, 0, 0, 0, 0}}};
  const W2 twist_b = w2_const(C_TWIST_B);
  G2W r{qx, qy, w_one};
  W12 f;
  {
    S12 one = s12_one();
    w12_from_s12(f, one);
  }
  W2 l0, l1, l2;
  int idx = 0;
  auto lineA = [&]() {
    const W2 a0 = table_w2(tabA, idx, 0, odd), a1 = w2_scale(table_w2(tabA, idx, 1, odd), sy), a2 = w2_scale(table_w2(tabA, idx, 2, odd), sx);
    f = w12_sparse_mul(f, w2_select(w_one, a0, liveA), w2_select(w_zero, a1, liveA), w2_select(w_zero, a2, liveA));
  };
  auto lineB = [&]() {
    if (PK_TABLE) { l0 = table_w2(tabB, idx, 0, odd); l1 = table_w2(tabB, idx, 1, odd); l2 = table_w2(tabB, idx, 2, odd); }
    f = w12_sparse_mul(f, w2_select(w_one, l0, liveB), w2_select(w_zero, w2_scale(l1, hy), liveB), w2_select(w_zero, w2_scale(l2, hx), liveB));
  };
  const u64 nz = BN_ATE_NAF_NZ, ng = BN_ATE_NAF_NEG;
#pragma unroll 1
  for (int it = 0; it < 64; ++it) {
    f = w12_sqr(f);
    lineA();
    if (!PK_TABLE) g2_doubling_step29(r, l0, l1, l2, twist_b);
    lineB();
    ++idx;
    if ((nz >> (63 - it)) & 1) {
      lineA();
      if (!PK_TABLE) g2_addition_step29(r, qx, ((ng >> (63 - it)) & 1) ? nqy : qy, l0, l1, l2);
      lineB();
      ++idx;
    }
  }
  S2 q1x, q1y, q2x, q2y;
  if (!PK_TABLE) { g2_psi_affine(q1x, q1y, qxs, qys); g2_psi_affine(q2x, q2y, q1x, q1y); }
  lineA();
  if (!PK_TABLE) g2_addition_step29(r, w2_from_s2(q1x), w2_from_s2(q1y), l0, l1, l2);
  lineB();
  ++idx;
  lineA();
  if (!PK_TABLE) g2_addition_step29(r, w2_from_s2(q2x), w2_from_s2(s2_neg(q2y)), l0, l1, l2);
  lineB();
  S12 fs, g;
  w12_to_s12(fs, f);
  final_exponentiation29(g, fs);
  const bool one = s12_is_one(g);
  if (active && !odd) okout[i] = one ? 1 : 0;
}

}  // namespace plk
