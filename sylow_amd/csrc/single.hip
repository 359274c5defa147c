// single.hip -- the one-element-per-lane kernels: the tower test hooks, and the slower twins of the lane-pair pairing / G2 /
// verification kernels (SYLOW_HIP_SINGLE_LANE=1 selects them: a second implementation for the parity tests and for A/B runs).
#include "host.hpp"

// ------------------------------------------------------------------ tower test hooks ----------
enum { OPX_RESIDUE_MUL = 16, OPX_FROB_ODD = 17, OPX_COPY = 18, OPX_FROB6 = 32 };
__global__ void __launch_bounds__(BLOCK) k_fp2_op(int op, const u64* a, const u64* b, u64* out, size_t n) {
  size_t i = TID;
  if (i >= n) return;
  Fp2 x = load_fp2(a, n, i, 0), r;
  if (op == OP_MUL) r = fp2_mul(x, load_fp2(b, n, i, 0));
  else if (op == OP_SQR) r = fp2_sqr(x);
  else if (op == OPX_RESIDUE_MUL) r = fp2_mul_xi(x);                 // Fp2::residue_mul (fp2.rs:99-107)
  else if (op == OPX_FROB_ODD) r = fp2_conj(x);                      // Fp2::frobenius(odd) (fp2.rs:119-133); even exponents are the identity
  else if (op == OPX_COPY) r = x;
  else r = fp2_inv(x);
  store_fp2(out, n, i, 0, r);
}
__global__ void HEAVY_BOUNDS k_fp6_op(int op, const u64* a, const u64* b, u64* out, size_t n) {
  size_t i = TID;
  if (i >= n) return;
  Fp6 x, y, r;
  load_fp6(x, a, n, i, 0);
  if (op == OP_MUL) { load_fp6(y, b, n, i, 0); fp6_mul(r, x, y); }
  else if (op == OP_SQR) fp6_sqr(r, x);                              // Fp6::square (fp6.rs:213-236)
  else if (op == OPX_RESIDUE_MUL) r = fp6_mul_v(x);                  // Fp6::residue_mul (fp6.rs:189-192)
  else if (op == OPX_COPY) r = x;
  else if (op >= OPX_FROB6 && op <= OPX_FROB6 + 5) {                 // Fp6::frobenius(e), e mod 6 (fp6.rs:205-211): the tables of
    const int e = op - OPX_FROB6;                                    // exponents 4 and 5 are those of 1 and 2 composed with 3
    r = x;
    if (e >= 3) { fp6_frobenius<3>(y, r); r = y; }
    if (e % 3 == 1) { fp6_frobenius<1>(y, r); r = y; }
    if (e % 3 == 2) { fp6_frobenius<2>(y, r); r = y; }
  }
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

// ------------------------------------------------------------------ fused BLS verify ------------
// The batch-verify shape sylow's own examples recommend (examples/verify_multiple_messages_same_signer.rs:41-60,
// threshold_signing.rs:92-121): e(sig, G2gen) * e(-H(msg), pk) == 1 with ONE shared-squaring Miller loop
// and ONE final exponentiation.  Pairs with an identity contribute 1, exactly as pairing() treats them
// (pairing.rs:876-886), so for points in G1 x G2 the boolean equals verify()'s (lib.rs:223-236).
// The G2 generator is fixed, so its 87 line-coefficient triples (pairing.rs:676-708) are computed once
// per device (host::gen_lines_sat) and read with wave-uniform (scalar) loads.
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
  u32* t = table;
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
                                          const u64* sigxy, const uint8_t* siginf, const u32* gen_table, uint8_t* okout, size_t n) {
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
    miller_loop_table(f, sx, sy, gen_table);     // Q = G2 generator: its [Ell; 87] comes from the per-device table
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
                                                const u64* sigxy, const uint8_t* siginf, const u32* gen_table, uint8_t* okout, size_t n) {
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
    Fp2 a0 = table_fp2(gen_table, at, 0), a1 = fp2_scale(table_fp2(gen_table, at, 1), sy), a2 = fp2_scale(table_fp2(gen_table, at, 2), sx);
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

// ================================================================== launchers ======================
namespace single {
int32_t g2_scalar_mul(const uint64_t* p_xy, const uint8_t* p_inf, const uint64_t* k, uint64_t* out_xy, uint8_t* out_inf, size_t n, void* stream) {
  k_g2_scalar_mul<<<GRID(n)>>>(p_xy, p_inf, k, out_xy, out_inf, n); LAUNCHED();
}
int32_t g2_normalize(const uint64_t* p_xyz, uint64_t* out_xy, uint8_t* out_inf, size_t n, void* stream) {
  k_g2_normalize<<<GRID(n)>>>(p_xyz, out_xy, out_inf, n); LAUNCHED();
}
int32_t g2_subgroup_check(const uint64_t* q_xy, const uint8_t* q_inf, uint8_t* status, size_t n, void* stream) {
  k_g2_subgroup_check<<<GRID(n)>>>(q_xy, q_inf, status, n); LAUNCHED();
}
int32_t g2_add(const uint64_t* a_xy, const uint8_t* a_inf, const uint64_t* b_xy, const uint8_t* b_inf, uint64_t* out_xy, uint8_t* out_inf, size_t n, void* stream) {
  k_g2_add<<<GRID(n)>>>(a_xy, a_inf, b_xy, b_inf, out_xy, out_inf, n); LAUNCHED();
}
int32_t g2_double(const uint64_t* a_xy, const uint8_t* a_inf, uint64_t* out_xy, uint8_t* out_inf, size_t n, void* stream) {
  k_g2_double<<<GRID(n)>>>(a_xy, a_inf, out_xy, out_inf, n); LAUNCHED();
}
int32_t miller_loop(const uint64_t* p_xy, const uint64_t* q_xy, uint64_t* f_out, size_t n, void* stream) {
  k_miller_loop<<<GRID(n)>>>(p_xy, q_xy, f_out, n); LAUNCHED();
}
int32_t final_exp(const uint64_t* f, uint64_t* gt_out, size_t n, void* stream) {
  k_final_exp<<<GRID(n)>>>(f, gt_out, n); LAUNCHED();
}
int32_t pairing(const uint64_t* p_xy, const uint8_t* p_inf, const uint64_t* q_xy, const uint8_t* q_inf, uint64_t* gt_out, size_t n, void* stream) {
  k_pairing<<<GRID(n)>>>(p_xy, p_inf, q_xy, q_inf, gt_out, n); LAUNCHED();
}
int32_t multi_pairing(const uint64_t* p_xy, const uint8_t* p_inf, const uint64_t* q_xy, const uint8_t* q_inf, const uint64_t* pair_offsets,
                      size_t n_jobs, size_t n_pairs, int32_t skip_infinity, uint64_t* gt_out, uint8_t* is_one, void* stream) {
  k_multi_pairing<<<GRID(n_jobs)>>>(p_xy, p_inf, q_xy, q_inf, pair_offsets, n_jobs, n_pairs, skip_infinity, gt_out, is_one); LAUNCHED();
}
int32_t gt_pow(const uint64_t* gt, const uint64_t* k, uint64_t* out, size_t n, void* stream) {
  k_gt_pow<<<GRID(n)>>>(gt, k, out, n); LAUNCHED();
}
int32_t build_gen_lines(u32* table, void* stream) {
  k_g2_lines<<<1, 64, 0, (hipStream_t)stream>>>(nullptr, 0, 0, table); LAUNCHED();
}
int32_t bls_verify(const uint64_t* pk_xy, const uint8_t* pk_inf, const uint8_t* msgs, const uint64_t* msg_offsets,
                   const uint64_t* sig_xy, const uint8_t* sig_inf, uint8_t* ok, size_t n, void* stream) {
  DstPrime dp; host::dst_arg(dp, nullptr, 0);
  const u32* gen = nullptr;
  int32_t rc = host::gen_lines_sat(&gen, (hipStream_t)stream);
  if (rc != SYLOW_HIP_OK) return rc;
  k_bls_verify<<<GRID(n)>>>(pk_xy, pk_inf, msgs, msg_offsets, dp, sig_xy, sig_inf, gen, ok, n); LAUNCHED();
}
int32_t bls_verify_fused(const uint64_t* pk_xy, const uint8_t* pk_inf, const uint8_t* msgs, const uint64_t* msg_offsets,
                         const uint64_t* sig_xy, const uint8_t* sig_inf, uint8_t* ok, size_t n, void* stream) {
  DstPrime dp; host::dst_arg(dp, nullptr, 0);
  const u32* gen = nullptr;
  int32_t rc = host::gen_lines_sat(&gen, (hipStream_t)stream);
  if (rc != SYLOW_HIP_OK) return rc;
  k_bls_verify_fused<false><<<GRID(n)>>>(pk_xy, pk_inf, nullptr, msgs, msg_offsets, dp, sig_xy, sig_inf, gen, ok, n); LAUNCHED();
}
int32_t bls_verify_same_signer(const uint64_t* pk_xy, const uint8_t* pk_inf, const uint8_t* msgs, const uint64_t* msg_offsets,
                               const uint64_t* sig_xy, const uint8_t* sig_inf, uint8_t* ok, size_t n, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  DstPrime dp; host::dst_arg(dp, nullptr, 0);
  const u32* gen = nullptr;
  int32_t rc = host::gen_lines_sat(&gen, st);
  if (rc != SYLOW_HIP_OK) return rc;
  host::Lease ws;
  if ((rc = ws.acquire(87 * 48 * sizeof(u32), st)) != SYLOW_HIP_OK) return rc;
  u32* table = (u32*)ws.p;
  k_g2_lines<<<1, 64, 0, st>>>(pk_xy, 1, 0, table);          // the key is a 1-element SoA array
  k_bls_verify_fused<true><<<GRID(n)>>>(pk_xy, pk_inf, table, msgs, msg_offsets, dp, sig_xy, sig_inf, gen, ok, n);
  const hipError_t e = hipGetLastError();
  rc = ws.release();
  return e != hipSuccess ? host::fail(e, "kernel launch") : rc;
}
}  // namespace single

// ================================================================== C ABI (entry points implemented only here) ==========
extern "C" {
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
int32_t sylow_hip_fp2_residue_mul_batch(const uint64_t* a, uint64_t* out, size_t n, void* stream) {
  ARGCHK(a && out); if (!n) return SYLOW_HIP_OK; k_fp2_op<<<GRID(n)>>>(OPX_RESIDUE_MUL, a, nullptr, out, n); LAUNCHED();
}
int32_t sylow_hip_fp2_frobenius_batch(const uint64_t* a, uint64_t exponent, uint64_t* out, size_t n, void* stream) {
  ARGCHK(a && out); if (!n) return SYLOW_HIP_OK; k_fp2_op<<<GRID(n)>>>((exponent & 1) ? OPX_FROB_ODD : OPX_COPY, a, nullptr, out, n); LAUNCHED();
}
int32_t sylow_hip_fp6_sqr_batch(const uint64_t* a, uint64_t* out, size_t n, void* stream) {
  ARGCHK(a && out); if (!n) return SYLOW_HIP_OK; k_fp6_op<<<GRID(n)>>>(OP_SQR, a, nullptr, out, n); LAUNCHED();
}
int32_t sylow_hip_fp6_residue_mul_batch(const uint64_t* a, uint64_t* out, size_t n, void* stream) {
  ARGCHK(a && out); if (!n) return SYLOW_HIP_OK; k_fp6_op<<<GRID(n)>>>(OPX_RESIDUE_MUL, a, nullptr, out, n); LAUNCHED();
}
int32_t sylow_hip_fp6_frobenius_batch(const uint64_t* a, uint64_t exponent, uint64_t* out, size_t n, void* stream) {
  ARGCHK(a && out); if (!n) return SYLOW_HIP_OK; k_fp6_op<<<GRID(n)>>>(OPX_FROB6 + (int)(exponent % 6), a, nullptr, out, n); LAUNCHED();
}
int32_t sylow_hip_fp12_mul_batch(const uint64_t* a, const uint64_t* b, uint64_t* out, size_t n, void* stream) {
  ARGCHK(a && b && out); if (!n) return SYLOW_HIP_OK;
  if (!host::single_lane()) return plkh::fp12_op(16, a, b, out, n, stream);
  k_fp12_op<<<GRID(n)>>>(OP12_MUL, a, b, out, n); LAUNCHED();
}
int32_t sylow_hip_fp12_sqr_batch(const uint64_t* a, uint64_t* out, size_t n, void* stream) {
  ARGCHK(a && out); if (!n) return SYLOW_HIP_OK;
  if (!host::single_lane()) return plkh::fp12_op(17, a, nullptr, out, n, stream);
  k_fp12_op<<<GRID(n)>>>(OP12_SQR, a, nullptr, out, n); LAUNCHED();
}
int32_t sylow_hip_fp12_inv_batch(const uint64_t* a, uint64_t* out, size_t n, void* stream) {
  ARGCHK(a && out); if (!n) return SYLOW_HIP_OK;
  if (!host::single_lane()) return plkh::fp12_op(26, a, nullptr, out, n, stream);
  k_fp12_op<<<GRID(n)>>>(OP12_INV, a, nullptr, out, n); LAUNCHED();
}
int32_t sylow_hip_fp12_frobenius_batch(const uint64_t* a, int32_t exponent, uint64_t* out, size_t n, void* stream) {
  ARGCHK(a && out && exponent >= 1 && exponent <= 3); if (!n) return SYLOW_HIP_OK;
  if (!host::single_lane()) return plkh::fp12_op(20 + exponent - 1, a, nullptr, out, n, stream);
  k_fp12_op<<<GRID(n)>>>(OP12_FROB1 + exponent - 1, a, nullptr, out, n); LAUNCHED();
}
int32_t sylow_hip_fp12_sparse_mul_batch(const uint64_t* f, const uint64_t* ell, uint64_t* out, size_t n, void* stream) {
  ARGCHK(f && ell && out); if (!n) return SYLOW_HIP_OK;
  if (!host::single_lane()) return plkh::fp12_op(18, f, ell, out, n, stream);
  k_fp12_op<<<GRID(n)>>>(OP12_SPARSE, f, ell, out, n); LAUNCHED();
}
// test hook (not in the public header's stable surface): Granger-Scott cyclotomic square
int32_t sylow_hip_fp12_cyclotomic_sqr_batch(const uint64_t* a, uint64_t* out, size_t n, void* stream) {
  ARGCHK(a && out); if (!n) return SYLOW_HIP_OK;
  if (!host::single_lane()) return plkh::fp12_op(19, a, nullptr, out, n, stream);
  k_fp12_op<<<GRID(n)>>>(OP12_CYCSQR, a, nullptr, out, n); LAUNCHED();
}

// test hook: raw k_fp12_op selector (8: product on the carry-free core, 9: cyclotomic square on it,
// 10 / 11: exp_by_neg_z on the carry-free / saturated core); selectors 16..28 (the lane-pair Fp12 layer) are served by
// sylow_hip_fp12_hook_batch in plk_pairing.hip, which forwards the others here
}  // extern "C"
namespace single {
int32_t fp12_hook(int32_t op, const uint64_t* a, const uint64_t* b, uint64_t* out, size_t n, void* stream) {
  k_fp12_op<<<GRID(n)>>>(op, a, b, out, n); LAUNCHED();
}
}  // namespace single
namespace single {
int32_t g2_precompute(const uint64_t* q_xy, uint64_t* coeffs, size_t n, void* stream) {
  k_g2_precompute<<<GRID(n)>>>(q_xy, coeffs, n); LAUNCHED();
}
}  // namespace single
