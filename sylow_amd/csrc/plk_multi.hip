// plk_multi.hip -- glued pairings on lane pairs: one product per job (ecPairing shape), raw glued Miller values, and the
// batch-wide product (chunked Miller loops + a log-depth tree of Fp12 products + one final exponentiation).
#include "plk_common.hpp"

#include <atomic>

namespace plk {
// ------------------------------------------------------------------ glued pairing ----------------------------------------
// Wave-uniform schedule: chunks of KMAX pairs share the squarings of
// one accumulator, a lane pair whose job has fewer pairs multiplies by the unit line.  Pair states live in the stack frame
// (they are touched once per loop iteration); the accumulator and the working point stay in registers.
struct PairStateW { G2W r; W2 qx, qy; S2 qxs, qys; F29 px, py; bool qinf, live; };
constexpr int KMAXW = 4;        // ecPairing / glued jobs (a few pairs each)
constexpr int KPROD = 8;        // batch-wide product: more pairs per shared squaring

// the raw glued Miller value of this lane pair's job [next, hi), any number of pairs.
// Out of line ON PURPOSE: the stack frame of a kernel is its own frame plus the deepest callee chain, so with the Miller part
// inlined next to the call of final_exponentiation29 the two frames ADD (4.3 - 6.1 KB per lane); as siblings they overlap
// (max instead of sum).  That matters beyond spill traffic: the runtime's scratch pool holds 2 waves per SIMD only up to
// 4 KB per lane (512 MB / 2048 waves / 64 lanes) -- measured: at 4.3 KB k_multi_pairing ran 89 % of k_pairing's wave
// occupancy, the 6.1 KB product kernel 65 %.
template <int KMAX>
BN_NOINLINE void glued_miller_chunks(W12& acc, const u64* pxy, const uint8_t* pinf, const u64* qxy, const uint8_t* qinf,
                                size_t next, size_t hi, size_t n_pairs, int skip_infinity, int odd) {
  const W2 w_one = w2_from_s2(s2_one()), w_zero = W2{F29{{0, 0, 0, 0, 0, 0, 0, 0, 0}}};
  {
    S12 one = s12_one();
    w12_from_s12(acc, one);
  }
  PairStateW st[KMAX];
  const u64 nz = BN_ATE_NAF_NZ, ng = BN_ATE_NAF_NEG;
#pragma unroll 1
  while (wave_max(next < hi ? 1 : 0)) {
    int k = 0;
#pragma unroll(KMAX == 2 ? 2 : 1)
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
    // per-pair bodies.  KMAX == 2 (the BLS / ecPairing shape) spells the two slots out, so that their states are plain values the
    // register allocator can keep in VGPRs (spilling only the once-per-step operands); larger KMAX walks the slots in the stack frame
    auto step_dbl = [&](PairStateW& s) { g2_doubling_step29(s.r, l0, l1, l2); apply(s); };
    auto step_add = [&](PairStateW& s, bool neg) {
      const W2 by = neg ? w2_neg(s.qy) : s.qy;
      g2_addition_step29(s.r, s.qx, by, l0, l1, l2);
      apply(s);
    };
    auto step_frob = [&](PairStateW& s, int step) {
      S2 q1x, q1y, q2x, q2y;
      g2_psi_affine(q1x, q1y, s.qxs, s.qys);
      g2_psi_affine(q2x, q2y, q1x, q1y);
      const bool qi = s.qinf;
      q1x = s2_select(q1x, s.qxs, qi); q1y = s2_select(q1y, s.qys, qi);
      q2x = s2_select(q2x, s.qxs, qi); q2y = s2_select(q2y, s.qys, qi);
      if (step == 0) g2_addition_step29(s.r, w2_from_s2(q1x), w2_from_s2(q1y), l0, l1, l2);
      else g2_addition_step29(s.r, w2_from_s2(q2x), w2_from_s2(s2_neg(q2y)), l0, l1, l2);
      apply(s);
    };
#pragma unroll 1
    for (int i = 0; i < 64; ++i) {
      f = w12_sqr(f);
      if constexpr (KMAX == 2) {
        step_dbl(st[0]);
        if (kw > 1) step_dbl(st[1]);
      } else {
#pragma unroll 1
        for (int j = 0; j < kw; ++j) step_dbl(st[j]);
      }
      if ((nz >> (63 - i)) & 1) {
        const bool neg = (ng >> (63 - i)) & 1;
        if constexpr (KMAX == 2) {
          step_add(st[0], neg);
          if (kw > 1) step_add(st[1], neg);
        } else {
#pragma unroll 1
          for (int j = 0; j < kw; ++j) step_add(st[j], neg);
        }
      }
    }
    // the two Frobenius additions; endomorphism() returns self for the identity (g2.rs:141-143)
#pragma unroll 1
    for (int step = 0; step < 2; ++step) {
      if constexpr (KMAX == 2) {
        step_frob(st[0], step);
        if (kw > 1) step_frob(st[1], step);
      } else {
#pragma unroll 1
        for (int j = 0; j < kw; ++j) step_frob(st[j], step);
      }
    }
    w12_mul_nl(acc, acc, f);
  }
}

// Jobs of at most TWO pairs (the BLS check e(sig, G2gen) e(-H, pk) and the k = 2 ecPairing shape), when every job of the wavefront
// is that small: one chunk, so no running product, and the two pair states are plain values -- the structure of the fused
// verifier (plk_verify.hip) with both G2 points general.  Slot A is pair `lo`, slot B pair `lo + 1`; a missing or skipped pair is
// a dead slot (generator point, unit line), exactly like the generic schedule, so the value is the same bit for bit.
BN_NOINLINE void glued_miller_upto2(W12& fout, const u64* pxy, const uint8_t* pinf, const u64* qxy, const uint8_t* qinf,
                               size_t lo, size_t hi, size_t n_pairs, int skip_infinity, int odd) {
  const W2 w_one = w2_from_s2(s2_one()), w_zero = W2{F29{{0, 0, 0, 0, 0, 0, 0, 0, 0}}};
  W12 f;                                             // a LOCAL accumulator (a reference parameter would be re-read from memory around every leaf call)
  {
    S12 one = s12_one();
    w12_from_s12(f, one);
  }
  auto setup = [&](PairStateW& s, size_t idx) {
    const bool exists = idx < hi;
    const size_t src = exists ? idx : 0;
    const bool pi = exists && pinf && pinf[src], qi0 = exists && qinf && qinf[src];
    const bool have = exists && !(skip_infinity && (pi || qi0));
    const bool ld = have && n_pairs != 0;
    s.live = have;
    s.qinf = have && qi0;
    s.px = f29_reduce(f29_from_fp(ld ? load_fp(pxy, n_pairs, src, 0) : fp_one()));
    s.py = f29_reduce(f29_from_fp(ld ? load_fp(pxy, n_pairs, src, 4) : fp_one()));
    s.qxs = ld ? load_s2(qxy, n_pairs, src, 0, odd) : s2_g2gen_x();
    s.qys = ld ? load_s2(qxy, n_pairs, src, 8, odd) : s2_g2gen_y();
    s.qx = w2_from_s2(s.qxs);
    s.qy = w2_from_s2(s.qys);
    s.r = G2W{s.qx, s.qy, s.qinf ? w_zero : w_one};
  };
  PairStateW A, B;
  setup(A, lo);
  setup(B, lo + 1);
  // a skipped first pair leaves slot A dead and B live: both slots are walked whenever any lane has a live B
  const bool anyA = wave_max(A.live ? 1 : 0) != 0, anyB = wave_max(B.live ? 1 : 0) != 0;
  if (!anyA && !anyB) { fout = f; return; }
  W2 l0, l1, l2;
  auto apply = [&](const PairStateW& s) {
    const bool lv = s.live;
    f = w12_sparse_mul(f, w2_select(w_one, l0, lv), w2_select(w_zero, w2_scale(l1, s.py), lv), w2_select(w_zero, w2_scale(l2, s.px), lv));
  };
  const u64 nz = BN_ATE_NAF_NZ, ng = BN_ATE_NAF_NEG;
#pragma unroll 1
  for (int i = 0; i < 64; ++i) {
    f = w12_sqr(f);
    if (anyA) { g2_doubling_step29(A.r, l0, l1, l2); apply(A); }
    if (anyB) { g2_doubling_step29(B.r, l0, l1, l2); apply(B); }
    if ((nz >> (63 - i)) & 1) {
      const bool neg = (ng >> (63 - i)) & 1;
      if (anyA) { g2_addition_step29(A.r, A.qx, neg ? w2_neg(A.qy) : A.qy, l0, l1, l2); apply(A); }
      if (anyB) { g2_addition_step29(B.r, B.qx, neg ? w2_neg(B.qy) : B.qy, l0, l1, l2); apply(B); }
    }
  }
  auto frob = [&](PairStateW& s, int step) {        // endomorphism() returns self for the identity (g2.rs:141-143)
    S2 q1x, q1y, q2x, q2y;
    g2_psi_affine(q1x, q1y, s.qxs, s.qys);
    g2_psi_affine(q2x, q2y, q1x, q1y);
    const bool qi = s.qinf;
    q1x = s2_select(q1x, s.qxs, qi); q1y = s2_select(q1y, s.qys, qi);
    q2x = s2_select(q2x, s.qxs, qi); q2y = s2_select(q2y, s.qys, qi);
    if (step == 0) g2_addition_step29(s.r, w2_from_s2(q1x), w2_from_s2(q1y), l0, l1, l2);
    else g2_addition_step29(s.r, w2_from_s2(q2x), w2_from_s2(s2_neg(q2y)), l0, l1, l2);
    apply(s);
  };
#pragma unroll 1
  for (int step = 0; step < 2; ++step) {
    if (anyA) frob(A, step);
    if (anyB) frob(B, step);
  }
  fout = f;
}

// The same value pair by pair: prod_i miller(P_i, Q_i) with every factor from the single-pair loop (miller_loop29g: invariants
// and the working point in LDS, no scratch access in its body) and one Fp12 product per pair.  Sharing the squarings saves 23 % of a
// Miller loop per extra pair on paper, but the shared loop keeps two to eight pair states alive next to the accumulator and pays for
// it in spill traffic: measured on one box per 2^18 jobs, shared 37.6 ms (k = 1) / 53.8 (k = 2) / 77.2 (k = 3) / 87.0 (k = 4)
// against 33.1 / 50.9 / 78.8 / 88.6 ms this way -- so jobs of one or two pairs (the BLS and k = 2 ecPairing shapes) come here and
// longer jobs keep the shared schedule.  Not for the reference-replay treatment of a G2 identity (Z = 0 walked through the
// formulas): the caller checks.
BN_NOINLINE void glued_miller_seq(W12& acc, const u64* pxy, const uint8_t* pinf, const u64* qxy, const uint8_t* qinf,
                             size_t lo, size_t hi, size_t n_pairs, int skip_infinity, int odd) {
  {
    S12 one = s12_one();
    w12_from_s12(acc, one);
  }
  const int rounds = wave_max((int)(hi - lo));
#pragma unroll 1
  for (int j = 0; j < rounds; ++j) {
    const size_t idx = lo + (size_t)j;
    const bool exists = idx < hi;
    const size_t src = exists ? idx : 0;
    const bool pi = exists && pinf && pinf[src], qi = exists && qinf && qinf[src];
    const bool live = exists && !(skip_infinity && (pi || qi));
    const bool ld = live && n_pairs != 0;
    const Fp px = ld ? load_fp(pxy, n_pairs, src, 0) : fp_one(), py = ld ? load_fp(pxy, n_pairs, src, 4) : fp_one();
    const S2 qx = ld ? load_s2(qxy, n_pairs, src, 0, odd) : s2_g2gen_x(), qy = ld ? load_s2(qxy, n_pairs, src, 8, odd) : s2_g2gen_y();
    S12 fs;
    miller_loop29g<true>(fs, px, py, qx, qy);
    if (!live) fs = s12_one();                       // a missing or skipped pair contributes 1
    W12 f;
    w12_from_s12(f, fs);
    w12_mul_nl(acc, acc, f);
  }
}

template <int KMAX>
__global__ void HEAVY_BOUNDS k_multi_pairing(const u64* pxy, const uint8_t* pinf, const u64* qxy, const uint8_t* qinf,
                                             const u64* offsets, size_t n_jobs, size_t n_pairs, int skip_infinity,
                                             u64* gout, uint8_t* is_one, int raw_miller) {
  const size_t t = TID, job = pair_index(t);
  const int odd = pair_role(t);
  const bool active = job < n_jobs;           // no early return: every lane takes part in the wave reductions
  const size_t lo = active ? offsets[job] : 0, hi = active ? offsets[job + 1] : 0;
  W12 acc;
  // reference-replay mode walks a G2 identity through the formulas: only the shared-schedule loops do that
  bool replay_identity = false;
  if (!skip_infinity && qinf) for (size_t idx = lo; idx < hi; ++idx) replay_identity = replay_identity || qinf[idx] != 0;
  if (KMAX != KPROD && wave_max(replay_identity ? 1 : 0) == 0 && wave_max((int)(hi - lo > 2 ? 3 : hi - lo)) <= 2) glued_miller_seq(acc, pxy, pinf, qxy, qinf, lo, hi, n_pairs, skip_infinity, odd);
  else if (KMAX == 2 && wave_max((int)(hi - lo > 2 ? 3 : hi - lo)) <= 2) glued_miller_upto2(acc, pxy, pinf, qxy, qinf, lo, hi, n_pairs, skip_infinity, odd);
  else glued_miller_chunks<KMAX>(acc, pxy, pinf, qxy, qinf, lo, hi, n_pairs, skip_infinity, odd);
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

// ------------------------------------------------------------------ glued pairings through line tables in HBM ---------------
// Jobs of two or more pairs (on average over the batch).  Sharing the squarings of one accumulator among k pairs needs k working G2 points next to the
// accumulator, which 256 registers do not hold (the in-register schedule above parks them in the stack frame and runs no faster than
// k separate loops).  So the work is cut where the data is smallest: PHASE A gives every (job, slot) its own lane pair, walks that
// pair's G2 point through the 87 steps with NO accumulator alive, and streams the lines -- already scaled by P: (l0, l1 P.y, l2 P.x),
// the operands of sparse_mul (pairing.rs:598) -- to HBM; PHASE B gives every job one lane pair that holds the accumulator and one line,
// and per step squares once and multiplies by the k lines it reads back.  Same field elements as the shared-squaring loop of
// glued_miller_loop (pairing.rs:970-1022): products commute and every value is an exact residue, so raw Miller values still match
// bit for bit.  A line is 27 R/N-class int32 digits per lane (three Fp2 coefficients x 9 limbs), stored as 7 x 16 bytes per lane,
// [line][slot][chunk][thread]: a wavefront writes / reads 1 KB contiguous per instruction.  19.5 KB per pair, written once and read
// once: 0.5 - 1 TB/s while these issue-bound kernels run, HBM the path otherwise leaves idle.
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));      // a native vector: loads keep their address space
constexpr int LT_CHUNKS = 7;
constexpr int LT_LINES = 87;
struct LineW { W2 l0, l4, l2; };
BN_DEV void line_put(u32x4* at, size_t stride, const LineW& L) {
  const i32 w[28] = {L.l0.c.v[0], L.l0.c.v[1], L.l0.c.v[2], L.l0.c.v[3], L.l0.c.v[4], L.l0.c.v[5], L.l0.c.v[6], L.l0.c.v[7], L.l0.c.v[8],
                     L.l4.c.v[0], L.l4.c.v[1], L.l4.c.v[2], L.l4.c.v[3], L.l4.c.v[4], L.l4.c.v[5], L.l4.c.v[6], L.l4.c.v[7], L.l4.c.v[8],
                     L.l2.c.v[0], L.l2.c.v[1], L.l2.c.v[2], L.l2.c.v[3], L.l2.c.v[4], L.l2.c.v[5], L.l2.c.v[6], L.l2.c.v[7], L.l2.c.v[8], 0};
#pragma unroll
  for (int c = 0; c < LT_CHUNKS; ++c) at[(size_t)c * stride] = u32x4{(u32)w[4 * c], (u32)w[4 * c + 1], (u32)w[4 * c + 2], (u32)w[4 * c + 3]};
}
template <class PTR>
BN_DEV LineW line_get(PTR at, size_t stride) {
  u32 w[28];
#pragma unroll
  for (int c = 0; c < LT_CHUNKS; ++c) {
    const u32x4 q = at[(size_t)c * stride];
    w[4 * c] = q.x; w[4 * c + 1] = q.y; w[4 * c + 2] = q.z; w[4 * c + 3] = q.w;
  }
  LineW L;
#pragma unroll
  for (int i = 0; i < 9; ++i) { L.l0.c.v[i] = (i32)w[i]; L.l4.c.v[i] = (i32)w[9 + i]; L.l2.c.v[i] = (i32)w[18 + i]; }
  return L;
}
// PHASE A: lane pair u = slot * jb + (job - job0) owns pair offsets[job] + slot.  A slot without a pair -- the job is shorter, or
// skip_infinity drops the pair (EIP-197: an identity on either side contributes 1) -- writes 87 unit lines.  In replay mode a G2
// identity walks through the formulas with Z = 0 exactly as in glued_miller_chunks.
template <bool ISO>
__global__ void HEAVY_BOUNDS k_pair_lines(const u64* pxy, const uint8_t* pinf, const u64* qxy, const uint8_t* qinf, const u64* offsets,
                                          size_t job0, size_t jb, size_t n_pairs, int kt, int skip_infinity, u32x4* table) {
  const size_t t = TID, u = pair_index(t);
  const int odd = pair_role(t);
  if (u >= (size_t)kt * jb) return;
  const size_t slot = u / jb, jl = u - slot * jb, job = job0 + jl;
  const size_t lo = offsets[job], hi = offsets[job + 1], idx = lo + slot;
  const bool exists = idx < hi;
  const bool pi = exists && pinf && pinf[idx], qi = exists && qinf && qinf[idx];
  const bool live = exists && !(skip_infinity && (pi || qi));
  const size_t stride = 2 * jb, line_step = (size_t)kt * LT_CHUNKS * stride;
  u32x4* at = table + slot * LT_CHUNKS * stride + 2 * jl + (size_t)odd;
  const W2 w_one = w2_from_s2(s2_one()), w_zero = W2{F29{{0, 0, 0, 0, 0, 0, 0, 0, 0}}};
  if (!live) {
    const LineW unit{w_one, w_zero, w_zero};
#pragma unroll 1
    for (int l = 0; l < LT_LINES; ++l) line_put(at + (size_t)l * line_step, stride, unit);
    return;
  }
  // ISO (jobs that end in a final exponentiation): the pair is walked on the isomorphic curves (bn254_pair29.hpp, g2_doubling_step29) -- its
  // lines are the reference's times factors in Fp*, which the final exponentiation kills; raw Miller values are built with ISO = false
  F29 px = f29_reduce(f29_from_fp(load_fp(pxy, n_pairs, idx, 0))), py = f29_reduce(f29_from_fp(load_fp(pxy, n_pairs, idx, 4)));
  S2 qxs = load_s2(qxy, n_pairs, idx, 0, odd), qys = load_s2(qxy, n_pairs, idx, 8, odd);
  W2 qx = w2_from_s2(qxs), qy = w2_from_s2(qys);
  if (ISO) {
    px = f29_mul(px, f29_iso_s2()); py = f29_mul(py, f29_iso_s3());
    qx = w2_scale(qx, f29_iso_s2()); qy = w2_scale(qy, f29_iso_s3());
    qxs = w2_to_s2(qx); qys = w2_to_s2(qy);
  }
  // G2Projective::from(&G2Affine): Z = infinity ? 0 : 1 (group.rs:506-517); the glued loop never looks at the flag again
  G2W r{qx, qy, qi ? w_zero : w_one};
  W2 l0, l1, l2;
  int line = 0;
  auto emit = [&]() {
    line_put(at + (size_t)line * line_step, stride, LineW{l0, w2_scale(l1, py), w2_scale(l2, px)});
    ++line;
  };
  const u64 nz = BN_ATE_NAF_NZ, ng = BN_ATE_NAF_NEG;
#pragma unroll 1
  for (int i = 0; i < 64; ++i) {
    g2_doubling_step29<ISO>(r, l0, l1, l2);
    emit();
    if ((nz >> (63 - i)) & 1) {
      g2_addition_step29(r, qx, ((ng >> (63 - i)) & 1) ? w2_neg(qy) : qy, l0, l1, l2);
      emit();
    }
  }
  S2 q1x, q1y, q2x, q2y;                              // endomorphism() returns self for the identity (g2.rs:141-143)
  g2_psi_affine(q1x, q1y, qxs, qys);
  g2_psi_affine(q2x, q2y, q1x, q1y);
  q1x = s2_select(q1x, qxs, qi); q1y = s2_select(q1y, qys, qi);
  q2x = s2_select(q2x, qxs, qi); q2y = s2_select(q2y, qys, qi);
  g2_addition_step29(r, w2_from_s2(q1x), w2_from_s2(q1y), l0, l1, l2);
  emit();
  g2_addition_step29(r, w2_from_s2(q2x), w2_from_s2(s2_neg(q2y)), l0, l1, l2);
  emit();
}
// G2Affine::precompute for a batch (pairing.rs:676-708) on lane pairs: the 87 line triples of every point as canonical words, SoA
// [87*24][n], triple t of point i at words 24t .. 24t+23 = (ell.0, ell.1, ell.2) as Fp2 each -- the reference's [Ell; 87] in its own
// order and with its own (unscaled) values: the same walk as k_pair_lines without a G1 point
__global__ void HEAVY_BOUNDS k_g2_precompute_pairs(const u64* qxy, u64* coeffs, size_t n) {
  const size_t t = TID, i = pair_index(t);
  const int odd = pair_role(t);
  if (i >= n) return;
  const S2 qxs = load_s2(qxy, n, i, 0, odd), qys = load_s2(qxy, n, i, 8, odd);
  const W2 qx = w2_from_s2(qxs), qy = w2_from_s2(qys);
  G2W r{qx, qy, w2_from_s2(s2_one())};
  W2 l0, l1, l2;
  int line = 0;
  auto emit = [&]() {
    store_s2(coeffs, n, i, 24 * line, odd, w2_to_s2(l0));
    store_s2(coeffs, n, i, 24 * line + 8, odd, w2_to_s2(w2_reduce(l1)));      // D-class differences: carry-normalise before leaving the core
    store_s2(coeffs, n, i, 24 * line + 16, odd, w2_to_s2(w2_reduce(l2)));
    ++line;
  };
  const u64 nz = BN_ATE_NAF_NZ, ng = BN_ATE_NAF_NEG;
#pragma unroll 1
  for (int it = 0; it < 64; ++it) {
    g2_doubling_step29(r, l0, l1, l2);
    emit();
    if ((nz >> (63 - it)) & 1) {
      g2_addition_step29(r, qx, ((ng >> (63 - it)) & 1) ? w2_neg(qy) : qy, l0, l1, l2);
      emit();
    }
  }
  S2 q1x, q1y, q2x, q2y;
  g2_psi_affine(q1x, q1y, qxs, qys);
  g2_psi_affine(q2x, q2y, q1x, q1y);
  g2_addition_step29(r, w2_from_s2(q1x), w2_from_s2(q1y), l0, l1, l2);
  emit();
  g2_addition_step29(r, w2_from_s2(q2x), w2_from_s2(s2_neg(q2y)), l0, l1, l2);
  emit();
}
// PHASE B: the raw glued Miller value of every job of the batch.  kw = the wavefront's largest slot count; a lane pair with fewer
// pairs reads the unit lines phase A wrote for its empty slots.  The accumulator is a LOCAL value (as a reference parameter it would
// live in the caller's frame and every leaf call would force it back to memory), the table pointer is re-qualified as global memory
// (through a call boundary it is a generic pointer: flat loads), and a line is loaded where it is used -- requesting it one
// operation ahead (27 more live registers across a line product) measured 1 % slower.
BN_NOINLINE void glued_miller_tables(W12& fout, const u32x4* at_generic, size_t stride, size_t line_step, int kw) {
  typedef const __attribute__((address_space(1))) u32x4* gptr;
  const gptr at = (gptr)at_generic;
  W12 f;
  {
    S12 one = s12_one();
    w12_from_s12(f, one);
  }
  const u64 nz = BN_ATE_NAF_NZ;
  int line = 0;
  auto lines = [&]() {
    const gptr row = at + (size_t)line * line_step;
    int sl = 0;
#pragma unroll 1
    for (; sl + 1 < kw; sl += 2) {      // two slots' lines first multiplied together (6 products), then into f (17: one coefficient of the merged line is zero): 23 instead of 26
      const LineW L = line_get(row + (size_t)sl * LT_CHUNKS * stride, stride);
      const LineW M = line_get(row + (size_t)(sl + 1) * LT_CHUNKS * stride, stride);
      const W12 ll = w12_line_product(L.l0, L.l4, L.l2, M.l0, M.l4, M.l2);
      f = w12_mul_line_pair(f, ll);
    }
    if (sl < kw) {
      const LineW L = line_get(row + (size_t)sl * LT_CHUNKS * stride, stride);
      f = w12_sparse_mul(f, L.l0, L.l4, L.l2);
    }
    ++line;
  };
#pragma unroll 1
  for (int it = 0; it < 64; ++it) {
    f = w12_sqr(f);
    lines();
    if ((nz >> (63 - it)) & 1) lines();
  }
  lines();
  lines();
  fout = f;
}
// fout: SoA stride n_out, job j of the batch at column out0 + j
__global__ void HEAVY_BOUNDS k_glued_from_tables(const u32x4* table, const u64* pxy, const uint8_t* pinf, const u64* qxy, const uint8_t* qinf,
                                                 const u64* offsets, size_t job0, size_t jb, size_t n_pairs, int kt, int skip_infinity,
                                                 u64* fout, size_t n_out, size_t out0) {
  const size_t t = TID, jl = pair_index(t);
  const int odd = pair_role(t);
  const bool active = jl < jb;                 // no early return: every lane takes part in the wave reductions
  const size_t job = job0 + (active ? jl : 0);
  const size_t lo = offsets[job], hi = offsets[job + 1];
  const size_t k = active ? hi - lo : 0;
  const int kw = wave_max((int)(k < (size_t)kt ? k : (size_t)kt));
  const size_t stride = 2 * jb;
  W12 acc;
  glued_miller_tables(acc, table + 2 * (active ? jl : 0) + (size_t)odd, stride, (size_t)kt * LT_CHUNKS * stride, kw);
  // pairs beyond the table's slots (a job much longer than the batch average): the in-register schedule, one more product
  if (wave_max(k > (size_t)kt ? 1 : 0)) {
    W12 rest;
    const size_t from = k > (size_t)kt ? lo + (size_t)kt : hi;
    glued_miller_chunks<KMAXW>(rest, pxy, pinf, qxy, qinf, from, active ? hi : from, n_pairs, skip_infinity, odd);
    w12_mul_nl(acc, acc, rest);
  }
  S12 fin;
  w12_to_s12(fin, acc);
  if (active) store_s12(fout, n_out, out0 + jl, odd, fin);
}
// PHASE C: final exponentiation of the batch's raw values (a kernel of its own: fused behind phase B it ran a third slower -- every
// wavefront of the one-round launch reaches the exponentiation, and its stack frame traffic, at the same moment)
__global__ void HEAVY_BOUNDS k_final_exp_jobs(const u64* fin, size_t n_in, size_t job0, size_t jb, size_t n_jobs, u64* gout, uint8_t* is_one) {
  const size_t t = TID, jl = pair_index(t);
  const int odd = pair_role(t);
  if (jl >= jb) return;
  S12 f, g;
  load_s12(f, fin, n_in, jl, odd);
  final_exponentiation29(g, f);
  if (gout) store_s12(gout, n_jobs, job0 + jl, odd, g);
  const bool one = s12_is_one(g);
  if (is_one && !odd) is_one[job0 + jl] = one ? 1 : 0;
}

// ------------------------------------------------------------------ one product over a whole batch ---------------------------
// glued_pairing over n pairs as ONE Gt (examples/verify_multiple_messages_same_signer.rs:41-60: 2n pairs, one final
// exponentiation, == identity).  The shared-squaring Miller value of a set of pairs is exactly the product of the per-pair Miller
// values ((prod f_i)^2 = prod f_i^2), so the batch is cut into chunks of KPROD pairs per lane pair (k_multi_pairing with
// raw_miller = 1), the chunk values are multiplied together by a log-depth tree of Fp12 products, and one lane pair runs the
// final exponentiation.
// range (device, NULL = [0, n_pairs)): the product covers pairs [range[0], range[1]) only -- one job of a multi-pairing batch
__global__ void k_chunk_offsets(u64* off, size_t n_jobs, size_t n_pairs, size_t chunk, const u64* range) {
  const size_t j = TID;
  if (j > n_jobs) return;
  const size_t lo = range ? (size_t)range[0] : 0, hi = range ? (size_t)range[1] : n_pairs;
  const size_t v = lo + j * chunk;
  off[j] = v < hi ? v : (hi > lo ? hi : lo);
}
// out[i] = in[2 i] * in[2 i + 1] (the odd tail is copied), SoA strides n_in / n_out
__global__ void HEAVY_BOUNDS k_fp12_tree_level(const u64* in, size_t n_in, u64* out, size_t n_out) {
  const size_t t = TID, i = pair_index(t);
  const int odd = pair_role(t);
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
// the last levels of a product tree (m <= BLOCK values, SoA stride `stride`, multiplied IN PLACE: value i absorbs value i + h, a level per
// barrier instead of a launch per level); the product leaves in `out` with stride 1
__global__ void HEAVY_BOUNDS k_fp12_tree_tail(u64* vals, size_t stride, size_t m, u64* out) {
  const size_t i = pair_index(threadIdx.x);
  const int odd = pair_role(threadIdx.x);
  while (m > 1) {
    const size_t h = (m + 1) / 2;
    if (i + h < m) {
      S12 a, b;
      load_s12(a, vals, stride, i, odd);
      load_s12(b, vals, stride, i + h, odd);
      W12 x, y, r;
      w12_from_s12(x, a);
      w12_from_s12(y, b);
      w12_mul_nl(r, x, y);
      w12_to_s12(a, r);
      store_s12(vals, stride, i, odd, a);
    }
    __threadfence_block();
    __syncthreads();
    m = h;
  }
  if (i == 0) {
    S12 a;
    load_s12(a, vals, stride, 0, odd);
    store_s12(out, 1, 0, odd, a);
  }
}
// f = 1 (the empty Miller product), SoA stride 1
__global__ void k_fp12_set_one(u64* out) {
  if (pair_index(TID) != 0) return;
  store_s12(out, 1, 0, pair_role(TID), s12_one());
}
// out = a * b, all SoA stride 1 (two raw Miller products)
__global__ void HEAVY_BOUNDS k_fp12_mul_pair(const u64* a, const u64* b, u64* out) {
  const size_t t = TID;
  const int odd = pair_role(t);
  if (pair_index(t) != 0) return;
  S12 sa, sb;
  load_s12(sa, a, 1, 0, odd);
  load_s12(sb, b, 1, 0, odd);
  W12 x, y, r;
  w12_from_s12(x, sa);
  w12_from_s12(y, sb);
  w12_mul_nl(r, x, y);
  w12_to_s12(sa, r);
  store_s12(out, 1, 0, odd, sa);
}
// column `col` of a G2 pair array (stride `stride`) <- the generator (src NULL) or element 0 of a one-key array
__global__ void k_g2_set_column(u64* qxy, uint8_t* qinf, size_t stride, size_t col, const u64* src_xy, const uint8_t* src_inf) {
  const size_t t = TID;
  const int odd = pair_role(t);
  if (pair_index(t) != 0) return;
  store_s2(qxy, stride, col, 0, odd, src_xy ? load_s2(src_xy, 1, 0, 0, odd) : s2_g2gen_x());
  store_s2(qxy, stride, col, 8, odd, src_xy ? load_s2(src_xy, 1, 0, 8, odd) : s2_g2gen_y());
  if (!odd) qinf[col] = (src_xy && src_inf && src_inf[0]) ? 1 : 0;
}
// The Miller value of ONE pair up to a factor in Fp* (isomorphic curves: it only ever feeds a final exponentiation; SoA stride-1 views of
// P, Q and the output), launched as <<<1, 64>>>: miller_loop29_wide.  An identity
// on either side gives 1 (the skip_infinity reading; the reference-replay reading of a G2 identity stays on the generic kernels).
__global__ void HEAVY_BOUNDS k_miller_single_wide(const u64* pxy, const uint8_t* pinf, const u64* qxy, const uint8_t* qinf, size_t stride, const u64* range, u64* fout) {
  __shared__ WideLds lds;
  const size_t t = TID;
  const int odd = pair_role(t);
  const size_t i = range ? (size_t)range[0] : 0;
  const bool empty = range && range[1] <= range[0];
  S12 f;
  if (empty || (pinf && pinf[i]) || (qinf && qinf[i])) {
    f = s12_one();
  } else {
    const Fp px = load_fp(pxy, stride, i, 0), py = load_fp(pxy, stride, i, 4);
    const S2 qx = load_s2(qxy, stride, i, 0, odd), qy = load_s2(qxy, stride, i, 8, odd);
    miller_loop29_wide<true>(f, px, py, qx, qy, &lds);
  }
  if (pair_index(t) == 0) store_s12(fout, 1, 0, odd, f);
}
// Small batches, one WAVEFRONT per EPW elements (EPW = 1: grid = n blocks of 64; EPW = 2: lanes 0-31 / 32-63 of a block hold one element
// each): element e < n is pair e of set A, element n + e pair e of set B (B optional).  qxy NULL = the G2 generator for every pair of that
// set.  An identity on either side gives 1.  ISO: Miller values up to factors in Fp* (isomorphic curves), SoA stride n: input of the
// k_final_exp_wide_* kernels only; ISO = false: the reference's raw Miller value (sylow_hip_miller_loop_batch at small sizes).  The loop runs on whatever the rows hold and the identity is selected afterwards, so that both halves of
// a wavefront reach every barrier together.
template <int EPW, bool ISO = true>
__global__ void HEAVY_BOUNDS k_miller_wide_batch(const u64* pa, const uint8_t* pa_inf, const u64* qa, const uint8_t* qa_inf, u64* fa,
                                                 const u64* pb, const uint8_t* pb_inf, const u64* qb, const uint8_t* qb_inf, u64* fb, size_t n, int bcast_b = 0) {
  __shared__ WideLds lds[EPW];                      // bcast_b: set B's G2 point is ONE point (a 1-element array) for every pair -- the same signer
  const size_t total = pb ? 2 * n : n;
  const int half = EPW == 2 ? (int)(threadIdx.x >> 5) : 0;
  const size_t e0 = (size_t)EPW * blockIdx.x + (size_t)half;
  const bool live = e0 < total;
  const size_t e = live ? e0 : total - 1;
  const bool second = e >= n;
  const size_t i = second ? e - n : e;
  const u64 *pxy = second ? pb : pa, *qxy = second ? qb : qa;
  const uint8_t *pinf = second ? pb_inf : pa_inf, *qinf = second ? qb_inf : qa_inf;
  u64* fout = second ? fb : fa;
  const int odd = pair_role(threadIdx.x);
  const bool one_q = second && bcast_b;
  const size_t qn = one_q ? 1 : n, qi = one_q ? 0 : i;
  const bool ident = (pinf && pinf[i]) || (qinf && qinf[qi]);
  S12 f;
  if (EPW == 1) {
    if (ident) {
      f = s12_one();
    } else {
      const Fp px = load_fp(pxy, n, i, 0), py = load_fp(pxy, n, i, 4);
      const S2 qx = qxy ? load_s2(qxy, qn, qi, 0, odd) : s2_g2gen_x(), qy = qxy ? load_s2(qxy, qn, qi, 8, odd) : s2_g2gen_y();
      miller_loop29_wide<ISO, 1>(f, px, py, qx, qy, &lds[0]);
    }
  } else {
    const Fp px = load_fp(pxy, n, i, 0), py = load_fp(pxy, n, i, 4);
    const S2 qx = qxy ? load_s2(qxy, qn, qi, 0, odd) : s2_g2gen_x(), qy = qxy ? load_s2(qxy, qn, qi, 8, odd) : s2_g2gen_y();
    miller_loop29_wide<ISO, EPW>(f, px, py, qx, qy, &lds[half]);
    if (ident) f = s12_one();
  }
  if (live && wide_j<EPW>((int)(threadIdx.x & 63u)) == 0) store_s12(fout, n, i, odd, f);
}
// final_exponentiation(fa_b * fb_b) (fb optional), one wavefront per EPW elements: Gt values (SoA stride n) and / or "== identity" flags
template <int EPW>
__global__ void HEAVY_BOUNDS k_final_exp_wide_batch(const u64* fa, const u64* fb, size_t n, u64* gout, uint8_t* is_one) {
  __shared__ WideLds lds[EPW];
  const int half = EPW == 2 ? (int)(threadIdx.x >> 5) : 0;
  const size_t e0 = (size_t)EPW * blockIdx.x + (size_t)half;
  const bool live = e0 < n;
  const size_t i = live ? e0 : n - 1;
  const int odd = pair_role(threadIdx.x);
  S12 f, g;
  load_s12(f, fa, n, i, odd);
  if (fb) {
    S12 h;
    load_s12(h, fb, n, i, odd);
    W12 x, y, r;
    w12_from_s12(x, f);
    w12_from_s12(y, h);
    w12_mul_wide_nl<EPW>(r, x, y, &lds[half]);
    w12_to_s12(f, r);
  }
  final_exponentiation29_wide<EPW>(g, f, &lds[half]);
  if (!live || wide_j<EPW>((int)(threadIdx.x & 63u)) != 0) return;
  if (gout) store_s12(gout, n, i, odd, g);
  const bool one = s12_is_one(g);
  if (is_one && !odd) is_one[i] = one ? 1 : 0;
}
// final_exponentiation(prod of the raw values of job j's pairs), one wavefront per job: raw SoA stride n_pairs (k_miller_wide_batch),
// job j owns pairs [offsets[j], offsets[j + 1]) (an empty job is the identity); Gt values (SoA stride n_jobs) and / or flags
__global__ void HEAVY_BOUNDS k_final_exp_wide_jobs(const u64* raw, size_t n_pairs, const u64* offsets, size_t n_jobs, u64* gout, uint8_t* is_one) {
  __shared__ WideLds lds;
  const size_t j = blockIdx.x;
  const int odd = pair_role(threadIdx.x);
  const size_t lo = offsets[j], hi = offsets[j + 1];
  S12 f, g;
  if (hi <= lo) {
    f = s12_one();
  } else {
    load_s12(f, raw, n_pairs, lo, odd);
    if (hi - lo > 1) {
      W12 acc, y;
      w12_from_s12(acc, f);
#pragma unroll 1
      for (size_t i = lo + 1; i < hi; ++i) {
        S12 h;
        load_s12(h, raw, n_pairs, i, odd);
        w12_from_s12(y, h);
        w12_mul_wide_nl(acc, acc, y, &lds);
      }
      w12_to_s12(f, acc);
    }
  }
  final_exponentiation29_wide(g, f, &lds);
  if (pair_index(threadIdx.x) != 0) return;
  if (gout) store_s12(gout, n_jobs, j, odd, g);
  const bool one = s12_is_one(g);
  if (is_one && !odd) is_one[j] = one ? 1 : 0;
}
// ONE element, launched as <<<1, 64>>>: all 32 lane pairs of the wavefront hold it and share the squarings of the hard part
// (final_exponentiation29_wide); wide = 0: lane pair 0 alone (the plain routine, the other lanes leave)
__global__ void HEAVY_BOUNDS k_final_exp_flag(const u64* fin, size_t n_in, u64* gout, uint8_t* is_one, int wide) {
  __shared__ WideLds lds;
  const size_t t = TID;
  const int odd = pair_role(t);
  if (!wide && pair_index(t) != 0) return;
  S12 f, g;
  if (n_in) load_s12(f, fin, n_in, 0, odd); else f = s12_one();      // empty product = identity (pairing.rs:1218-1219)
  if (wide) final_exponentiation29_wide(g, f, &lds); else final_exponentiation29(g, f);
  if (pair_index(t) != 0) return;
  if (gout) store_s12(gout, 1, 0, odd, g);
  const bool one = s12_is_one(g);
  if (is_one && !odd) is_one[0] = one ? 1 : 0;
}

// ------------------------------------------------------------------ Miller loops against precomputed line tables ------------
// G2PreComputed::miller_loop(&G1Affine) (pairing.rs:590-619) and glued_miller_loop(&[G2PreComputed], &[G1Affine])
// (pairing.rs:970-1022) consuming tables a host cached from sylow_hip_g2_precompute_batch: coeffs is the canonical SoA array
// [87*24][m] (triple t of table k = words 24t..24t+23 = ell.0, ell.1, ell.2), pair i reads table tab_idx[i] (or table i when
// tab_idx is NULL).  No G2 arithmetic: per step one shared squaring and, per pair, three coefficient loads, two scalings by the
// G1 coordinates and one sparse product.  Canonical words enter the carry-free core directly: the 29-bit digits of x times
// R'^2 mod p in one carry-free product (R' = 2^261), output N-class.
BN_DEV F29 f29_from_plain(const Fp& x) {
  F29 d;
  d.v[0] = (i32)(x.v[0] & BN_M29);
#pragma unroll
  for (int i = 1; i < 9; ++i) {
    const int bit = 29 * i, w = bit >> 5, sh = bit & 31;
    const u32 lo = x.v[w] >> sh;
    const u32 hi = (sh > 3 && w + 1 < 8) ? (x.v[w + 1] << (32 - sh)) : 0u;
    d.v[i] = (i32)((lo | hi) & BN_M29);
  }
  const F29 rr{{0x059bac10, 0x0d1503a3, 0x018016b8, 0x10ab0ca8, 0x02632639, 0x02c0169f, 0x169bfd53, 0x11869d4c, 0x002a11a6}};
  return f29_mul_leaf(W_ARGS(d), W_ARGS(rr));
}
// job j owns pairs [offsets[j], offsets[j+1]) (offsets NULL: job j = pair j); wave-uniform schedule, a lane pair whose job has
// fewer pairs than the wavefront maximum multiplies by the unit line
__global__ void HEAVY_BOUNDS k_miller_precomputed(const u64* coeffs, size_t m, const u64* tab_idx, const u64* pxy, size_t n_pairs,
                                                  const u64* offsets, size_t n_jobs, u64* fout) {
  const size_t t = TID, job = pair_index(t);
  const int odd = pair_role(t);
  const bool active = job < n_jobs;
  const size_t lo = !active ? 0 : offsets ? offsets[job] : job, hi = !active ? 0 : offsets ? offsets[job + 1] : job + 1;
  const int kw = wave_max((int)(hi - lo));
  const W2 w_one = w2_from_s2(s2_one()), w_zero = W2{F29{{0, 0, 0, 0, 0, 0, 0, 0, 0}}};
  W12 f;
  {
    S12 one = s12_one();
    w12_from_s12(f, one);
  }
  const u64 nz = BN_ATE_NAF_NZ;
  int idx = 0;
  // the first pair's G1 coordinates stay in registers (the plain, one-pair-per-job loop never reloads them)
  const bool ld0 = lo < hi && n_pairs != 0;
  const F29 px0 = ld0 ? f29_from_plain(load_plain(pxy, n_pairs, lo, 0)) : w_one.c;
  const F29 py0 = ld0 ? f29_from_plain(load_plain(pxy, n_pairs, lo, 4)) : w_one.c;
  auto lines = [&]() {
#pragma unroll 1
    for (int j = 0; j < kw; ++j) {
      const bool live = lo + (size_t)j < hi;
      const size_t pi = live ? lo + (size_t)j : 0;
      const bool ld = live && n_pairs != 0;
      const size_t tb = !ld ? 0 : tab_idx ? (size_t)tab_idx[pi] : pi;
      const F29 px = j == 0 ? px0 : ld ? f29_from_plain(load_plain(pxy, n_pairs, pi, 0)) : w_one.c;
      const F29 py = j == 0 ? py0 : ld ? f29_from_plain(load_plain(pxy, n_pairs, pi, 4)) : w_one.c;
      const int w0 = 24 * idx + 4 * odd;
      const W2 l0 = ld ? W2{f29_from_plain(load_plain(coeffs, m, tb, w0))} : w_one;
      const W2 l1 = ld ? W2{f29_from_plain(load_plain(coeffs, m, tb, w0 + 8))} : w_zero;
      const W2 l2 = ld ? W2{f29_from_plain(load_plain(coeffs, m, tb, w0 + 16))} : w_zero;
      f = w12_sparse_mul(f, w2_select(w_one, l0, live), w2_select(w_zero, w2_scale(l1, py), live), w2_select(w_zero, w2_scale(l2, px), live));
    }
    ++idx;
  };
#pragma unroll 1
  for (int it = 0; it < 64; ++it) {
    f = w12_sqr(f);
    lines();
    if ((nz >> (63 - it)) & 1) lines();
  }
  lines();
  lines();
  S12 fs;
  w12_to_s12(fs, f);
  if (active) store_s12(fout, n_jobs, job, odd, fs);
}
}  // namespace plk

__global__ void __launch_bounds__(BLOCK) k_evm_pair_finalize(const uint8_t* pst, const u64* offsets, size_t n_jobs, const uint8_t* is_one, uint8_t* result, uint8_t* status) {
  size_t j = TID;
  if (j >= n_jobs) return;
  uint8_t st = SYLOW_HIP_ST_OK;
  for (u64 k = offsets[j]; k < offsets[j + 1]; ++k) if (!st && pst[k]) st = pst[k];     // first failing pair, like the `?` in run_pair
  status[j] = st;
  result[j] = st ? 0 : is_one[j];
}

// Jobs of two or more pairs on average: lines to HBM, then the table-driven loop, then the final exponentiations (see k_pair_lines).
// Job batches share one leased workspace, reused in stream order: 19.5 KB per slot and job, batches of whole GPU rounds (2^16 lane
// pairs) up to TBL_BYTES.  SYLOW_HIP_OPT_MULTI_TABLES = 0 / 1 forces the in-register / the table route for every job size (A/B runs).
static int multi_tables_mode() { return (int)host::option(SYLOW_HIP_OPT_MULTI_TABLES); }
// Bytes the line tables of one call may take: the host's bound (sylow_hip_set_scratch_limit), or by default a quarter of the device memory
// that was FREE when this device's first multi-pair call arrived, at most 12 GB (an empty MI355X: 12 GB, nine rounds of two-slot jobs; a GPU
// that is shared and mostly full: proportionally less -- never a constant that ignores the other tenants).  hipMemGetInfo is asked once per
// device (it costs tens of microseconds; the small aggregate verifications are 2 ms calls).
static size_t table_budget() {
  const size_t lim = host::scratch_limit();
  if (lim) return lim;
  static std::atomic<size_t> cache[64];
  int d = 0;
  if (hipGetDevice(&d) != hipSuccess || d < 0 || d >= 64) { (void)hipGetLastError(); return (size_t)12 << 30; }
  size_t v = cache[d].load(std::memory_order_relaxed);
  if (!v) {
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) { (void)hipGetLastError(); free_b = (size_t)48 << 30; }
    v = free_b / 4;
    if (v > ((size_t)12 << 30)) v = (size_t)12 << 30;
    if (v < ((size_t)64 << 20)) v = (size_t)64 << 20;
    cache[d].store(v, std::memory_order_relaxed);
  }
  return v;
}
// bytes of line tables one job of kt slots takes (k_pair_lines' layout)
static size_t table_bytes_per_job(size_t kt) { return (size_t)plk::LT_LINES * kt * plk::LT_CHUNKS * 2 * sizeof(plk::u32x4); }
static size_t table_slots(size_t n_jobs, size_t n_pairs) {       // slots per job: the batch average, rounded up, 1..8; longer jobs take the in-register tail
  size_t kt = n_jobs ? (n_pairs + n_jobs - 1) / n_jobs : 1;
  return kt < 1 ? 1 : kt > 8 ? 8 : kt;
}
// The table route slices a batch into blocks of at least 1024 jobs: a budget below 1024 jobs' tables (20 - 160 MB by the job size) cannot be
// honoured by it, so such a batch takes the in-register schedule (no table at all) -- the bound a host sets is never silently exceeded.
static bool use_tables(size_t n_jobs, size_t n_pairs) {
  const int m = multi_tables_mode();
  if (m == 0) return false;
  if (m != 1 && n_pairs < 2 * n_jobs) return false;
  if (!n_pairs) return false;
  const size_t need = table_bytes_per_job(table_slots(n_jobs, n_pairs)) * (n_jobs < 1024 ? n_jobs : 1024);
  return need <= table_budget();
}
// SYLOW_HIP_OPT_WIDE_TAIL = 0: the single final exponentiation of the one-boolean shapes on one lane pair (default: on the whole wavefront)
static int wide_tail() { return host::option(SYLOW_HIP_OPT_WIDE_TAIL) == 0 ? 0 : 1; }
// SYLOW_HIP_OPT_WIDE_PACK = t: the small-batch kernels put two elements on a wavefront above t elements (0 = never, 1 = always: A/B runs and
// tests/test_gpu_routes.py).  Default: the number of compute units -- one wavefront per CU is the fastest shape (1.72 ms per pairing up
// to 256), a second wavefront on a CU already costs more (1.9 ms) than a wavefront of two elements (1.76 ms), and from there on the
// packed form has half the wavefronts: 1024 pairings 1.91 against 2.11 ms, 2048 2.15 against 3.74 ms, 4096 3.8 against 4.25 ms.
static size_t wide_pack() {
  const long long o = host::option(SYLOW_HIP_OPT_WIDE_PACK);
  if (o >= 0) return (size_t)o;
  const unsigned cus = host::compute_units();
  return (size_t)(cus ? cus : 256);
}
// Two elements per wavefront (16 lane pairs each) cost a second product pass in the dense Fp12 products of the final exponentiation only.
static void launch_miller_wide(const u64* pa, const uint8_t* pa_inf, const u64* qa, const uint8_t* qa_inf, u64* fa,
                               const u64* pb, const uint8_t* pb_inf, const u64* qb, const uint8_t* qb_inf, u64* fb, size_t n, hipStream_t st, int bcast_b = 0) {
  const size_t units = pb ? 2 * n : n;
  if (wide_pack() && units > wide_pack())
    plk::k_miller_wide_batch<2><<<dim3((unsigned)((units + 1) / 2)), dim3(64), 0, st>>>(pa, pa_inf, qa, qa_inf, fa, pb, pb_inf, qb, qb_inf, fb, n, bcast_b);
  else
    plk::k_miller_wide_batch<1><<<dim3((unsigned)units), dim3(64), 0, st>>>(pa, pa_inf, qa, qa_inf, fa, pb, pb_inf, qb, qb_inf, fb, n, bcast_b);
}
static void launch_final_exp_wide(const u64* fa, const u64* fb, size_t n, u64* gout, uint8_t* is_one, hipStream_t st) {
  if (wide_pack() && n > wide_pack())
    plk::k_final_exp_wide_batch<2><<<dim3((unsigned)((n + 1) / 2)), dim3(64), 0, st>>>(fa, fb, n, gout, is_one);
  else
    plk::k_final_exp_wide_batch<1><<<dim3((unsigned)n), dim3(64), 0, st>>>(fa, fb, n, gout, is_one);
}
// eq_i = [ a_i == b_i ] for Gt values (SoA stride n, canonical limbs)
__global__ void __launch_bounds__(BLOCK) k_gt_eq_flags(const u64* a, const u64* b, size_t n, uint8_t* eq) {
  const size_t i = TID;
  if (i >= n) return;
  u64 d = 0;
#pragma unroll 8
  for (int w = 0; w < 48; ++w) d |= a[(size_t)w * n + i] ^ b[(size_t)w * n + i];
  eq[i] = d == 0 ? 1 : 0;
}
namespace plkh {
// Small batches on one wavefront per one or two elements (k_miller_wide_batch / k_final_exp_wide_batch): up to this many pairings the
// latency route beats the one-lane-pair kernels (2048 resident wavefronts of two elements each, and one more half-round; docs/DESIGN_LOG.md R5-8.3)
// SYLOW_HIP_OPT_WIDE_MAX / _WIDE_VERIFY_MAX move the two caps (crossover runs, tools/dbg/time_small.py)
size_t wide_batch_max() {
  const size_t v = (size_t)host::option_or(SYLOW_HIP_OPT_WIDE_MAX, 0);
  return !wide_tail() ? 0 : v ? v : wide_pack() ? 6144 : 2048;      // 6144 pairings: three half-rounds of wavefronts, 3.6 against 4.2 ms; 7168: 4.6
}
// ... and up to this many verifications (2 n Miller loops + n final exponentiations; two rounds of wavefronts at the cap: 4096
// verifications 3.8 against 5.4 ms on the lane-pair kernel, 6144: 5.7 against 5.4)
size_t wide_verify_max() {
  const size_t v = (size_t)host::option_or(SYLOW_HIP_OPT_WIDE_VERIFY_MAX, 0);
  return !wide_tail() ? 0 : v ? v : wide_pack() ? 4096 : 1024;
}
// pairing(P_i, Q_i), i < n: raw values through `scratch` (48 n words), Gt values to gt_out (SoA stride n)
int32_t pairing_wide_batch(const uint64_t* p_xy, const uint8_t* p_inf, const uint64_t* q_xy, const uint8_t* q_inf, uint64_t* scratch, uint64_t* gt_out, size_t n, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  launch_miller_wide(p_xy, p_inf, q_xy, q_inf, scratch, nullptr, nullptr, nullptr, nullptr, nullptr, n, st);
  launch_final_exp_wide(scratch, nullptr, n, gt_out, nullptr, st);
  LAUNCHED();
}
// ok_i = [ e(sig_i, G2gen) e(hneg_i, pk_i) == 1 ], i < n; scratch: 96 n words
int32_t verify_wide_batch(const uint64_t* pk_xy, const uint8_t* pk_inf, const uint64_t* hneg, const uint8_t* hneg_inf, const uint64_t* sig_xy, const uint8_t* sig_inf,
                          uint64_t* scratch, uint8_t* ok, size_t n, void* stream, int one_key) {
  hipStream_t st = (hipStream_t)stream;
  u64 *fa = scratch, *fb = scratch + 48 * n;
  launch_miller_wide(sig_xy, sig_inf, nullptr, nullptr, fa, hneg, hneg_inf, pk_xy, pk_inf, fb, n, st, one_key);
  launch_final_exp_wide(fa, fb, n, nullptr, ok, st);
  LAUNCHED();
}
// the reference's raw Miller values (no identity flags: the raw entry point has none), i < n <= wide_batch_max()
int32_t miller_raw_wide_batch(const uint64_t* p_xy, const uint64_t* q_xy, uint64_t* f_out, size_t n, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  if (wide_pack() && n > wide_pack())
    plk::k_miller_wide_batch<2, false><<<dim3((unsigned)((n + 1) / 2)), dim3(64), 0, st>>>(p_xy, nullptr, q_xy, nullptr, f_out, nullptr, nullptr, nullptr, nullptr, nullptr, n);
  else
    plk::k_miller_wide_batch<1, false><<<dim3((unsigned)n), dim3(64), 0, st>>>(p_xy, nullptr, q_xy, nullptr, f_out, nullptr, nullptr, nullptr, nullptr, nullptr, n);
  LAUNCHED();
}
// final_exponentiation(f_i), i < n <= wide_batch_max()
int32_t final_exp_wide_batch(const uint64_t* f, uint64_t* gt_out, size_t n, void* stream) {
  launch_final_exp_wide(f, nullptr, n, gt_out, nullptr, (hipStream_t)stream);
  LAUNCHED();
}
// ok_i = [ pairing(sig_i, G2gen) == pairing(h_i, pk_i) ] evaluated literally (two final exponentiations, a comparison of Gt values);
// scratch: 192 n words
int32_t verify_two_pairings_wide_batch(const uint64_t* pk_xy, const uint8_t* pk_inf, const uint64_t* h, const uint8_t* h_inf, const uint64_t* sig_xy, const uint8_t* sig_inf,
                                       uint64_t* scratch, uint8_t* ok, size_t n, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  u64 *fa = scratch, *fb = scratch + 48 * n, *ga = scratch + 96 * n, *gb = scratch + 144 * n;
  launch_miller_wide(sig_xy, sig_inf, nullptr, nullptr, fa, h, h_inf, pk_xy, pk_inf, fb, n, st);
  launch_final_exp_wide(fa, nullptr, n, ga, nullptr, st);
  launch_final_exp_wide(fb, nullptr, n, gb, nullptr, st);
  k_gt_eq_flags<<<GRID(n)>>>(ga, gb, n, ok);
  LAUNCHED();
}
}  // namespace plkh
static bool single_job_route(size_t n_jobs, size_t n_pairs, int32_t skip_infinity);
static int32_t single_job_product(const uint64_t* p_xy, const uint8_t* p_inf, const uint64_t* q_xy, const uint8_t* q_inf, const uint64_t* pair_offsets,
                                  size_t n_jobs, size_t n_pairs, uint64_t* gt_out, uint8_t* is_one, void* stream);
static int32_t multi_pairing_tables(const uint64_t* p_xy, const uint8_t* p_inf, const uint64_t* q_xy, const uint8_t* q_inf, const uint64_t* pair_offsets,
                                    size_t n_jobs, size_t n_pairs, int32_t skip_infinity, uint64_t* gt_out, uint8_t* is_one, int raw_miller, int iso, void* stream) {
  // iso: the line tables may be built on the isomorphic curves (k_pair_lines<true>) -- whenever the value ends in a final exponentiation, here or
  // in the caller; 0 only where the reference's raw Miller value itself is the result (glued_miller_loop_batch)
  hipStream_t st = (hipStream_t)stream;
  const size_t kt = table_slots(n_jobs, n_pairs);
  constexpr size_t ROUND = 65536;
  const size_t TBL_BYTES = table_budget();
  const size_t per_job = table_bytes_per_job(kt);
  const size_t rounds = TBL_BYTES / (per_job * ROUND);
  // whole rounds of the GPU's 2^16 resident lane pairs while the budget allows; under a host-set limit below one round
  // (sylow_hip_set_scratch_limit) as many jobs as fit, in blocks of 1024 -- phase B / C then run under-filled, the price of the bound
  size_t slice = rounds >= 1 ? rounds * ROUND : (TBL_BYTES / per_job) & ~(size_t)1023;
  if (slice < 1024) slice = 1024;
  size_t jb_max = n_jobs < slice ? n_jobs : slice;
  size_t w_raw = raw_miller ? 0 : 48 * jb_max * sizeof(u64);
  host::Lease ws;
  int32_t rc = ws.acquire(per_job * jb_max + w_raw, st);
  if (rc != SYLOW_HIP_OK && jb_max > ROUND) {                  // a device short of memory: one round per batch
    (void)hipGetLastError();
    jb_max = ROUND;
    w_raw = raw_miller ? 0 : 48 * jb_max * sizeof(u64);
    rc = ws.acquire(per_job * jb_max + w_raw, st);
  }
  if (rc != SYLOW_HIP_OK) {                                    // no room for a table at all: the in-register schedule needs no workspace
    (void)hipGetLastError();
    plk::k_multi_pairing<plk::KMAXW><<<GRID(2 * n_jobs)>>>(p_xy, p_inf, q_xy, q_inf, pair_offsets, n_jobs, n_pairs, skip_infinity, gt_out, is_one, raw_miller); LAUNCHED();
  }
  plk::u32x4* table = (plk::u32x4*)ws.p;
  u64* raw = (u64*)((uint8_t*)ws.p + per_job * jb_max);
  for (size_t job0 = 0; job0 < n_jobs; job0 += jb_max) {
    const size_t jb = n_jobs - job0 < jb_max ? n_jobs - job0 : jb_max;
    if (!iso) plk::k_pair_lines<false><<<GRID(2 * kt * jb)>>>(p_xy, p_inf, q_xy, q_inf, pair_offsets, job0, jb, n_pairs, (int)kt, skip_infinity, table);
    else plk::k_pair_lines<true><<<GRID(2 * kt * jb)>>>(p_xy, p_inf, q_xy, q_inf, pair_offsets, job0, jb, n_pairs, (int)kt, skip_infinity, table);
    if (raw_miller) {
      plk::k_glued_from_tables<<<GRID(2 * jb)>>>(table, p_xy, p_inf, q_xy, q_inf, pair_offsets, job0, jb, n_pairs, (int)kt, skip_infinity, gt_out, n_jobs, job0);
    } else {
      plk::k_glued_from_tables<<<GRID(2 * jb)>>>(table, p_xy, p_inf, q_xy, q_inf, pair_offsets, job0, jb, n_pairs, (int)kt, skip_infinity, raw, jb, 0);
      plk::k_final_exp_jobs<<<GRID(2 * jb)>>>(raw, jb, job0, jb, n_jobs, gt_out, is_one);
    }
  }
  const hipError_t e = hipGetLastError();
  rc = ws.release();
  return e != hipSuccess ? host::fail(e, "kernel launch") : rc;
}

extern "C" {
int32_t sylow_hip_multi_pairing_batch(const uint64_t* p_xy, const uint8_t* p_inf, const uint64_t* q_xy, const uint8_t* q_inf,
                                      const uint64_t* pair_offsets, size_t n_jobs, size_t n_pairs, int32_t skip_infinity,
                                      uint64_t* gt_out, uint8_t* is_one, void* stream) {
  ARGCHK(pair_offsets && (gt_out || is_one) && (n_pairs == 0 || (p_xy && q_xy))); if (!n_jobs) return SYLOW_HIP_OK;
  if (single_job_route(n_jobs, n_pairs, skip_infinity)) return single_job_product(p_xy, p_inf, q_xy, q_inf, pair_offsets, n_jobs, n_pairs, gt_out, is_one, stream);
  if (use_tables(n_jobs, n_pairs)) return multi_pairing_tables(p_xy, p_inf, q_xy, q_inf, pair_offsets, n_jobs, n_pairs, skip_infinity, gt_out, is_one, 0, /*iso=*/1, stream);
  // chunks of KMAX pairs share the squarings; any KMAX is correct for any job size.  Batches that average at most two pairs per
  // job (the BLS / ecPairing k = 2 shape) take the two-slot instantiation: its pair states are a third of the stack frame
  if (n_pairs <= 2 * n_jobs) { plk::k_multi_pairing<2><<<GRID(2 * n_jobs)>>>(p_xy, p_inf, q_xy, q_inf, pair_offsets, n_jobs, n_pairs, skip_infinity, gt_out, is_one, 0); LAUNCHED(); }
  plk::k_multi_pairing<plk::KMAXW><<<GRID(2 * n_jobs)>>>(p_xy, p_inf, q_xy, q_inf, pair_offsets, n_jobs, n_pairs, skip_infinity, gt_out, is_one, 0); LAUNCHED();
}
int32_t sylow_hip_g2_precompute_batch(const uint64_t* q_xy, uint64_t* coeffs, size_t n, void* stream) {
  ARGCHK(q_xy && coeffs); if (!n) return SYLOW_HIP_OK;
  plk::k_g2_precompute_pairs<<<GRID(2 * n)>>>(q_xy, coeffs, n); LAUNCHED();
}
int32_t sylow_hip_glued_miller_loop_batch(const uint64_t* p_xy, const uint64_t* q_xy, const uint64_t* pair_offsets, size_t n_jobs, size_t n_pairs,
                                           uint64_t* f_out, void* stream) {
  ARGCHK(pair_offsets && f_out && (n_pairs == 0 || (p_xy && q_xy))); if (!n_jobs) return SYLOW_HIP_OK;
  if (use_tables(n_jobs, n_pairs)) return multi_pairing_tables(p_xy, nullptr, q_xy, nullptr, pair_offsets, n_jobs, n_pairs, 0, f_out, nullptr, 1, /*iso=*/0, stream);
  plk::k_multi_pairing<plk::KMAXW><<<GRID(2 * n_jobs)>>>(p_xy, nullptr, q_xy, nullptr, pair_offsets, n_jobs, n_pairs, 0, f_out, nullptr, 1); LAUNCHED();
}

int32_t sylow_hip_miller_loop_precomputed_batch(const uint64_t* coeffs, size_t n_tables, const uint64_t* table_idx, const uint64_t* p_xy,
                                                uint64_t* f_out, size_t n, void* stream) {
  ARGCHK(coeffs && p_xy && f_out && n_tables && (table_idx || n_tables == n)); if (!n) return SYLOW_HIP_OK;
  plk::k_miller_precomputed<<<GRID(2 * n)>>>(coeffs, n_tables, table_idx, p_xy, n, nullptr, n, f_out); LAUNCHED();
}
int32_t sylow_hip_glued_miller_loop_precomputed_batch(const uint64_t* coeffs, size_t n_tables, const uint64_t* table_idx, const uint64_t* p_xy,
                                                      const uint64_t* pair_offsets, size_t n_jobs, size_t n_pairs, uint64_t* f_out, void* stream) {
  ARGCHK(pair_offsets && f_out && (n_pairs == 0 || (coeffs && p_xy && n_tables && (table_idx || n_tables == n_pairs)))); if (!n_jobs) return SYLOW_HIP_OK;
  plk::k_miller_precomputed<<<GRID(2 * n_jobs)>>>(coeffs, n_tables, table_idx, p_xy, n_pairs, pair_offsets, n_jobs, f_out); LAUNCHED();
}

// Chunked Miller loops + product tree: leaves ONE raw Miller product (SoA stride 1 = 48 contiguous words) in the leased workspace.
// n_pairs > 0.  The caller releases the lease after enqueueing whatever consumes *result.
static int32_t miller_product_tree(const uint64_t* p_xy, const uint8_t* p_inf, const uint64_t* q_xy, const uint8_t* q_inf, size_t n_pairs,
                                   int32_t skip_infinity, host::Lease& ws, u64** result, void* stream, const u64* range = nullptr) {
  hipStream_t st = (hipStream_t)stream;
  // pairs per lane pair: as few as keep the whole product inside ONE round of the GPU (2^16 lane pairs resident), at most KPROD --
  // a small product is latency-bound (one pair per lane pair on the single-pair loop: one Miller loop deep), a large one
  // throughput-bound (shared squarings)
  size_t chunk = (n_pairs + 65535) / 65536;
  if (chunk > (size_t)plk::KPROD) chunk = plk::KPROD;
  const size_t n_jobs = (n_pairs + chunk - 1) / chunk;
  // workspace: chunk offsets + two ping-pong buffers of Fp12 values
  const size_t n_off = (n_jobs + 2) & ~(size_t)1, n_a = 48 * n_jobs, n_b = 48 * ((n_jobs + 1) / 2);
  int32_t rc = ws.acquire((n_off + n_a + n_b) * sizeof(u64), st);
  if (rc != SYLOW_HIP_OK) return rc;
  u64 *off = (u64*)ws.p, *bufa = off + n_off, *bufb = bufa + n_a;
  // ONE pair (the collapsed halves of the aggregate verifiers): pure latency on one lane pair -- the whole wavefront takes it
  if (n_pairs == 1 && skip_infinity && wide_tail()) {
    plk::k_miller_single_wide<<<1, 64, 0, st>>>(p_xy, p_inf, q_xy, q_inf, 1, range, bufa);
    *result = bufa;
    return SYLOW_HIP_OK;
  }
  // FEW pairs (small aggregate verifications, short products): one pair per lane pair would be one Miller loop deep on lone wavefronts
  // (2.5 ms) -- a wavefront per one or two pairs instead (0.6 ms), then the same product tree over the n_pairs values
  if (chunk == 1 && !range && skip_infinity && n_pairs <= plkh::wide_batch_max()) {
    launch_miller_wide(p_xy, p_inf, q_xy, q_inf, bufa, nullptr, nullptr, nullptr, nullptr, nullptr, n_pairs, st);
  } else {
  plk::k_chunk_offsets<<<GRID(n_jobs + 1)>>>(off, n_jobs, n_pairs, chunk, range);
  // chunks of two or more pairs: lines to HBM + the table-driven loop (SYLOW_HIP_OPT_MULTI_TABLES = 0: the in-register KPROD-slot schedule)
  const size_t round_table = (size_t)65536 * chunk * plk::LT_LINES * plk::LT_CHUNKS * 2 * sizeof(plk::u32x4);     // one round of chunk-slot jobs
  if (chunk >= 2 && multi_tables_mode() != 0 && (round_table <= table_budget() || multi_tables_mode() == 1)) {
    rc = multi_pairing_tables(p_xy, p_inf, q_xy, q_inf, off, n_jobs, n_pairs, skip_infinity, bufa, nullptr, 1, /*iso=*/1, stream);
    if (rc != SYLOW_HIP_OK) return rc;
  }
  else if (chunk <= 2) plk::k_multi_pairing<plk::KMAXW><<<GRID(2 * n_jobs)>>>(p_xy, p_inf, q_xy, q_inf, off, n_jobs, n_pairs, skip_infinity, bufa, nullptr, 1);
  else plk::k_multi_pairing<plk::KPROD><<<GRID(2 * n_jobs)>>>(p_xy, p_inf, q_xy, q_inf, off, n_jobs, n_pairs, skip_infinity, bufa, nullptr, 1);
  }
  u64 *cur = bufa, *nxt = bufb;
  size_t m = n_jobs;
  while (m > (size_t)BLOCK) {
    const size_t h = (m + 1) / 2;
    plk::k_fp12_tree_level<<<GRID(2 * h)>>>(cur, m, nxt, h);
    u64* tmp = cur; cur = nxt; nxt = tmp;
    m = h;
  }
  if (m > 1) {                                     // the rest of the tree in one block; the product lands in the other buffer, stride 1
    plk::k_fp12_tree_tail<<<1, BLOCK, 0, st>>>(cur, m, m, nxt);
    cur = nxt;
  }
  *result = cur;
  return SYLOW_HIP_OK;
}
static int32_t finish(host::Lease& ws) {
  const hipError_t e = hipGetLastError();
  const int32_t rc = ws.release();
  return e != hipSuccess ? host::fail(e, "kernel launch") : rc;
}
// FEW jobs with few pairs (a single ecPairing call, a single Groth16-style check, a handful of them): one wavefront per pair for the
// Miller loops, one per job for the product of its pairs and the final exponentiation -- 1.1 + 1.3 ms of latency whatever the job size,
// against a whole glued loop and a final exponentiation on one lane pair per job (one job of 4 pairs: 7.9 ms).  The jobs' pair ranges
// stay on the device.  EIP-197 reading of identities only (skip_infinity).
static bool single_job_route(size_t n_jobs, size_t n_pairs, int32_t skip_infinity) {
  const size_t cap = plkh::wide_batch_max();
  return cap != 0 && n_pairs >= 1 && n_pairs <= cap && n_jobs <= 1024 && skip_infinity;
}
static int32_t single_job_product(const uint64_t* p_xy, const uint8_t* p_inf, const uint64_t* q_xy, const uint8_t* q_inf, const uint64_t* pair_offsets,
                                  size_t n_jobs, size_t n_pairs, uint64_t* gt_out, uint8_t* is_one, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  host::Lease ws;
  if (n_jobs == 1 && n_pairs > 256) {               // one LONG job: a log-depth product tree instead of a chain of products on one wavefront
    u64* prod = nullptr;
    int32_t rc = miller_product_tree(p_xy, p_inf, q_xy, q_inf, n_pairs, 1, ws, &prod, stream, pair_offsets);
    if (rc != SYLOW_HIP_OK) return rc;
    plk::k_final_exp_flag<<<1, 64, 0, st>>>(prod, 1, gt_out, is_one, wide_tail());
    return finish(ws);
  }
  int32_t rc = ws.acquire(48 * n_pairs * sizeof(u64), st);
  if (rc != SYLOW_HIP_OK) return rc;
  u64* raw = (u64*)ws.p;
  launch_miller_wide(p_xy, p_inf, q_xy, q_inf, raw, nullptr, nullptr, nullptr, nullptr, nullptr, n_pairs, st);
  plk::k_final_exp_wide_jobs<<<dim3((unsigned)n_jobs), dim3(64), 0, st>>>(raw, n_pairs, pair_offsets, n_jobs, gt_out, is_one);
  return finish(ws);
}
int32_t sylow_hip_pairing_product_batch(const uint64_t* p_xy, const uint8_t* p_inf, const uint64_t* q_xy, const uint8_t* q_inf,
                                        size_t n_pairs, int32_t skip_infinity, uint64_t* gt_out, uint8_t* is_one, void* stream) {
  ARGCHK((gt_out || is_one) && (n_pairs == 0 || (p_xy && q_xy)));
  hipStream_t st = (hipStream_t)stream;
  if (n_pairs == 0) { plk::k_final_exp_flag<<<1, 64, 0, st>>>(nullptr, 0, gt_out, is_one, wide_tail()); LAUNCHED(); }
  host::Lease ws;
  if (n_pairs >= 2 && n_pairs <= 256 && skip_infinity && plkh::wide_batch_max() != 0) {
    // a short product: one wavefront per Miller loop, then one wavefront multiplies the values and exponentiates (2.0 - 3.0 ms against 3.3)
    int32_t rc = ws.acquire((48 * n_pairs + 2) * sizeof(u64), st);
    if (rc != SYLOW_HIP_OK) return rc;
    u64 *off = (u64*)ws.p, *raw = off + 2;
    plk::k_chunk_offsets<<<1, 64, 0, st>>>(off, 1, n_pairs, n_pairs, nullptr);
    launch_miller_wide(p_xy, p_inf, q_xy, q_inf, raw, nullptr, nullptr, nullptr, nullptr, nullptr, n_pairs, st);
    plk::k_final_exp_wide_jobs<<<1, 64, 0, st>>>(raw, n_pairs, off, 1, gt_out, is_one);
    return finish(ws);
  }
  u64* prod = nullptr;
  int32_t rc = miller_product_tree(p_xy, p_inf, q_xy, q_inf, n_pairs, skip_infinity, ws, &prod, stream);
  if (rc != SYLOW_HIP_OK) return rc;
  plk::k_final_exp_flag<<<1, 64, 0, st>>>(prod, 1, gt_out, is_one, wide_tail());
  return finish(ws);
}
int32_t sylow_hip_pairing_product_partial_batch(const uint64_t* p_xy, const uint8_t* p_inf, const uint64_t* q_xy, const uint8_t* q_inf,
                                                size_t n_pairs, int32_t skip_infinity, uint64_t* f_out, void* stream) {
  ARGCHK(f_out && (n_pairs == 0 || (p_xy && q_xy)));
  hipStream_t st = (hipStream_t)stream;
  if (n_pairs == 0) { plk::k_fp12_set_one<<<1, 64, 0, st>>>(f_out); LAUNCHED(); }
  host::Lease ws;
  u64* prod = nullptr;
  int32_t rc = miller_product_tree(p_xy, p_inf, q_xy, q_inf, n_pairs, skip_infinity, ws, &prod, stream);
  if (rc != SYLOW_HIP_OK) return rc;
  hipError_t e = hipMemcpyAsync(f_out, prod, 48 * sizeof(u64), hipMemcpyDeviceToDevice, st);
  rc = finish(ws);
  return e != hipSuccess ? host::fail(e, "hipMemcpyAsync(partial product)") : rc;
}
// Aggregate verification (examples/verify_multiple_messages_same_signer.rs:41-60, threshold_signing.rs:92-121): the reference
// glues the 2n pairs (sig_i, G2gen), (-H(m_i), pk_i) into one product and compares it with the identity.  Bilinearity collapses the
// G2gen half: prod_i e(sig_i, G2gen) = e(sum_i sig_i, G2gen) -- n additions in G1 instead of n Miller loops -- and, for ONE key, the
// other half too: prod_i e(-H(m_i), pk) = e(-sum_i H(m_i), pk).  The Gt value is the same group element either way, so gt_out and
// the boolean are the reference's.  This entry leaves the shard's raw Miller product (for the cross-GPU aggregate, collective.hip).
// `weights` (NULL = none): w_i as Fp values [4][n]; the product becomes prod_i [e(sig_i, G2gen) e(-H(m_i), pk_i)]^(w_i) = e(sum w_i sig_i, G2gen)
// prod_i e(-w_i H(m_i), pk_i) -- the small-exponent batch test (SURVEY.md e1 "alternative aggregate check"): with weights drawn after
// the signatures are fixed, a batch that contains an invalid signature passes with probability at most 2^-(bits of the weights).
// A short-lived side stream for work that does not depend on the long kernels of the caller's stream (here: the sum of the
// signatures and the one Miller loop it feeds run beside the batch's hashing instead of after it).  open(): the side stream waits for
// everything the caller's stream holds at this point; join(): the caller's stream waits for the side work.  Any failure to create the
// stream or its events degrades to the caller's stream (same results, no overlap); SYLOW_HIP_OPT_AGG_FORK = 0 forces that.
using host::Fork;
static int32_t aggregate_partial(const uint64_t* pk_xy, const uint8_t* pk_inf, size_t n_pk, const uint8_t* msgs, const uint64_t* msg_offsets,
                                 const uint64_t* sig_xy, const uint8_t* sig_inf, const uint64_t* weights, size_t n, uint64_t* f_out, void* stream) {
  ARGCHK(f_out && (n == 0 || (pk_xy && msgs && msg_offsets && sig_xy && (n_pk == 1 || n_pk == n))));
  hipStream_t st = (hipStream_t)stream;
  if (n == 0) { plk::k_fp12_set_one<<<1, 64, 0, st>>>(f_out); LAUNCHED(); }
  const bool one_key = n_pk == 1 && n != 1;
  // scratch (u64 words): H or -H [8][n], the summation tree(s) [12][n], the two collapsed pairs G1 [8] + [8] / G2 [16] + [16], flags
  const size_t w_h = 8 * n, w_acc = 12 * n, w_flags = (2 * n + 4 + 7) / 8, w_sw = weights ? 8 * n : 0;
  host::Lease ws;
  int32_t rc = ws.acquire((w_h + (one_key ? 2 : 1) * w_acc + w_sw + 16 + 32 + w_flags) * sizeof(u64), st);
  if (rc != SYLOW_HIP_OK) return rc;
  u64 *hxy = (u64*)ws.p, *acc = hxy + w_h, *acc2 = acc + (one_key ? w_acc : 0), *sw = acc2 + w_acc, *pa2 = sw + w_sw, *pb2 = pa2 + 8, *qa2 = pb2 + 8, *qb2 = qa2 + 16;
  uint8_t *hinf = (uint8_t*)(qb2 + 16), *p2inf = hinf + n, *q2inf = p2inf + 2, *swinf = q2inf + 2;
  host::Lease wa, wb;
  u64 *pa = nullptr, *pb = nullptr;
  // The G2gen half -- e(sum_i sig_i, G2gen), or e(sum_i w_i sig_i, G2gen) with weights: a (scalar multiplication,) summation tree and ONE
  // Miller loop, a few ms of pure latency -- depends on the signatures (and weights) only: it runs on a side stream beside the hashing.
  Fork fork;
  hipStream_t sd = fork.open(st, host::option(SYLOW_HIP_OPT_AGG_FORK) != 0);
  {
    const u64* sp = sig_xy;
    const uint8_t* spi = sig_inf;
    if (weights) {                                          // sig_i -> w_i sig_i (scratch)
      rc = sylow_hip_g1_scalar_mul_batch(sig_xy, sig_inf, weights, sw, swinf, n, sd);
      sp = sw; spi = swinf;
    }
    if (rc == SYLOW_HIP_OK) rc = g1h::sum(sp, spi, n, acc, pb2, p2inf + 1, 1, 0, 0, sd);
    if (rc == SYLOW_HIP_OK) {
      plk::k_g2_set_column<<<1, 64, 0, sd>>>(qb2, q2inf + 1, 1, 0, nullptr, nullptr);
      rc = miller_product_tree(pb2, p2inf + 1, qb2, q2inf + 1, 1, 1, wb, &pb, sd);
    }
  }
  // one key and no weights: only the SUM of the hashes is needed -- they go projective straight into the summation tree's array
  const bool hash_into_tree = one_key && !weights;
  if (rc == SYLOW_HIP_OK) rc = hash_into_tree ? g1h::hash_to_g1_proj(msgs, msg_offsets, acc2, n, stream)
                                              : g1h::hash_to_g1(msgs, msg_offsets, hxy, hinf, n, /*negate=*/one_key ? 0 : 1, stream);
  if (rc == SYLOW_HIP_OK && weights) rc = sylow_hip_g1_scalar_mul_batch(hxy, hinf, weights, hxy, hinf, n, stream);      // H_i <- w_i H_i (in place)
  if (rc == SYLOW_HIP_OK && !one_key) {
    // prod_i e(-H_i, pk_i) over the batch
    rc = miller_product_tree(hxy, hinf, pk_xy, pk_inf, n, 1, wa, &pa, stream);
  } else if (rc == SYLOW_HIP_OK) {
    // one key: the other half collapses too -- e(-sum H, pk), one more single-pair loop
    rc = hash_into_tree ? g1h::sum_tree(acc2, n, pa2, p2inf, 1, 0, /*negate=*/1, stream)
                        : g1h::sum(hxy, hinf, n, acc2, pa2, p2inf, 1, 0, /*negate=*/1, stream);
    if (rc == SYLOW_HIP_OK) {
      plk::k_g2_set_column<<<1, 64, 0, st>>>(qa2, q2inf, 1, 0, pk_xy, pk_inf);
      rc = miller_product_tree(pa2, p2inf, qa2, q2inf, 1, 1, wa, &pa, stream);
    }
  }
  const int32_t rj = fork.join(st);
  if (rc == SYLOW_HIP_OK) rc = rj;
  if (rc == SYLOW_HIP_OK) plk::k_fp12_mul_pair<<<1, 64, 0, st>>>(pa, pb, f_out);
  const hipError_t e = hipGetLastError();
  if (wb.slot >= 0) wb.st = st;      // the caller's stream has joined the side stream and still reads the block: its release is ordered there
  const int32_t r1 = wa.release(), r2 = wb.release(), r3 = ws.release();
  if (rc != SYLOW_HIP_OK) return rc;
  if (e != hipSuccess) return host::fail(e, "kernel launch");
  return r1 != SYLOW_HIP_OK ? r1 : r2 != SYLOW_HIP_OK ? r2 : r3;
}
int32_t sylow_hip_bls_aggregate_partial_batch(const uint64_t* pk_xy, const uint8_t* pk_inf, size_t n_pk, const uint8_t* msgs, const uint64_t* msg_offsets,
                                              const uint64_t* sig_xy, const uint8_t* sig_inf, size_t n, uint64_t* f_out, void* stream) {
  return aggregate_partial(pk_xy, pk_inf, n_pk, msgs, msg_offsets, sig_xy, sig_inf, nullptr, n, f_out, stream);
}
int32_t sylow_hip_bls_weighted_partial_batch(const uint64_t* pk_xy, const uint8_t* pk_inf, size_t n_pk, const uint8_t* msgs, const uint64_t* msg_offsets,
                                             const uint64_t* sig_xy, const uint8_t* sig_inf, const uint64_t* weights, size_t n, uint64_t* f_out, void* stream) {
  ARGCHK(weights || !n);
  return aggregate_partial(pk_xy, pk_inf, n_pk, msgs, msg_offsets, sig_xy, sig_inf, weights, n, f_out, stream);
}
int32_t sylow_hip_fp12_product_final_exp(const uint64_t* parts, size_t k, uint64_t* gt_out, uint8_t* is_one, void* stream) {
  ARGCHK((gt_out || is_one) && (parts || !k));
  hipStream_t st = (hipStream_t)stream;
  if (k <= 1) { plk::k_final_exp_flag<<<1, 64, 0, st>>>(parts, k, gt_out, is_one, wide_tail()); LAUNCHED(); }
  host::Lease ws;
  const size_t n_a = 48 * ((k + 1) / 2), n_b = 48 * ((k + 3) / 4);
  int32_t rc = ws.acquire((n_a + n_b) * sizeof(u64), st);
  if (rc != SYLOW_HIP_OK) return rc;
  const u64* cur = parts;
  u64 *nxt = (u64*)ws.p, *other = nxt + n_a;
  size_t m = k;
  bool own = false;                                 // the caller's array is never multiplied in place: one out-of-place level first
  while (m > 1 && (m > (size_t)BLOCK || !own)) {
    const size_t h = (m + 1) / 2;
    plk::k_fp12_tree_level<<<GRID(2 * h)>>>(cur, m, nxt, h);
    cur = nxt; u64* tmp = nxt; nxt = other; other = tmp;
    m = h;
    own = true;
  }
  if (m > 1) {
    plk::k_fp12_tree_tail<<<1, BLOCK, 0, st>>>((u64*)cur, m, m, nxt);
    cur = nxt;
  }
  plk::k_final_exp_flag<<<1, 64, 0, st>>>(cur, 1, gt_out, is_one, wide_tail());
  return finish(ws);
}

int32_t sylow_hip_evm_ecpairing_batch(const uint8_t* in, const uint64_t* pair_offsets, size_t n_jobs, size_t n_pairs,
                                      uint8_t* result, uint8_t* status, void* stream) {
  ARGCHK(pair_offsets && result && status && (in || !n_pairs)); if (!n_jobs) return SYLOW_HIP_OK;
  hipStream_t st = (hipStream_t)stream;
  const size_t np = n_pairs ? n_pairs : 1;
  // workspace: decoded SoA points, flags, per-pair status, per-job product flag
  const size_t bytes = np * (8 + 16) * 8 + 3 * np + n_jobs + 64;
  host::Lease lease;
  int32_t rc = lease.acquire(bytes, st);
  if (rc != SYLOW_HIP_OK) return rc;
  uint8_t* ws = (uint8_t*)lease.p;
  u64* pxy = (u64*)ws;
  u64* qxy = pxy + 8 * np;
  uint8_t* pinf = (uint8_t*)(qxy + 16 * np);
  uint8_t* qinf = pinf + np;
  uint8_t* pst = qinf + np;
  uint8_t* isone = pst + np;
  if (n_pairs) rc = plkh::evm_decode_pairs(in, n_pairs, pxy, pinf, qxy, qinf, pst, stream);
  if (rc == SYLOW_HIP_OK) {
    if (single_job_route(n_jobs, n_pairs, 1)) rc = single_job_product(pxy, pinf, qxy, qinf, pair_offsets, n_jobs, n_pairs, nullptr, isone, stream);
    else if (use_tables(n_jobs, n_pairs)) rc = multi_pairing_tables(pxy, pinf, qxy, qinf, pair_offsets, n_jobs, n_pairs, /*skip_infinity=*/1, nullptr, isone, 0, /*iso=*/1, stream);
    else if (n_pairs <= 2 * n_jobs) plk::k_multi_pairing<2><<<GRID(2 * n_jobs)>>>(pxy, pinf, qxy, qinf, pair_offsets, n_jobs, n_pairs, /*skip_infinity=*/1, nullptr, isone, 0);
    else plk::k_multi_pairing<plk::KMAXW><<<GRID(2 * n_jobs)>>>(pxy, pinf, qxy, qinf, pair_offsets, n_jobs, n_pairs, /*skip_infinity=*/1, nullptr, isone, 0);
  }
  if (rc == SYLOW_HIP_OK) k_evm_pair_finalize<<<GRID(n_jobs)>>>(pst, pair_offsets, n_jobs, isone, result, status);
  const int32_t rc2 = finish(lease);
  return rc != SYLOW_HIP_OK ? rc : rc2;
}
}  // extern "C"
