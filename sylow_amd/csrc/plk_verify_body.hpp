// plk_verify_body.hpp -- the fused BLS check e(sig, G2gen) e(-H(m), pk) == 1 as a device routine, shared by the two units that launch it:
// plk_verify.hip (one lane pair per element: the metric's second kernel) and plk_quad.hip (BN_QUAD 1: one lane quad per element, mid-size
// batches).  Element geometry comes from plk_common.hpp (elem_index / elem_writer / ELEMS_PER_BLOCK).
#pragma once
#include "plk_common.hpp"

namespace plk {
// block-cooperative copy of a line table into LDS
BN_DEV void stage_table(i32* lds, const i32* src) {
  for (int k = threadIdx.x; k < LINE_TABLE_WORDS; k += blockDim.x) lds[k] = src[k];
}
// coefficient c (0: l1 / l0, 1: l2 / l0) of line `at`, and its unit word
BN_DEV W2 table_w2(const i32* tab, int at, int c, int odd) {
  const i32* t = tab + ((at * 2 + c) * 2 + odd) * 9;
  return W2{F29{{t[0], t[1], t[2], t[3], t[4], t[5], t[6], t[7], t[8]}}};
}
BN_DEV i32 table_unit(const i32* tab, int at) { return tab[LINE_TABLE_LINES * 36 + at]; }

// e(sig, G2gen) * e(-H(msg), pk) == 1 with one shared-squaring Miller loop and one final exponentiation: the boolean of lib.rs:223-236
// (FE(a) == FE(b) <=> FE(a conj(b)) == 1, conj(miller(H, pk)) = miller(-H, pk)).  PK_TABLE: one public key for the whole batch, its lines
// precomputed.  -H(m_i) comes from k_hash_to_g1 (affine SoA hneg / hneg_inf): one Keccak expansion per element, and the hashing code's
// registers and stack frame stay out of this kernel (hashing inside it -- each lane of a pair mapping one field element -- measured 1 % slower).
template <bool PK_TABLE>
BN_DEV void bls_verify_fused_body(const u64* pkxy, const uint8_t* pkinf, const i32* pk_table,
                                  const u64* hneg, const uint8_t* hneg_inf,
                                  const u64* sigxy, const uint8_t* siginf, const i32* gen_table, uint8_t* okout, size_t n, size_t m, const Stagger& st) {
  __shared__ i32 tabA[LINE_TABLE_WORDS];
  __shared__ i32 tabB[PK_TABLE ? LINE_TABLE_WORDS : 1];
  // staggered launch (k_pairing, plk_pairing.hip): role 1 parks the two-pair Miller value, role 2 finishes a parked chunk
  unsigned chunk;
  int role = stagger_role(st, chunk);
  const size_t t = (size_t)chunk * blockDim.x + threadIdx.x, i = elem_index(t);
  const int odd = pair_role(t);
  const size_t np = (size_t)st.count * ELEMS_PER_BLOCK, ip = (size_t)(chunk - st.first) * ELEMS_PER_BLOCK + (i & (ELEMS_PER_BLOCK - 1));
  if (role == 2 && !stagger_wait(st, chunk)) role = 0;           // parked values not visible within the bound: recompute the chunk whole
  if (role == 2) {
    S12 fs, g;
    load_s12(fs, st.park, np, ip, odd);
    final_exponentiation29(g, fs);
    const bool one = s12_is_one(g);
    if (!odd && elem_writer(t)) okout[i] = one ? 1 : 0;
    return;
  }
  stage_table(tabA, gen_table);
  if (PK_TABLE) stage_table(tabB, pk_table);
  __syncthreads();
  const bool active = i < m;                 // n = the arrays' SoA stride, m <= n = the elements of this launch
  const size_t ii = active ? i : 0;          // out-of-range lanes recompute element 0 (uniform control flow), store nothing
  const Fp hxs = load_fp(hneg, n, ii, 0), hys = load_fp(hneg, n, ii, 4);     // pair B is (-H, pk)
  const bool hinf = hneg_inf[ii] != 0;
  // Loop invariants that are read once or twice per step live in LDS, [limb][thread] (see miller_loop29g): the signature's and
  // -H(m)'s coordinates for the line scalings and, without a key table, the key's for the addition steps.
  __shared__ i32 lds[PK_TABLE ? 36 : 54][256];
  const bool liveA = !(siginf && siginf[ii]);
  const bool liveB = !(hinf || (pkinf && pkinf[PK_TABLE ? 0 : ii]));
  // A dead pair (a point at infinity on either side: its pairing is 1) keeps the loop's instruction stream and contributes nothing: its
  // G1 coordinates are stored as ZERO, so that the two line coefficients they scale vanish and the line degenerates to its constant
  // coefficient.  For pair A that coefficient is the table's small integer (replaced by 1 below).  For pair B without a key table it is
  // the Fp2 value l0 of the stepped point -- a dead pair B steps the generator, whose 87 constants are non-zero
  // (tests/test_oracle_kats.py::test_generator_line_constants_nonzero) -- and a factor in Fp2* changes neither the final exponentiation's
  // value nor this kernel's boolean (c^(p^6 - 1) = 1 for c in Fp6*).  No per-line selects, no unit / zero constants held across the loop.
  const F29 f29_zero{{0, 0, 0, 0, 0, 0, 0, 0, 0}};
  lds_put9(lds, 0, liveA ? f29_reduce(f29_from_fp(load_fp(sigxy, n, ii, 0))) : f29_zero);
  lds_put9(lds, 1, liveA ? f29_reduce(f29_from_fp(load_fp(sigxy, n, ii, 4))) : f29_zero);
  // Without a key table pair B runs on the isomorphic curves (bn254_pair29.hpp: g2_doubling_step29<ISO>): -H and the key go through phi, the
  // pair's Miller value picks up a factor in Fp* that the final exponentiation kills, and every doubling step saves its twist-constant product
  constexpr bool ISO = !PK_TABLE;
  lds_put9(lds, 2, liveB ? (ISO ? f29_mul(f29_reduce(f29_from_fp(hxs)), f29_iso_s2()) : f29_reduce(f29_from_fp(hxs))) : f29_zero);
  lds_put9(lds, 3, liveB ? (ISO ? f29_mul(f29_reduce(f29_from_fp(hys)), f29_iso_s3()) : f29_reduce(f29_from_fp(hys))) : f29_zero);
  auto SX = [&]() { return lds_get9(lds, 0); };
  auto SY = [&]() { return lds_get9(lds, 1); };
  auto HX = [&]() { return lds_get9(lds, 2); };
  auto HY = [&]() { return lds_get9(lds, 3); };
  // a dead pair B steps the generator instead (any curve point keeps the arithmetic defined) and multiplies by the unit line
  auto key_x = [&]() { return (PK_TABLE || !liveB) ? s2_g2gen_x() : load_s2(pkxy, n, ii, 0, odd); };
  auto key_y = [&]() { return (PK_TABLE || !liveB) ? s2_g2gen_y() : load_s2(pkxy, n, ii, 8, odd); };
  G2W r;
  {
    W2 qx = w2_from_s2(key_x()), qy = w2_from_s2(key_y());
    if (ISO) W2_SCALE2(qx, qy, qx, f29_iso_s2(), qy, f29_iso_s3());
    if (!PK_TABLE) { lds_put9(lds, 4, qx.c); lds_put9(lds, 5, qy.c); }
    r = G2W{qx, qy, w2_from_s2(s2_one())};
  }
  auto QX = [&]() { return W2{lds_get9(lds, PK_TABLE ? 0 : 4)}; };
  auto QY = [&](bool neg) {
    const W2 y{lds_get9(lds, PK_TABLE ? 0 : 5)};
    return neg ? w2_neg(y) : y;                                             // -Q: a D-class product operand
  };
  W12 f;
  {
    S12 one = s12_one();
    w12_from_s12(f, one);
  }
  W2 l0, l1, l2;
  int idx = 0;
  auto lineA = [&]() {
    W2 a1, a2;
    W2_SCALE2(a1, a2, table_w2(tabA, idx, 0, odd), SY(), table_w2(tabA, idx, 1, odd), SX());
    f = w12_sparse_mul_unit(f, liveA ? table_unit(tabA, idx) : 1, a1, a2);
  };
  auto lineB = [&]() {
    if (PK_TABLE) {
      W2 b1, b2;
      W2_SCALE2(b1, b2, table_w2(tabB, idx, 0, odd), HY(), table_w2(tabB, idx, 1, odd), HX());
      f = w12_sparse_mul_unit(f, liveB ? table_unit(tabB, idx) : 1, b1, b2);
    } else {
      W2 s1, s2;
      W2_SCALE2(s1, s2, l1, HY(), l2, HX());
      f = w12_sparse_mul(f, l0, s1, s2);
    }
  };
  const u64 nz = BN_ATE_NAF_NZ, ng = BN_ATE_NAF_NEG;
#pragma unroll 1
  for (int it = 0; it < 64; ++it) {
    f = w12_sqr(f);
    lineA();
    if (!PK_TABLE) g2_doubling_step29<ISO>(r, l0, l1, l2);
    lineB();
    ++idx;
    if ((nz >> (63 - it)) & 1) {
      lineA();
      if (!PK_TABLE) g2_addition_step29(r, QX(), QY(((ng >> (63 - it)) & 1) != 0), l0, l1, l2);
      lineB();
      ++idx;
    }
  }
  S2 q1x, q1y, q2x, q2y;
  if (!PK_TABLE) { g2_psi_affine(q1x, q1y, w2_to_s2(QX()), w2_to_s2(QY(false))); g2_psi_affine(q2x, q2y, q1x, q1y); }   // phi(key) from LDS: no live registers across the loop
  lineA();
  if (!PK_TABLE) g2_addition_step29(r, w2_from_s2(q1x), w2_from_s2(q1y), l0, l1, l2);
  lineB();
  ++idx;
  lineA();
  if (!PK_TABLE) g2_addition_step29(r, w2_from_s2(q2x), w2_from_s2(s2_neg(q2y)), l0, l1, l2);
  lineB();
  S12 fs, g;
  w12_to_s12(fs, f);
  if (role == 1) {                                         // whole chunks only: every lane is active
    store_s12(st.park, np, ip, odd, fs);
    stagger_publish(st, chunk);
    return;
  }
  final_exponentiation29(g, fs);
  const bool one = s12_is_one(g);
  if (active && !odd && elem_writer(t)) okout[i] = one ? 1 : 0;
}
}  // namespace plk
