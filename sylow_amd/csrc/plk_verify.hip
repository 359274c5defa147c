// plk_verify.hip -- BLS verification on lane pairs: lib.rs:223-236 as written (two pairings), the fused two-pair check with one
// final exponentiation, the same-signer shape with both line tables in LDS, and the line-table builder (G2Affine::precompute
// of one point in the carry-free lane-pair layout).
#include "plk_common.hpp"
#include "plk_verify_body.hpp"

namespace plk {
// ------------------------------------------------------------------ hash to G1 on a lane pair ---------------------------
// g1.rs:307-331: map(u0) + map(u1).  The even lane maps u0, the odd lane u1 (the two SvdW maps are independent), the points
// are exchanged and both lanes finish with the same complete addition and affine normalisation.
BN_DEV bool hash_to_g1_pair(Fp& hx, Fp& hy, bool& hinf, const uint8_t* msg, size_t msg_len, const DstPrime& dp) {
  const bool odd = lane_odd();
  u64 em[12], half[6];
  expand_message_xmd96_words(em, msg, msg_len, dp);
#pragma unroll
  for (int k = 0; k < 6; ++k) half[k] = odd ? em[6 + k] : em[k];
  const Fp u = fp_from_be48_words(half);
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
// Layout and normalisation: plk_common.hpp (LINE_TABLE_WORDS).  One block of LINES_BLOCK threads: lane pair 0 walks the point
// through the 87 steps of G2Affine::precompute (pairing.rs:676-708) and leaves the raw lines in LDS, then lane pair j divides
// line j by its first coefficient (87 independent Fp2 inversions side by side instead of 87 in a row).
// The generator (qxy == nullptr) or element idx of an SoA G2 array.
constexpr int LINES_BLOCK = 192;
__global__ void __launch_bounds__(LINES_BLOCK) k_g2_lines29(const u64* qxy, size_t n, size_t idx, i32* table) {
  __shared__ i32 raw[LINE_TABLE_LINES][3][2][9];
  const int odd = pair_role(threadIdx.x);
  if (pair_index(threadIdx.x) == 0) {
    S2 qxs = s2_g2gen_x(), qys = s2_g2gen_y();
    if (qxy) { qxs = load_s2(qxy, n, idx, 0, odd); qys = load_s2(qxy, n, idx, 8, odd); }
    const W2 qx = w2_from_s2(qxs), qy = w2_from_s2(qys), nqy = w2_from_s2(s2_neg(qys));
    G2W r{qx, qy, w2_from_s2(s2_one())};
    W2 l0, l1, l2;
    int at = 0;
    auto put = [&]() {
      const W2 c[3] = {l0, w2_reduce(l1), w2_reduce(l2)};
      for (int k = 0; k < 3; ++k) for (int j = 0; j < 9; ++j) raw[at][k][odd][j] = c[k].c.v[j];
      ++at;
    };
    const u64 nz = BN_ATE_NAF_NZ, ng = BN_ATE_NAF_NEG;
#pragma unroll 1
    for (int i = 0; i < 64; ++i) {
      g2_doubling_step29(r, l0, l1, l2); put();
      if ((nz >> (63 - i)) & 1) { g2_addition_step29(r, qx, ((ng >> (63 - i)) & 1) ? nqy : qy, l0, l1, l2); put(); }
    }
    S2 q1x, q1y, q2x, q2y;
    g2_psi_affine(q1x, q1y, qxs, qys);
    g2_psi_affine(q2x, q2y, q1x, q1y);
    g2_addition_step29(r, w2_from_s2(q1x), w2_from_s2(q1y), l0, l1, l2); put();
    g2_addition_step29(r, w2_from_s2(q2x), w2_from_s2(s2_neg(q2y)), l0, l1, l2); put();
  }
  __syncthreads();
  const int line = (int)(pair_index(threadIdx.x));
  if (line >= LINE_TABLE_LINES) return;
  auto get = [&](int k) {
    const i32* t = raw[line][k][odd];
    return W2{F29{{t[0], t[1], t[2], t[3], t[4], t[5], t[6], t[7], t[8]}}};
  };
  const W2 l0 = get(0), l1 = get(1), l2 = get(2);
  u32 nzero = 0;                                           // R-class digits are unique: the value is zero iff every limb is
  for (int j = 0; j < 9; ++j) nzero |= (u32)l0.c.v[j];
  nzero |= swap_u32(nzero);
  const bool unit = nzero != 0;
  const W2 inv = w2_inv(l0);
  const W2 c[2] = {unit ? w2_reduce(w2_mul(l1, inv)) : l1, unit ? w2_reduce(w2_mul(l2, inv)) : l2};
  for (int k = 0; k < 2; ++k) for (int j = 0; j < 9; ++j) table[((line * 2 + k) * 2 + odd) * 9 + j] = c[k].c.v[j];
  if (!odd) table[LINE_TABLE_LINES * 36 + line] = unit ? 1 : 0;
}
// ------------------------------------------------------------------ BLS verification ----------------------------------------
// lib.rs:223-236 as written: pairing(sig, G2gen) == pairing(H(msg), pk), two Miller loops and two final exponentiations
__global__ void HEAVY_BOUNDS k_bls_verify(const u64* pkxy, const uint8_t* pkinf, const uint8_t* msgs, const u64* off, DstPrime dp,
                                          const u64* sigxy, const uint8_t* siginf, const i32* gen_table, uint8_t* okout, size_t n) {
  __shared__ i32 tabA[LINE_TABLE_WORDS];
  stage_table(tabA, gen_table);
  __syncthreads();
  const size_t t = TID, i = pair_index(t);
  const int odd = pair_role(t);
  if (i >= n) return;
  Fp hx, hy; bool hinf;
  hash_to_g1_pair(hx, hy, hinf, msgs + off[i], (size_t)(off[i + 1] - off[i]), dp);
  S12 lhs, rhs;
  if (siginf && siginf[i]) {
    lhs = s12_one();
  } else {
    // G2PreComputed::miller_loop (pairing.rs:590-619) against the generator's line table (lines divided by their first
    // coefficient: the value differs from the reference's raw Miller value by an Fp2 factor, the pairing does not)
    const F29 sx = f29_reduce(f29_from_fp(load_fp(sigxy, n, i, 0))), sy = f29_reduce(f29_from_fp(load_fp(sigxy, n, i, 4)));
    W12 f;
    {
      S12 one = s12_one();
      w12_from_s12(f, one);
    }
    const u64 nz = BN_ATE_NAF_NZ;
    int idx = 0;
    auto line = [&]() {
      f = w12_sparse_mul_unit(f, table_unit(tabA, idx), w2_scale(table_w2(tabA, idx, 0, odd), sy), w2_scale(table_w2(tabA, idx, 1, odd), sx));
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
    miller_loop29g<false>(f, hx, hy, qx, qy);
    final_exponentiation29(rhs, f);
  }
  bool eq = s2_eq(lhs.c0.c0, rhs.c0.c0) && s2_eq(lhs.c0.c1, rhs.c0.c1) && s2_eq(lhs.c0.c2, rhs.c0.c2) &&
            s2_eq(lhs.c1.c0, rhs.c1.c0) && s2_eq(lhs.c1.c1, rhs.c1.c1) && s2_eq(lhs.c1.c2, rhs.c1.c2);
  if (!odd) okout[i] = eq ? 1 : 0;
}

template <bool PK_TABLE>
__global__ void HEAVY_BOUNDS k_bls_verify_fused(const u64* pkxy, const uint8_t* pkinf, const i32* pk_table,
                                                const u64* hneg, const uint8_t* hneg_inf,
                                                const u64* sigxy, const uint8_t* siginf, const i32* gen_table, uint8_t* okout, size_t n, size_t m, Stagger st) {
  ClockProbe pb;
  probe_begin(pb, st.clk);
  bls_verify_fused_body<PK_TABLE>(pkxy, pkinf, pk_table, hneg, hneg_inf, sigxy, siginf, gen_table, okout, n, m, st);
  probe_end(pb, st.clk);
}
}  // namespace plk

namespace plkh {
size_t line_table_bytes() { return plk::LINE_TABLE_WORDS * sizeof(bn254::i32); }
int32_t build_lines29(const uint64_t* q_xy, size_t n, size_t idx, bn254::i32* table, void* stream) {
  plk::k_g2_lines29<<<1, plk::LINES_BLOCK, 0, (hipStream_t)stream>>>(q_xy, n, idx, table); LAUNCHED();
}
}  // namespace plkh

// The fused check as two launches: -H(m_i) for the batch (hash.hip: k_hash_to_g1, one element per lane, four wavefronts per SIMD, 64 bytes
// per element through a leased block), then the pairing kernel reading it.
template <bool PK_TABLE>
static int32_t launch_fused(const uint64_t* pk_xy, const uint8_t* pk_inf, const bn254::i32* pk_table, const uint8_t* msgs, const uint64_t* msg_offsets,
                            const DstPrime& dp, const uint64_t* sig_xy, const uint8_t* sig_inf, const bn254::i32* gen, uint8_t* ok, size_t n, void* stream) {
  host::Lease ws;
  int32_t rc = ws.acquire(8 * n * sizeof(u64) + n, (hipStream_t)stream);
  if (rc != SYLOW_HIP_OK) return rc;
  u64* hneg = (u64*)ws.p;
  uint8_t* hinf = (uint8_t*)(hneg + 8 * n);
  rc = g1h::hash_to_g1(msgs, msg_offsets, hneg, hinf, n, /*negate=*/1, stream);
  if (rc == SYLOW_HIP_OK && n <= plkh::quad_batch_max()) {      // mid-size batches: a lane quad per element (plk_quad.hip)
    rc = plkh::verify_fused_quad(PK_TABLE ? 1 : 0, pk_xy, pk_inf, pk_table, hneg, hinf, sig_xy, sig_inf, gen, ok, n, n, stream);
    const int32_t r2 = ws.release();
    return rc != SYLOW_HIP_OK ? rc : r2;
  }
  // whole rounds + a short tail: the tail on quads on a side stream beside the rounds (sylow_hip_pairing_batch, plk_pairing.hip); the side stream
  // forks here, behind the hashing kernel
  const size_t tail = rc == SYLOW_HIP_OK ? plkh::tail_split(n) : 0, m = n - tail;
  host::Fork fk;
  hipStream_t side = tail ? fk.open((hipStream_t)stream) : (hipStream_t)stream;
  plk::Stagger sg{0, 0, 0, 0, nullptr, nullptr, nullptr};
  host::Lease wp;
  const size_t nblk = (2 * m + BLOCK - 1) / BLOCK, full = (2 * m) / BLOCK;
  if (rc == SYLOW_HIP_OK) {
    const hipError_t es = plkh::stagger_setup(sg, wp, nblk, full, (hipStream_t)stream, plkh::blocks_per_cu(plk::k_bls_verify_fused<PK_TABLE>));
    if (es != hipSuccess) rc = host::fail(es, "stagger flags");
  }
  if (rc == SYLOW_HIP_OK)
    plk::k_bls_verify_fused<PK_TABLE><<<dim3((unsigned)(nblk + sg.count)), dim3(BLOCK), 0, (hipStream_t)stream>>>(pk_xy, pk_inf, pk_table, hneg, hinf, sig_xy, sig_inf, gen, ok, n, m, sg);
  hipError_t e = hipGetLastError();
  if (tail && rc == SYLOW_HIP_OK && e == hipSuccess) {
    rc = plkh::verify_fused_quad(PK_TABLE ? 1 : 0, (PK_TABLE || !pk_xy) ? pk_xy : pk_xy + m, (PK_TABLE || !pk_inf) ? pk_inf : pk_inf + m, pk_table, hneg + m, hinf + m,
                                 sig_xy + m, sig_inf ? sig_inf + m : nullptr, gen, ok + m, n, tail, side);
    const int32_t rj = fk.join((hipStream_t)stream);
    if (rc == SYLOW_HIP_OK) rc = rj;
  }
  const int32_t r3 = wp.release();
  const int32_t r2 = ws.release();
  if (rc != SYLOW_HIP_OK) return rc;
  return e != hipSuccess ? host::fail(e, "kernel launch") : (r2 != SYLOW_HIP_OK ? r2 : r3);
}

extern "C" {
// verify (lib.rs:223-236): pairing(sig, G2gen) == pairing(H(msg), pk).  The default entry point answers with ONE final
// exponentiation: FE(a) == FE(b) <=> FE(a conj(b)) == 1 (FE is a homomorphism onto unitary elements, FE(conj b) = FE(b)^-1), and
// conj(miller(H, pk)) = miller(-H, pk) line by line (negating P negates exactly the line's odd-in-w coefficient) -- for every
// input, in the subgroup or not.  The literal two-pairing evaluation stays available as *_two_pairings_batch.
static int32_t verify_one_final_exp(const uint64_t* pk_xy, const uint8_t* pk_inf, const uint8_t* msgs, const uint64_t* msg_offsets,
                                    const uint64_t* sig_xy, const uint8_t* sig_inf, uint8_t* ok, size_t n, void* stream) {
  ARGCHK(pk_xy && msgs && msg_offsets && sig_xy && ok); if (!n) return SYLOW_HIP_OK;
  // a few verifications are pure latency on one lane pair each (6.8 ms): the same product e(sig, G2gen) e(-H, pk) with the same reading of
  // identities on one wavefront per Miller loop and per final exponentiation (3 ms)
  if (n <= plkh::wide_verify_max()) {  // small batches: -H(m_i), then a wavefront per Miller loop and per final exponentiation
    host::Lease ws;
    int32_t rc = ws.acquire((8 + 96) * n * sizeof(u64) + n, (hipStream_t)stream);
    if (rc != SYLOW_HIP_OK) return rc;
    u64* hneg = (u64*)ws.p;
    u64* scratch = hneg + 8 * n;
    uint8_t* hinf = (uint8_t*)(scratch + 96 * n);
    rc = g1h::hash_to_g1(msgs, msg_offsets, hneg, hinf, n, /*negate=*/1, stream);
    if (rc == SYLOW_HIP_OK) rc = plkh::verify_wide_batch(pk_xy, pk_inf, hneg, hinf, sig_xy, sig_inf, scratch, ok, n, stream);
    const int32_t r2 = ws.release();
    return rc != SYLOW_HIP_OK ? rc : r2;
  }
  DstPrime dp; host::dst_arg(dp, nullptr, 0);
  const bn254::i32* gen = nullptr;
  int32_t rc = host::gen_lines29(&gen, (hipStream_t)stream);
  if (rc != SYLOW_HIP_OK) return rc;
  return launch_fused<false>(pk_xy, pk_inf, nullptr, msgs, msg_offsets, dp, sig_xy, sig_inf, gen, ok, n, stream);
}
int32_t sylow_hip_bls_verify_batch(const uint64_t* pk_xy, const uint8_t* pk_inf, const uint8_t* msgs, const uint64_t* msg_offsets,
                                   const uint64_t* sig_xy, const uint8_t* sig_inf, uint8_t* ok, size_t n, void* stream) {
  return verify_one_final_exp(pk_xy, pk_inf, msgs, msg_offsets, sig_xy, sig_inf, ok, n, stream);
}
int32_t sylow_hip_bls_verify_fused_batch(const uint64_t* pk_xy, const uint8_t* pk_inf, const uint8_t* msgs, const uint64_t* msg_offsets,
                                         const uint64_t* sig_xy, const uint8_t* sig_inf, uint8_t* ok, size_t n, void* stream) {
  return verify_one_final_exp(pk_xy, pk_inf, msgs, msg_offsets, sig_xy, sig_inf, ok, n, stream);
}
int32_t sylow_hip_bls_verify_two_pairings_batch(const uint64_t* pk_xy, const uint8_t* pk_inf, const uint8_t* msgs, const uint64_t* msg_offsets,
                                                const uint64_t* sig_xy, const uint8_t* sig_inf, uint8_t* ok, size_t n, void* stream) {
  ARGCHK(pk_xy && msgs && msg_offsets && sig_xy && ok); if (!n) return SYLOW_HIP_OK;
  if (n <= plkh::wide_verify_max()) {  // small batches: H(m_i), a wavefront per one or two Miller loops / final exponentiations, Gt values compared
    host::Lease ws;
    int32_t rc = ws.acquire((8 + 192) * n * sizeof(u64) + n, (hipStream_t)stream);
    if (rc != SYLOW_HIP_OK) return rc;
    u64* h = (u64*)ws.p;
    u64* scratch = h + 8 * n;
    uint8_t* hinf = (uint8_t*)(scratch + 192 * n);
    rc = g1h::hash_to_g1(msgs, msg_offsets, h, hinf, n, /*negate=*/0, stream);
    if (rc == SYLOW_HIP_OK) rc = plkh::verify_two_pairings_wide_batch(pk_xy, pk_inf, h, hinf, sig_xy, sig_inf, scratch, ok, n, stream);
    const int32_t r2 = ws.release();
    return rc != SYLOW_HIP_OK ? rc : r2;
  }
  DstPrime dp; host::dst_arg(dp, nullptr, 0);
  const bn254::i32* gen = nullptr;
  int32_t rc = host::gen_lines29(&gen, (hipStream_t)stream);
  if (rc != SYLOW_HIP_OK) return rc;
  plk::k_bls_verify<<<GRID(2 * n)>>>(pk_xy, pk_inf, msgs, msg_offsets, dp, sig_xy, sig_inf, gen, ok, n); LAUNCHED();
}
int32_t sylow_hip_bls_verify_same_signer_batch(const uint64_t* pk_xy, const uint8_t* pk_inf, const uint8_t* msgs, const uint64_t* msg_offsets,
                                               const uint64_t* sig_xy, const uint8_t* sig_inf, uint8_t* ok, size_t n, void* stream) {
  ARGCHK(pk_xy && msgs && msg_offsets && sig_xy && ok); if (!n) return SYLOW_HIP_OK;
  hipStream_t st = (hipStream_t)stream;
  if (n <= plkh::wide_verify_max()) {  // small batches: the one-wavefront route of bls_verify_batch with the one key read by every pair
    host::Lease wsm;
    int32_t rcs = wsm.acquire((8 + 96) * n * sizeof(u64) + n, st);
    if (rcs != SYLOW_HIP_OK) return rcs;
    u64* hneg = (u64*)wsm.p;
    u64* scratch = hneg + 8 * n;
    uint8_t* hinf = (uint8_t*)(scratch + 96 * n);
    rcs = g1h::hash_to_g1(msgs, msg_offsets, hneg, hinf, n, /*negate=*/1, stream);
    if (rcs == SYLOW_HIP_OK) rcs = plkh::verify_wide_batch(pk_xy, pk_inf, hneg, hinf, sig_xy, sig_inf, scratch, ok, n, stream, /*one_key=*/1);
    const int32_t r2s = wsm.release();
    return rcs != SYLOW_HIP_OK ? rcs : r2s;
  }
  DstPrime dp; host::dst_arg(dp, nullptr, 0);
  const bn254::i32* gen = nullptr;
  int32_t rc = host::gen_lines29(&gen, st);
  if (rc != SYLOW_HIP_OK) return rc;
  host::Lease ws;
  if ((rc = ws.acquire(plk::LINE_TABLE_WORDS * sizeof(bn254::i32), st)) != SYLOW_HIP_OK) return rc;
  bn254::i32* table = (bn254::i32*)ws.p;
  plk::k_g2_lines29<<<1, plk::LINES_BLOCK, 0, st>>>(pk_xy, 1, 0, table);     // the key is a 1-element SoA array
  rc = launch_fused<true>(pk_xy, pk_inf, table, msgs, msg_offsets, dp, sig_xy, sig_inf, gen, ok, n, stream);
  const int32_t r2 = ws.release();
  return rc != SYLOW_HIP_OK ? rc : r2;
}
// The same check against a line table the host cached for the key (sylow_hip_g2_line_table: `G2PreComputed` cached per pk,
// examples/verify_multiple_messages_same_signer.rs:41-60): no G2 arithmetic at all, nothing rebuilt per call.
int32_t sylow_hip_g2_line_table_words(void) { return plk::LINE_TABLE_WORDS; }
int32_t sylow_hip_g2_line_table(const uint64_t* q_xy, size_t n, size_t idx, int32_t* table, void* stream) {
  ARGCHK(q_xy && table && idx < n);
  return plkh::build_lines29(q_xy, n, idx, table, stream);
}
int32_t sylow_hip_bls_verify_line_table_batch(const int32_t* pk_table, const uint8_t* pk_inf, const uint8_t* msgs, const uint64_t* msg_offsets,
                                              const uint64_t* sig_xy, const uint8_t* sig_inf, uint8_t* ok, size_t n, void* stream) {
  ARGCHK(pk_table && msgs && msg_offsets && sig_xy && ok); if (!n) return SYLOW_HIP_OK;
  DstPrime dp; host::dst_arg(dp, nullptr, 0);
  const bn254::i32* gen = nullptr;
  int32_t rc = host::gen_lines29(&gen, (hipStream_t)stream);
  if (rc != SYLOW_HIP_OK) return rc;
  return launch_fused<true>(nullptr, pk_inf, pk_table, msgs, msg_offsets, dp, sig_xy, sig_inf, gen, ok, n, stream);
}
}  // extern "C"
