// plk_quad.hip -- pairing() on lane QUADS: the whole lane-pair tower (bn254_pair29.hpp) compiled with BN_QUAD 1, for batches too small to fill
// the chip with one lane pair per element.
//
// One lane pair per element is the most efficient shape (plk_pairing.hip: 32 elements per wavefront) but its latency is one whole pairing on
// a single wavefront: 4.3 ms with one wavefront per SIMD -- the time of EVERY batch between the one-wavefront-per-element cap (6144) and
// 32 768 elements, however few SIMDs it fills (round 5: 16 384 pairings at 3.8 M/s against 9.7 M/s at 2^20).  Here an element takes a quad
// of lanes = two lane pairs holding the same state; every pair of independent product leaves is split between them and the results are
// exchanged by one DPP quad permutation (bn254_pair29.hpp "leaf PAIRS").  Same formulas, same operand classes, the same integer into every
// Montgomery reduction: Gt values are bit for bit those of k_pairing (tests/test_gpu_quad.py compares every row).
#define BN_QUAD 1
#include "plk_common.hpp"
#include "plk_verify_body.hpp"

namespace plk {
// pairing.rs:870-893 -- thread t: role pair_role(t) of sub-pair quad_sub(t) of element quad_index(t); 16 elements per wavefront
// n = the SoA stride of the arrays, m <= n = the elements of this launch (the tail of a larger batch: pointers already advanced to its first element)
// clk: the live clock probe's accumulator (sylow_hip_clock_probe), or NULL
__global__ void HEAVY_BOUNDS k_pairing_quad(const u64* pxy, const uint8_t* pinf, const u64* qxy, const uint8_t* qinf, u64* gout, size_t n, size_t m, u64* clk) {
  const size_t t = TID, i = quad_index(t);
  const int odd = pair_role(t);
  if (i >= m) return;                                            // all four lanes of a quad leave together
  ClockProbe pb;
  probe_begin(pb, clk);
  // As the tail of a larger batch a quad wavefront shares its SIMD with a lane-pair wavefront that started earlier and, being older, wins every
  // arbitration: the tail then crawls until the rounds are done (traced at 32768 + 16384 pairings: 6.4 ms for a tail that takes 3.0 ms alone,
  // 6.7 ms for the batch).  With static priority the SHORT job runs at its own pace (3.2 ms) and the long one fills its gaps: 6.45 ms -- the two
  // kernels are different code, 70 KB of Miller loop each, so a mixed SIMD reaches 0.75 of the issue rate where two wavefronts of one kernel
  // reach 0.87 (equal shares would give 5.4 ms).  Alone on its SIMD the priority changes nothing.
  __builtin_amdgcn_s_setprio(2);
  const bool either_zero = (pinf && pinf[i]) || (qinf && qinf[i]);
  S12 g;
  if (either_zero) {
    g = s12_one();
  } else {
    const Fp px = load_fp(pxy, n, i, 0), py = load_fp(pxy, n, i, 4);
    const S2 qx = load_s2(qxy, n, i, 0, odd), qy = load_s2(qxy, n, i, 8, odd);
    S12 f;
    miller_loop29g<true, true>(f, px, py, qx, qy);               // isomorphic curves, as k_pairing: a factor in Fp*, gone after the next line
    final_exponentiation29(g, f);
  }
  if (quad_sub(t) == 0) store_s12(gout, n, i, odd, g);
  probe_end(pb, clk);
}
// raw Miller loop and final exponentiation on quads (the reference's curves: the raw value bit for bit)
__global__ void HEAVY_BOUNDS k_miller_loop_quad(const u64* pxy, const u64* qxy, u64* fout, size_t n) {
  const size_t t = TID, i = quad_index(t);
  const int odd = pair_role(t);
  if (i >= n) return;
  const Fp px = load_fp(pxy, n, i, 0), py = load_fp(pxy, n, i, 4);
  const S2 qx = load_s2(qxy, n, i, 0, odd), qy = load_s2(qxy, n, i, 8, odd);
  S12 f;
  miller_loop29g<true>(f, px, py, qx, qy);
  if (quad_sub(t) == 0) store_s12(fout, n, i, odd, f);
}
__global__ void HEAVY_BOUNDS k_final_exp_quad(const u64* fin, u64* gout, size_t n) {
  const size_t t = TID, i = quad_index(t);
  const int odd = pair_role(t);
  if (i >= n) return;
  S12 f, g;
  load_s12(f, fin, n, i, odd);
  final_exponentiation29(g, f);
  if (quad_sub(t) == 0) store_s12(gout, n, i, odd, g);
}
// lib.rs:223-236 as e(sig, G2gen) e(-H(m), pk) == 1 (plk_verify_body.hpp) on quads; never staggered (at most one wavefront per SIMD)
template <bool PK_TABLE>
__global__ void HEAVY_BOUNDS k_bls_verify_fused_quad(const u64* pkxy, const uint8_t* pkinf, const i32* pk_table, const u64* hneg, const uint8_t* hneg_inf,
                                                     const u64* sigxy, const uint8_t* siginf, const i32* gen_table, uint8_t* okout, size_t n, size_t m, u64* clk) {
  const Stagger none{0, 0, 0, 0, nullptr, nullptr, nullptr};
  __builtin_amdgcn_s_setprio(2);                                 // see k_pairing_quad
  ClockProbe pb;
  probe_begin(pb, clk);
  bls_verify_fused_body<PK_TABLE>(pkxy, pkinf, pk_table, hneg, hneg_inf, sigxy, siginf, gen_table, okout, n, m, none);
  probe_end(pb, clk);
}
}  // namespace plk

namespace plkh {
// Largest batch that takes a quad per element: up to one wavefront per SIMD of quads (16 per wavefront: 4 x CUs x 16 elements; 16 384 on this
// part).  Above that the lane-pair kernel's single round is as fast.  SYLOW_HIP_OPT_QUAD_MAX moves it (0: never).
size_t quad_batch_max() {
  const long long o = host::option(SYLOW_HIP_OPT_QUAD_MAX);
  if (o >= 0) return (size_t)o;
  const unsigned cus = host::compute_units();
  return (size_t)(cus ? cus : 256) * 4 * 16;
}
// Tail of a batch of k whole rounds (one wavefront per SIMD of lane pairs each: 128 elements per CU) plus a remainder that fits the quad route:
// the remainder's size, or 0 (no split).  Only for one or two whole rounds: a third round fills every CU's LDS with two lane-pair blocks until the
// end, so the tail could only follow the rounds (measured: 100 000 pairings 15.4 ms split against 15.2 ms as one grid) -- and only when the batch
// is not a whole number of rounds.  SYLOW_HIP_OPT_TAIL_SPLIT = 0 switches it off.
size_t tail_split(size_t n) {
  if (host::option(SYLOW_HIP_OPT_TAIL_SPLIT) == 0) return 0;
  const unsigned cus = host::compute_units();
  const size_t round = (size_t)(cus ? cus : 256) * (BLOCK / 2);
  if (n <= round || n >= 3 * round) return 0;
  const size_t r = n % round;
  return (r && r <= quad_batch_max()) ? r : 0;
}
// pointers at the first element of the range, n = the arrays' SoA stride, m = elements
int32_t pairing_quad_range(const uint64_t* p_xy, const uint8_t* p_inf, const uint64_t* q_xy, const uint8_t* q_inf, uint64_t* gt_out, size_t n, size_t m, void* stream) {
  plk::k_pairing_quad<<<GRID(4 * m)>>>(p_xy, p_inf, q_xy, q_inf, gt_out, n, m, host::clock_probe()); LAUNCHED();
}
int32_t miller_loop_quad_batch(const uint64_t* p_xy, const uint64_t* q_xy, uint64_t* f_out, size_t n, void* stream) {
  plk::k_miller_loop_quad<<<GRID(4 * n)>>>(p_xy, q_xy, f_out, n); LAUNCHED();
}
int32_t verify_fused_quad(int pk_is_table, const uint64_t* pk_xy, const uint8_t* pk_inf, const bn254::i32* pk_table, const uint64_t* hneg, const uint8_t* hneg_inf,
                          const uint64_t* sig_xy, const uint8_t* sig_inf, const bn254::i32* gen, uint8_t* ok, size_t n, size_t m, void* stream) {
  if (pk_is_table) plk::k_bls_verify_fused_quad<true><<<GRID(4 * m)>>>(pk_xy, pk_inf, pk_table, hneg, hneg_inf, sig_xy, sig_inf, gen, ok, n, m, host::clock_probe());
  else plk::k_bls_verify_fused_quad<false><<<GRID(4 * m)>>>(pk_xy, pk_inf, pk_table, hneg, hneg_inf, sig_xy, sig_inf, gen, ok, n, m, host::clock_probe());
  LAUNCHED();
}
int32_t final_exp_quad_batch(const uint64_t* f, uint64_t* gt_out, size_t n, void* stream) {
  plk::k_final_exp_quad<<<GRID(4 * n)>>>(f, gt_out, n); LAUNCHED();
}
}  // namespace plkh
