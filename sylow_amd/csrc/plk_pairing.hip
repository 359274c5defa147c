// plk_pairing.hip -- pairing(), Miller loop and final exponentiation on lane pairs + the carry-free core: the kernels
// behind BASELINE.json's metric (plk::k_pairing), and the lane-pair Fp12 test hook.
#include "plk_common.hpp"

namespace plk {
// ------------------------------------------------------------------ pairing(), Miller loop, final exponentiation ------
__global__ void HEAVY_BOUNDS k_miller_loop(const u64* pxy, const u64* qxy, u64* fout, size_t n) {
  const size_t t = TID, i = pair_index(t);
  const int odd = pair_role(t);
  if (i >= n) return;
  const Fp px = load_fp(pxy, n, i, 0), py = load_fp(pxy, n, i, 4);
  const S2 qx = load_s2(qxy, n, i, 0, odd), qy = load_s2(qxy, n, i, 8, odd);
  S12 f;
  miller_loop29g<true>(f, px, py, qx, qy);
  store_s12(fout, n, i, odd, f);
}
__global__ void HEAVY_BOUNDS k_final_exp(const u64* fin, u64* gout, size_t n) {
  const size_t t = TID, i = pair_index(t);
  const int odd = pair_role(t);
  if (i >= n) return;
  S12 f, g;
  load_s12(f, fin, n, i, odd);
  final_exponentiation29(g, f);
  store_s12(gout, n, i, odd, g);
}
// pairing.rs:870-893
//
// STAGGERED launch (round 5).  Left alone, the two wavefronts a SIMD holds start together and stay in the SAME phase -- both in the Miller
// loop, then both in the final exponentiation -- for the first rounds of a launch; co-resident wavefronts in DIFFERENT phases run 11 % faster
// (tools/dbg/mix_phases.py: half a round of Miller loops beside half a round of final exponentiations, 3.52 ms against 3.90 ms for the same
// work phase by phase), because they stall on different things at different times.  So launches of two rounds or more are skewed by half a period: of the first
// resident set of blocks (2 per CU; block b and b + 256 share a CU on this part) the second half only runs the Miller loop and parks the value
// in a leased block (role M), the blocks that follow them into those slots therefore start their Miller loops while their SIMD partners are
// in the final exponentiation, and the parked values are finished by extra blocks at the end of the grid (role F).  Same arithmetic, same
// results; 1.6 % of a 2^20 batch takes the detour through 384 bytes of HBM per element.
// n = the SoA stride of every array, m <= n = the elements this launch covers (a batch whose tail runs on the lane-quad route beside it)
BN_DEV void pairing_body(const u64* pxy, const uint8_t* pinf, const u64* qxy, const uint8_t* qinf, u64* gout, size_t n, size_t m, const Stagger& st) {
  unsigned chunk;
  int role = stagger_role(st, chunk);
  const size_t t = (size_t)chunk * blockDim.x + threadIdx.x, i = pair_index(t);
  const int odd = pair_role(t);
  const size_t np = (size_t)st.count * (BLOCK / 2), ip = (size_t)(chunk - st.first) * (BLOCK / 2) + (i & (BLOCK / 2 - 1));   // parked element index
  if (role == 2 && !stagger_wait(st, chunk)) role = 0;           // parked values not visible within the bound: recompute the chunk whole
  if (role == 2) {                                               // chunks of role 1 / 2 are always whole (stagger_setup only skews full chunks)
    S12 f, g;
    load_s12(f, st.park, np, ip, odd);
    final_exponentiation29(g, f);
    store_s12(gout, n, i, odd, g);
    return;
  }
  if (i >= m) return;
  const bool either_zero = (pinf && pinf[i]) || (qinf && qinf[i]);
  S12 g;
  if (either_zero) {
    g = s12_one();                 // Miller value forced to one; final_exponentiation(1) == 1
    if (role == 1) store_s12(st.park, np, ip, odd, g);
  } else {
    const Fp px = load_fp(pxy, n, i, 0), py = load_fp(pxy, n, i, 4);
    const S2 qx = load_s2(qxy, n, i, 0, odd), qy = load_s2(qxy, n, i, 8, odd);
    S12 f;
    miller_loop29g<true, true>(f, px, py, qx, qy);          // on the isomorphic curves: the value differs from the reference's raw Miller value by a factor in Fp*, gone after the next line
    if (role == 1) store_s12(st.park, np, ip, odd, f);
    else final_exponentiation29(g, f);
  }
  if (role == 1) { stagger_publish(st, chunk); return; }
  store_s12(gout, n, i, odd, g);
}
__global__ void HEAVY_BOUNDS k_pairing(const u64* pxy, const uint8_t* pinf, const u64* qxy, const uint8_t* qinf, u64* gout, size_t n, size_t m, Stagger st) {
  ClockProbe pb;
  probe_begin(pb, st.clk);
  pairing_body(pxy, pinf, qxy, qinf, gout, n, m, st);
  probe_end(pb, st.clk);
}

// test hook: the lane-pair Fp12 layer one operation at a time (ops 16.. of sylow_hip_fp12_hook_batch); `b` carries the second
// operand, or the three line coefficients (ell_0, ell_vw, ell_vv) in its first 24 words for the sparse product
using namespace plkh;      // the OPW_* selectors (host.hpp)
__global__ void HEAVY_BOUNDS k_w12_op(int op, const u64* a, const u64* b, u64* out, size_t n) {
  const size_t t = TID, i = pair_index(t);
  const int odd = pair_role(t);
  if (i >= n) return;
  S12 sx, sy, sr;
  load_s12(sx, a, n, i, odd);
  if (b) {
    if (op == OPW_SPARSE || op == OPW_SPARSE_UNIT) {      // three line coefficients: 24 words, whatever the width of the array behind them
      sy.c0.c0 = load_s2(b, n, i, 0, odd); sy.c0.c1 = load_s2(b, n, i, 8, odd); sy.c0.c2 = load_s2(b, n, i, 16, odd);
      sy.c1 = sy.c0;
    } else {
      load_s12(sy, b, n, i, odd);
    }
  }
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
      case OPW_SPARSE_UNIT: r = w12_sparse_mul_unit(x, (i32)(i & 1), y.c0.c1, y.c0.c2); break;
      default: exp_by_neg_z29(r, x); break;
    }
    w12_to_s12(sr, r);
  }
  store_s12(out, n, i, odd, sr);
}
}  // namespace plk

// Fp12 on the lane-pair layer for the public tower entry points (tower.hip forwards them here)
namespace plkh {
int32_t fp12_op(int32_t op, const uint64_t* a, const uint64_t* b, uint64_t* out, size_t n, void* stream) {
  plk::k_w12_op<<<GRID(2 * n)>>>(op, a, b, out, n); LAUNCHED();
}
}  // namespace plkh

namespace plkh {
// SYLOW_HIP_OPT_STAGGER = 0: every block runs its whole element (A/B runs); 2: the skew with the parking blocks' flags muted, so that every
// finishing block times out and takes its recompute fallback (tests/test_gpu_routes.py runs the parity tests under both)
static int stagger_mode() { return (int)host::option_or(SYLOW_HIP_OPT_STAGGER, 1); }
// Fills `sg` for a staggered launch of `nblk` blocks (`full` of them whole chunks) when the batch is at least two rounds of the resident
// blocks (2 per CU): measured on k_pairing, 2^17 elements 14.46 -> 14.00 ms, 2^18 27.86 -> 27.33, 2^19 54.95 -> 54.3, 2^20 108.5 -> 108.0;
// exactly one round (2^16) LOSES 2 % (the parked half runs its final exponentiations beside the other half's), so smaller batches stay plain.
// Leaves sg.count = 0 (plain launch) when the lease fails.  The caller releases `ws` after the launch.
hipError_t stagger_setup(plk::Stagger& sg, host::Lease& ws, size_t nblk, size_t full, hipStream_t st, int resident) {
  sg = plk::Stagger{0, 0, (unsigned)nblk, stagger_mode() == 2 ? 1u : 0u, nullptr, nullptr, host::clock_probe()};
  // the first resident set is `resident` blocks per CU (2 for these kernels: asked of the occupancy API, not assumed); its second half parks
  const unsigned cus = host::compute_units(), half = cus * (unsigned)(resident > 1 ? resident : 2) / 2;
  if (stagger_mode() == 0 || !cus || resident < 2 || full < 4 * (size_t)half || nblk >= 0x7fffffffu) return hipSuccess;
  const size_t park_bytes = (size_t)half * (BLOCK / 2) * 48 * sizeof(u64), flag_bytes = ((size_t)half * sizeof(unsigned) + 255) & ~(size_t)255;
  if (ws.acquire(park_bytes + flag_bytes, st) != SYLOW_HIP_OK) { (void)hipGetLastError(); return hipSuccess; }
  sg.first = half; sg.count = half;
  sg.park = (u64*)ws.p; sg.done = (unsigned*)((uint8_t*)ws.p + park_bytes);
  return hipMemsetAsync(sg.done, 0, flag_bytes, st);
}
}  // namespace plkh

extern "C" {
int32_t sylow_hip_miller_loop_batch(const uint64_t* p_xy, const uint64_t* q_xy, uint64_t* f_out, size_t n, void* stream) {
  ARGCHK(p_xy && q_xy && f_out); if (!n) return SYLOW_HIP_OK;
  // small batches: one wavefront per one or two Miller loops on the reference's curves (same field values step by step: the raw value)
  if (n <= plkh::wide_batch_max()) return plkh::miller_raw_wide_batch(p_xy, q_xy, f_out, n, stream);
  if (n <= plkh::quad_batch_max()) return plkh::miller_loop_quad_batch(p_xy, q_xy, f_out, n, stream);     // mid-size batches: a lane quad per element (plk_quad.hip)
  plk::k_miller_loop<<<GRID(2 * n)>>>(p_xy, q_xy, f_out, n); LAUNCHED();
}
int32_t sylow_hip_final_exp_batch(const uint64_t* f, uint64_t* gt_out, size_t n, void* stream) {
  ARGCHK(f && gt_out); if (!n) return SYLOW_HIP_OK;
  if (n <= plkh::wide_batch_max()) return plkh::final_exp_wide_batch(f, gt_out, n, stream);      // small batches: one wavefront per one or two elements
  if (n <= plkh::quad_batch_max()) return plkh::final_exp_quad_batch(f, gt_out, n, stream);
  plk::k_final_exp<<<GRID(2 * n)>>>(f, gt_out, n); LAUNCHED();
}
int32_t sylow_hip_pairing_batch(const uint64_t* p_xy, const uint8_t* p_inf, const uint64_t* q_xy, const uint8_t* q_inf, uint64_t* gt_out, size_t n, void* stream) {
  ARGCHK(p_xy && q_xy && gt_out); if (!n) return SYLOW_HIP_OK;
  // a few pairings are pure latency on one lane pair each: a wavefront per pairing instead; same Gt, an identity on either side gives the
  // identity of Gt either way (pairing.rs:876-886)
  if (n <= plkh::wide_batch_max()) {      // small batches: one wavefront per pairing (2.3 ms against 5.4 ms on one lane pair each)
    host::Lease ws;
    int32_t rc = ws.acquire(48 * n * sizeof(u64), (hipStream_t)stream);
    if (rc != SYLOW_HIP_OK) return rc;
    rc = plkh::pairing_wide_batch(p_xy, p_inf, q_xy, q_inf, (u64*)ws.p, gt_out, n, stream);
    const int32_t r2 = ws.release();
    return rc != SYLOW_HIP_OK ? rc : r2;
  }
  // mid-size batches (up to one wavefront per SIMD of quads): a lane QUAD per element, 1.5 x the speed of a lone lane pair (plk_quad.hip)
  if (n <= plkh::quad_batch_max()) return plkh::pairing_quad_range(p_xy, p_inf, q_xy, q_inf, gt_out, n, n, stream);
  // A batch of k whole rounds of one wavefront per SIMD plus a short tail would leave the tail's few blocks as second wavefronts of their SIMDs
  // for a whole extra pairing time (32 768 pairings 4.3 ms, 33 000: 7.3 ms): the tail takes the quad route on a side stream BESIDE the rounds
  hipStream_t st = (hipStream_t)stream;
  const size_t tail = plkh::tail_split(n), m = n - tail;
  host::Fork fk;
  hipStream_t side = tail ? fk.open(st) : st;
  // staggered launch (see k_pairing): needs one full resident set of blocks (2 per CU) made of whole chunks
  const size_t nblk = (2 * m + BLOCK - 1) / BLOCK, full = (2 * m) / BLOCK;
  plk::Stagger sg;
  host::Lease ws;
  HIPCHK(plkh::stagger_setup(sg, ws, nblk, full, st, plkh::blocks_per_cu(plk::k_pairing)));
  plk::k_pairing<<<dim3((unsigned)(nblk + sg.count)), dim3(BLOCK), 0, st>>>(p_xy, p_inf, q_xy, q_inf, gt_out, n, m, sg);
  const hipError_t e = hipGetLastError();
  int32_t rc = ws.release();
  if (tail && e == hipSuccess && rc == SYLOW_HIP_OK) {
    rc = plkh::pairing_quad_range(p_xy + m, p_inf ? p_inf + m : nullptr, q_xy + m, q_inf ? q_inf + m : nullptr, gt_out + m, n, tail, side);
    const int32_t rj = fk.join(st);
    if (rc == SYLOW_HIP_OK) rc = rj;
  }
  return e != hipSuccess ? host::fail(e, "kernel launch") : rc;
}
// test hook: raw Fp12 selector.  0..11: the one-element-per-lane layer (tower.hip: 8 product on the carry-free core, 9 cyclotomic square on
// it, 10 / 11 exp_by_neg_z on the carry-free / saturated core); 16..29: the lane-pair Fp12 layer (plk::k_w12_op)
int32_t sylow_hip_fp12_hook_batch(int32_t op, const uint64_t* a, const uint64_t* b, uint64_t* out, size_t n, void* stream) {
  ARGCHK(a && out && op >= 0 && (op <= 11 || (op >= 16 && op <= plk::OPW_LAST))); if (!n) return SYLOW_HIP_OK;
  if (op < 16) return towerh::fp12_hook(op, a, b, out, n, stream);
  plk::k_w12_op<<<GRID(2 * n)>>>(op, a, b, out, n); LAUNCHED();
}
}  // extern "C"
