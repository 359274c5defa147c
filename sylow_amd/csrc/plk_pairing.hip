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
__global__ void HEAVY_BOUNDS k_pairing(const u64* pxy, const uint8_t* pinf, const u64* qxy, const uint8_t* qinf, u64* gout, size_t n) {
  const size_t t = TID, i = pair_index(t);
  const int odd = pair_role(t);
  if (i >= n) return;
  const bool either_zero = (pinf && pinf[i]) || (qinf && qinf[i]);
  S12 g;
  if (either_zero) {
    g = s12_one();                 // Miller value forced to one; final_exponentiation(1) == 1
  } else {
    const Fp px = load_fp(pxy, n, i, 0), py = load_fp(pxy, n, i, 4);
    const S2 qx = load_s2(qxy, n, i, 0, odd), qy = load_s2(qxy, n, i, 8, odd);
    S12 f;
    miller_loop29g<true, true>(f, px, py, qx, qy);          // on the isomorphic curves: the value differs from the reference's raw Miller value by a factor in Fp*, gone after the next line
    final_exponentiation29(g, f);
  }
  store_s12(gout, n, i, odd, g);
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

extern "C" {
int32_t sylow_hip_miller_loop_batch(const uint64_t* p_xy, const uint64_t* q_xy, uint64_t* f_out, size_t n, void* stream) {
  ARGCHK(p_xy && q_xy && f_out); if (!n) return SYLOW_HIP_OK;
  plk::k_miller_loop<<<GRID(2 * n)>>>(p_xy, q_xy, f_out, n); LAUNCHED();
}
int32_t sylow_hip_final_exp_batch(const uint64_t* f, uint64_t* gt_out, size_t n, void* stream) {
  ARGCHK(f && gt_out); if (!n) return SYLOW_HIP_OK;
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
  plk::k_pairing<<<GRID(2 * n)>>>(p_xy, p_inf, q_xy, q_inf, gt_out, n); LAUNCHED();
}
// test hook: raw Fp12 selector.  0..11: the one-element-per-lane layer (tower.hip: 8 product on the carry-free core, 9 cyclotomic square on
// it, 10 / 11 exp_by_neg_z on the carry-free / saturated core); 16..29: the lane-pair Fp12 layer (plk::k_w12_op)
int32_t sylow_hip_fp12_hook_batch(int32_t op, const uint64_t* a, const uint64_t* b, uint64_t* out, size_t n, void* stream) {
  ARGCHK(a && out && op >= 0 && (op <= 11 || (op >= 16 && op <= plk::OPW_LAST))); if (!n) return SYLOW_HIP_OK;
  if (op < 16) return towerh::fp12_hook(op, a, b, out, n, stream);
  plk::k_w12_op<<<GRID(2 * n)>>>(op, a, b, out, n); LAUNCHED();
}
}  // extern "C"
