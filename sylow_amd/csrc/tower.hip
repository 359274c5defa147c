// tower.hip -- the extension tower as entry points of its own, one element per lane: Fp2 / Fp6 operators (fields/fp2.rs, fp6.rs) and the
// raw Fp12 selector behind sylow_hip_fp12_hook_batch (ops 0..11: the one-element-per-lane Fp12 layer of bn254_tower.hpp / bn254_f29.hpp,
// which the tests hold against the lane-pair layer and the oracle).  The Fp12 operators themselves run on the lane-pair layer (plkh::fp12_op).
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
  return plkh::fp12_op(plkh::OPW_MUL, a, b, out, n, stream);
}
int32_t sylow_hip_fp12_sqr_batch(const uint64_t* a, uint64_t* out, size_t n, void* stream) {
  ARGCHK(a && out); if (!n) return SYLOW_HIP_OK;
  return plkh::fp12_op(plkh::OPW_SQR, a, nullptr, out, n, stream);
}
int32_t sylow_hip_fp12_inv_batch(const uint64_t* a, uint64_t* out, size_t n, void* stream) {
  ARGCHK(a && out); if (!n) return SYLOW_HIP_OK;
  return plkh::fp12_op(plkh::OPW_S_INV, a, nullptr, out, n, stream);
}
int32_t sylow_hip_fp12_frobenius_batch(const uint64_t* a, int32_t exponent, uint64_t* out, size_t n, void* stream) {
  ARGCHK(a && out && exponent >= 1 && exponent <= 3); if (!n) return SYLOW_HIP_OK;
  return plkh::fp12_op(plkh::OPW_FROB1 + exponent - 1, a, nullptr, out, n, stream);
}
int32_t sylow_hip_fp12_sparse_mul_batch(const uint64_t* f, const uint64_t* ell, uint64_t* out, size_t n, void* stream) {
  ARGCHK(f && ell && out); if (!n) return SYLOW_HIP_OK;
  return plkh::fp12_op(plkh::OPW_SPARSE, f, ell, out, n, stream);
}
// test hook (not in the public header's stable surface): Granger-Scott cyclotomic square
int32_t sylow_hip_fp12_cyclotomic_sqr_batch(const uint64_t* a, uint64_t* out, size_t n, void* stream) {
  ARGCHK(a && out); if (!n) return SYLOW_HIP_OK;
  return plkh::fp12_op(plkh::OPW_CYCSQR, a, nullptr, out, n, stream);
}

// test hook: raw k_fp12_op selector (8: product on the carry-free core, 9: cyclotomic square on it,
// 10 / 11: exp_by_neg_z on the carry-free / saturated core); selectors 16..28 (the lane-pair Fp12 layer) are served by
// sylow_hip_fp12_hook_batch in plk_pairing.hip, which forwards the others here
}  // extern "C"
namespace towerh {
int32_t fp12_hook(int32_t op, const uint64_t* a, const uint64_t* b, uint64_t* out, size_t n, void* stream) {
  k_fp12_op<<<GRID(n)>>>(op, a, b, out, n); LAUNCHED();
}
}  // namespace towerh
