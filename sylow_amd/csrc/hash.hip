// hash.hip -- hash-to-G1 (g1.rs:307-331: XMD-Keccak256 + two SvdW maps + one complete addition), one element per lane, in a unit of
// its own because it is compiled for FOUR wavefronts per SIMD.  The map is chains of dependent cheap instructions (the Jacobi
// symbols' borrow chains, safegcd, the Keccak rounds): at the two wavefronts per SIMD the 256-register kernels run with, the SIMD
// idles between dependent issues; at four (128 registers, 32 B more stack frame) it measured 8.95 -> 8.0 ms per 2^20 (3 / 5 / 6
// wavefronts: 8.40 / 8.26 / 8.34 ms; the multiply-add-bound G1 scalar multiplication LOSES at four: 15.5 -> 16.9 ms, spills).
// amdgpu_waves_per_eu only applies to kernels; the device functions below them (this unit's own copies of svdw_map, fp_inv_safegcd,
// fp_is_square, fp_pow_words, the Keccak expansion) inherit the budget because EVERY kernel of the unit carries the same attribute --
// keep it that way.
#include "host.hpp"

__global__ void __launch_bounds__(BLOCK) __attribute__((amdgpu_waves_per_eu(4, 4)))
k_hash_to_g1(const uint8_t* msgs, const u64* off, DstPrime dp, u64* oxy, uint8_t* oinf, uint8_t* status, size_t n, int negate, u64* proj) {
  size_t i = TID;
  if (i >= n) return;
  G1P h;
  bool ok = hash_to_g1(h, msgs + off[i], (size_t)(off[i + 1] - off[i]), dp);
  if (proj) {                                            // projective [12][n], the summation tree's own layout: no inversion per element
    store_fp(proj, n, i, 0, h.x); store_fp(proj, n, i, 4, h.y); store_fp(proj, n, i, 8, h.z);
    return;
  }
  Fp x, y; bool inf;
  g1_to_affine(x, y, inf, h);
  if (negate && !inf) y = fp_neg(y);                     // -H(m): the G1 side of the e(sig, G2gen) e(-H, pk) == 1 shapes
  store_fp(oxy, n, i, 0, x); store_fp(oxy, n, i, 4, y);
  oinf[i] = inf ? 1 : 0;
  if (status) status[i] = ok ? SYLOW_HIP_ST_OK : SYLOW_HIP_ST_CANNOT_HASH;
}

namespace g1h {
constexpr size_t HASH_WIDE_MAX = 16384;       // 8 messages per wavefront: up to two wavefronts per SIMD
int32_t hash_to_g1_dst(const uint8_t* msgs, const uint64_t* msg_offsets, const DstPrime& dp, uint64_t* out_xy, uint8_t* out_inf, size_t n, int negate, void* stream) {
  if (!n) return SYLOW_HIP_OK;
  // single calls and small batches: eight lanes per message (sign_wide.hip) -- one hash ~1.0 -> 0.36 ms, the head of every single verification
  if (plkh::wide_batch_max() != 0 && n <= HASH_WIDE_MAX) return hash_to_g1_wide(msgs, msg_offsets, dp, out_xy, out_inf, n, negate, stream);
  k_hash_to_g1<<<GRID(n)>>>(msgs, msg_offsets, dp, out_xy, out_inf, nullptr, n, negate, nullptr); LAUNCHED();
}
// H(m_i) projective, straight into a summation tree's scratch array acc [12][n] (library DST): the shape of "sum of the hashes"
int32_t hash_to_g1_proj(const uint8_t* msgs, const uint64_t* msg_offsets, uint64_t* acc, size_t n, void* stream) {
  if (!n) return SYLOW_HIP_OK;
  DstPrime dp; host::dst_arg(dp, nullptr, 0);
  k_hash_to_g1<<<GRID(n)>>>(msgs, msg_offsets, dp, nullptr, nullptr, nullptr, n, 0, acc); LAUNCHED();
}
int32_t hash_to_g1(const uint8_t* msgs, const uint64_t* msg_offsets, uint64_t* out_xy, uint8_t* out_inf, size_t n, int negate, void* stream) {
  DstPrime dp; host::dst_arg(dp, nullptr, 0);
  return hash_to_g1_dst(msgs, msg_offsets, dp, out_xy, out_inf, n, negate, stream);
}
}  // namespace g1h
