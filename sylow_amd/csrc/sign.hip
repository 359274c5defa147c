// sign.hip -- batched BLS signing (lib.rs:179-187: sk * H(m)), one element per lane, in a unit of its own because it is compiled for THREE
// wavefronts per SIMD: the kernel is hash-to-G1 (chains of dependent cheap instructions, fastest at four wavefronts: hash.hip) followed by a
// G1 scalar multiplication (multiply-add bound, indifferent between two and three, slower at four).  At three (168 registers) it measured
// 22.05 -> 20.85 ms per 2^20 on one box against the two-wavefront build; the scalar multiplication alone 14.7 -> 14.6 ms.
// amdgpu_waves_per_eu only applies to kernels; the device functions below inherit the budget because the unit's only kernel carries it.
#include "host.hpp"

// lib.rs:179-187
__global__ void __launch_bounds__(BLOCK) __attribute__((amdgpu_waves_per_eu(3, 3)))
k_bls_sign(const u64* sk, const uint8_t* msgs, const u64* off, DstPrime dp, u64* oxy, uint8_t* oinf, size_t n, uint8_t* tables) {
  size_t i = TID;
  if (i >= n) return;
  G1P h;
  hash_to_g1(h, msgs + off[i], (size_t)(off[i + 1] - off[i]), dp);
  u32 k[8];
  load_scalar(k, sk, n, i);
  G1P s = tables ? g1_scalar_mul_ws(h, k, tables + i * G1_TABLE_BYTES_PER_LANE) : g1_scalar_mul(h, k);
  Fp x, y; bool inf;
  g1_to_affine(x, y, inf, s);
  store_fp(oxy, n, i, 0, x); store_fp(oxy, n, i, 4, y);
  oinf[i] = inf ? 1 : 0;
}


constexpr size_t SIGN_WIDE_MAX = 16384;      // 8 signatures per wavefront: up to two wavefronts per SIMD (16 384: 1.36 against 1.74 ms; 24 576: 1.98 against 1.76)
namespace g1h {
size_t sign_wide_max() { return (size_t)host::option_or(SYLOW_HIP_OPT_SIGN_WIDE_MAX, (long long)SIGN_WIDE_MAX); }     // the option: crossover measurements (tools/dbg/time_sign.py)
}

extern "C" {
int32_t sylow_hip_bls_sign_batch(const uint64_t* sk, const uint8_t* msgs, const uint64_t* msg_offsets,
                                 uint64_t* sig_xy, uint8_t* sig_inf, size_t n, void* stream) {
  ARGCHK(sk && msgs && msg_offsets && sig_xy && sig_inf); if (!n) return SYLOW_HIP_OK;
  // single calls and small batches: eight lanes per signature (sign_wide.hip) -- one signature 2.1 -> 0.67 ms; the one-lane kernel below wins
  // once the batch fills the chip's lanes (SYLOW_HIP_OPT_WIDE_TAIL = 0 switches every one-wavefront-per-element route off, this one included)
  if (plkh::wide_batch_max() != 0 && n <= g1h::sign_wide_max()) return g1h::sign_wide(sk, msgs, msg_offsets, sig_xy, sig_inf, n, stream);
  DstPrime dp; host::dst_arg(dp, nullptr, 0);
  host::Lease ws;
  // window tables in a leased global block, one contiguous KB per lane (bn254_pairing.hpp: G1TableGlobal); a failed lease keeps them in the
  // stack frame
  uint8_t* tables = nullptr;
  if (ws.acquire(n * G1_TABLE_BYTES_PER_LANE, (hipStream_t)stream) == SYLOW_HIP_OK) tables = (uint8_t*)ws.p;
  else (void)hipGetLastError();
  k_bls_sign<<<GRID(n)>>>(sk, msgs, msg_offsets, dp, sig_xy, sig_inf, n, tables);
  const hipError_t e = hipGetLastError();
  const int32_t rc = ws.release();
  return e != hipSuccess ? host::fail(e, "kernel launch") : rc;
}
}  // extern "C"
