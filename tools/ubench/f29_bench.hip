// Microbenchmark / correctness probe for the unsaturated 9 x 29-bit signed-limb field representation
// (Montgomery R = 2^261): carry-free column accumulation with v_mad_i64_i32 vs the saturated 8 x 32 multiplier.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "../../sylow_amd/csrc/bn254_fp.hpp"
using namespace bn254;
typedef int32_t i32; typedef int64_t i64;
struct F29 { i32 v[9]; };
#define M29 0x1fffffff
__device__ __forceinline__ F29 f29_mul(const F29& a, const F29& b) {
  const i32 p[9] = {0x187cfd47, 0x10460b6, 0x1c72a34f, 0x2d522d0, 0x1585d978, 0x2db40c0, 0xa6e141, 0xe5c2634, 0x30644e};
  i32 m[9]; F29 r; i64 acc = 0;
#pragma unroll
  for (int k = 0; k < 9; ++k) {
#pragma unroll
    for (int i = 0; i <= k; ++i) acc += (i64)a.v[i] * b.v[k - i];
#pragma unroll
    for (int i = 0; i < k; ++i) acc += (i64)m[i] * p[k - i];
    m[k] = (i32)(((u32)acc * 0x4866389u) & M29);
    acc += (i64)m[k] * p[0];
    acc >>= 29;
  }
#pragma unroll
  for (int k = 9; k < 17; ++k) {
#pragma unroll
    for (int i = k - 8; i < 9; ++i) acc += (i64)a.v[i] * b.v[k - i];
#pragma unroll
    for (int i = k - 8; i < 9; ++i) acc += (i64)m[i] * p[k - i];
    r.v[k - 9] = (i32)((u32)acc & M29);
    acc >>= 29;
  }
  r.v[8] = (i32)acc;
  return r;
}
struct F29x2 { F29 c0, c1; };
// fused Fp2 product: (a0 b0 - a1 b1, a0 b1 + a1 b0), 4 products, 2 reductions, inputs normalized
__device__ __forceinline__ F29 f29_dot2(const F29& a, const F29& b, const F29& c, const F29& d) {   // a*b + c*d
  const i32 p[9] = {0x187cfd47, 0x10460b6, 0x1c72a34f, 0x2d522d0, 0x1585d978, 0x2db40c0, 0xa6e141, 0xe5c2634, 0x30644e};
  i32 m[9]; F29 r; i64 acc = 0;
#pragma unroll
  for (int k = 0; k < 9; ++k) {
#pragma unroll
    for (int i = 0; i <= k; ++i) { acc += (i64)a.v[i] * b.v[k - i]; acc += (i64)c.v[i] * d.v[k - i]; }
#pragma unroll
    for (int i = 0; i < k; ++i) acc += (i64)m[i] * p[k - i];
    m[k] = (i32)(((u32)acc * 0x4866389u) & M29);
    acc += (i64)m[k] * p[0];
    acc >>= 29;
  }
#pragma unroll
  for (int k = 9; k < 17; ++k) {
#pragma unroll
    for (int i = k - 8; i < 9; ++i) { acc += (i64)a.v[i] * b.v[k - i]; acc += (i64)c.v[i] * d.v[k - i]; }
#pragma unroll
    for (int i = k - 8; i < 9; ++i) acc += (i64)m[i] * p[k - i];
    r.v[k - 9] = (i32)((u32)acc & M29);
    acc >>= 29;
  }
  r.v[8] = (i32)acc;
  return r;
}
__device__ __forceinline__ F29x2 f29x2_mul(const F29x2& a, const F29x2& b) {
  F29 na1; for (int i = 0; i < 9; ++i) na1.v[i] = -a.c1.v[i];
  return F29x2{f29_dot2(a.c0, b.c0, na1, b.c1), f29_dot2(a.c0, b.c1, a.c1, b.c0)};
}
template <int V>
__global__ void __launch_bounds__(256) k(const u32* in, u32* out, int n, int iters) {
  int i = blockIdx.x * blockDim.x + threadIdx.x; if (i >= n) return;
  if (V == 0) {       // saturated 8x32 (current product code), two independent chains
    Fp x, y, z, w;
    for (int k = 0; k < 8; ++k) { x.v[k] = in[k * n + i]; y.v[k] = in[(k + 8) * n + i]; z.v[k] = x.v[k] ^ 5; w.v[k] = y.v[k] ^ 9; }
    x.v[7] &= 0x0fffffff; y.v[7] &= 0x0fffffff; z.v[7] &= 0x0fffffff; w.v[7] &= 0x0fffffff;
    for (int it = 0; it < iters; ++it) { x = fp_mul_inline(x, y); z = fp_mul_inline(z, w); }
    for (int k = 0; k < 8; ++k) out[k * n + i] = x.v[k] ^ z.v[k];
  } else if (V == 1) { // 9x29 plain mul
    F29 x, y, z, w;
    for (int k = 0; k < 9; ++k) { x.v[k] = in[k * n + i] & M29; y.v[k] = in[(k + 8) * n + i] & M29; z.v[k] = x.v[k] ^ 5; w.v[k] = y.v[k] ^ 9; }
    x.v[8] &= 0xfffff; y.v[8] &= 0xfffff; z.v[8] &= 0xfffff; w.v[8] &= 0xfffff;
    for (int it = 0; it < iters; ++it) { x = f29_mul(x, y); z = f29_mul(z, w); }
    for (int k = 0; k < 9; ++k) out[k * n + i] = x.v[k] ^ z.v[k];
  } else {             // fused Fp2 product (counts as 3 Karatsuba-equivalent muls)
    F29x2 a, b;
    for (int k = 0; k < 9; ++k) { a.c0.v[k] = in[k * n + i] & M29; a.c1.v[k] = (in[k * n + i] >> 3) & M29; b.c0.v[k] = in[(k + 8) * n + i] & M29; b.c1.v[k] = (in[(k + 8) * n + i] >> 2) & M29; }
    a.c0.v[8] &= 0xfffff; a.c1.v[8] &= 0xfffff; b.c0.v[8] &= 0xfffff; b.c1.v[8] &= 0xfffff;
    for (int it = 0; it < iters; ++it) a = f29x2_mul(a, b);
    for (int k = 0; k < 9; ++k) out[k * n + i] = a.c0.v[k] ^ a.c1.v[k];
  }
}
template <int V> void run(const char* name, u32* din, u32* dout, double per_iter) {
  int iters = 200;
  for (int mult : {2, 4}) {
    int nn = 256 * 4 * 64 * mult;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<V><<<nn / 256, 256>>>(din, dout, nn, iters); hipDeviceSynchronize();
    hipEventRecord(e0); k<V><<<nn / 256, 256>>>(din, dout, nn, iters); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-28s waves/SIMD=%d  %.3f ms  %.1f G Fp-mul-equiv/s\n", name, mult, ms, per_iter * iters * nn / ms / 1e6);
  }
}
int main() {
  int n = 256 * 4 * 64 * 4;
  std::vector<u32> h(17 * n); uint64_t s = 88172645463325252ull;
  for (auto& x : h) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; x = (u32)(s >> 16); }
  u32 *din, *dout; hipMalloc(&din, h.size() * 4); hipMalloc(&dout, 9 * n * 4);
  hipMemcpy(din, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  run<0>("8x32 saturated fp_mul", din, dout, 2);
  run<1>("9x29 signed fp_mul", din, dout, 2);
  run<2>("9x29 fused Fp2 mul (=3 muls)", din, dout, 3);
  // correctness probe: (a*b) in 9x29 vs big-int on host for one lane
  int nn = 64; k<1><<<1, 64>>>(din, dout, nn, 1);
  std::vector<u32> o(9 * nn), in(17 * nn); hipMemcpy(o.data(), dout, 9 * nn * 4, hipMemcpyDeviceToHost);
  // host check of lane 0, chain x only is mixed with z by xor -> recompute both on host with __int128-free big arithmetic is long; print operands/result for the python checker
  hipMemcpy(in.data(), din, 17 * 64 * 4, hipMemcpyDeviceToHost);
  printf("CHECK");
  for (int k2 = 0; k2 < 9; ++k2) printf(" %u", h[k2 * nn + 0]);      // note: layout uses n=nn stride in the probe launch
  printf(" |");
  for (int k2 = 0; k2 < 9; ++k2) printf(" %u", h[(k2 + 8) * nn + 0]);
  printf(" |");
  for (int k2 = 0; k2 < 9; ++k2) printf(" %u", o[k2 * nn + 0]);
  printf("\n");
  return 0;
}
