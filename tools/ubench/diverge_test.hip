// does a per-lane trip count around the out-of-line tower routines behave? (debugging aid)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "../../sylow_amd/csrc/bn254_pairing.hpp"
using namespace bn254;
template <int MODE>
__global__ void k(const u32* in, u32* out, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x; if (i >= n) return;
  Fp a; for (int k = 0; k < 8; ++k) a.v[k] = in[k];
  int t = 1 + (i & 1);
  if (MODE == 0) {           // divergent loop around noinline fp_mul
#pragma unroll 1
    for (int j = 0; j < t; ++j) a = fp_mul(a, a);
  } else if (MODE == 1) {    // same with inline body
#pragma unroll 1
    for (int j = 0; j < t; ++j) a = fp_mul_inline(a, a);
  } else if (MODE == 2) {    // divergent loop around fp6_mul (reference args)
    Fp6 x = Fp6{Fp2{a, a}, Fp2{a, a}, Fp2{a, a}};
#pragma unroll 1
    for (int j = 0; j < t; ++j) fp6_mul(x, x, x);
    a = x.c1.c0;
  } else {                   // uniform count, reference
    Fp6 x = Fp6{Fp2{a, a}, Fp2{a, a}, Fp2{a, a}};
    fp6_mul(x, x, x);
    Fp6 y; fp6_mul(y, x, x);
    a = (i & 1) ? y.c1.c0 : x.c1.c0;
  }
  for (int k = 0; k < 8; ++k) out[i * 8 + k] = a.v[k];
}
int main() {
  u32 h[8] = {0x12345678, 0x9abcdef0, 0x0fedcba9, 0x87654321, 0x11111111, 0x22222222, 0x33333333, 0x01234567};
  u32 *din, *dout; hipMalloc(&din, 32); hipMalloc(&dout, 64 * 32 * 4);
  hipMemcpy(din, h, 32, hipMemcpyHostToDevice);
  std::vector<u32> r[4];
  for (int m = 0; m < 4; ++m) {
    if (m == 0) k<0><<<1, 64>>>(din, dout, 64); if (m == 1) k<1><<<1, 64>>>(din, dout, 64);
    if (m == 2) k<2><<<1, 64>>>(din, dout, 64); if (m == 3) k<3><<<1, 64>>>(din, dout, 64);
    r[m].resize(64 * 8); hipMemcpy(r[m].data(), dout, 64 * 32, hipMemcpyDeviceToHost);
  }
  auto cmp = [&](int a, int b, const char* n) { int bad = 0; for (int i = 0; i < 64; ++i) for (int k = 0; k < 8; ++k) bad += r[a][i * 8 + k] != r[b][i * 8 + k]; printf("%s mismatching words: %d\n", n, bad); };
  cmp(0, 1, "noinline fp_mul divergent vs inline divergent");
  cmp(2, 3, "fp6_mul divergent vs uniform");
  printf("lane0 %08x lane1 %08x (mode0) | lane0 %08x lane1 %08x (mode1)\n", r[0][0], r[0][8], r[1][0], r[1][8]);
  return 0;
}
