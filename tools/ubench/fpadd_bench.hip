// fpadd_bench.hip -- the HBM-bound Fp micro-batch (2 x 32 B read + 32 B written per element, struct-of-arrays limb planes) in several
// memory-access shapes: elements per lane (2 = one 16-byte access per plane, 4 = two), block size, non-temporal accesses.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -I../../sylow_amd/csrc fpadd_bench.hip -o fpadd_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#include "bn254_fr.hpp"
using namespace bn254;

template <int E, bool NT>
__device__ __forceinline__ void ld(Fp (&e)[E], const u64* __restrict__ base, size_t n, size_t i) {
#pragma unroll
  for (int k = 0; k < 4; ++k) {
#pragma unroll
    for (int h = 0; h < E / 2; ++h) {
      const ulonglong2* p = reinterpret_cast<const ulonglong2*>(base + (size_t)k * n + i + 2 * h);
      ulonglong2 w;
      if (NT) { w.x = __builtin_nontemporal_load(&p->x); w.y = __builtin_nontemporal_load(&p->y); } else w = *p;
      e[2 * h].v[2 * k] = (u32)w.x; e[2 * h].v[2 * k + 1] = (u32)(w.x >> 32);
      e[2 * h + 1].v[2 * k] = (u32)w.y; e[2 * h + 1].v[2 * k + 1] = (u32)(w.y >> 32);
    }
  }
}
template <int E, bool NT>
__device__ __forceinline__ void st(u64* __restrict__ base, size_t n, size_t i, const Fp (&e)[E]) {
#pragma unroll
  for (int k = 0; k < 4; ++k) {
#pragma unroll
    for (int h = 0; h < E / 2; ++h) {
      ulonglong2 w;
      w.x = (u64)e[2 * h].v[2 * k] | ((u64)e[2 * h].v[2 * k + 1] << 32);
      w.y = (u64)e[2 * h + 1].v[2 * k] | ((u64)e[2 * h + 1].v[2 * k + 1] << 32);
      ulonglong2* p = reinterpret_cast<ulonglong2*>(base + (size_t)k * n + i + 2 * h);
      if (NT) { __builtin_nontemporal_store(w.x, &p->x); __builtin_nontemporal_store(w.y, &p->y); } else *p = w;
    }
  }
}
template <int E, int BLOCK, bool NT>
__global__ void __launch_bounds__(BLOCK) k_add(const u64* __restrict__ a, const u64* __restrict__ b, u64* __restrict__ out, size_t n) {
  const size_t i = ((size_t)blockIdx.x * BLOCK + threadIdx.x) * E;
  if (i >= n) return;
  Fp x[E], y[E], r[E];
  ld<E, NT>(x, a, n, i);
  ld<E, NT>(y, b, n, i);
#pragma unroll
  for (int j = 0; j < E; ++j) r[j] = fp_add(fp_reduce_plain(x[j]), fp_reduce_plain(y[j]));
  st<E, NT>(out, n, i, r);
}
// float4 copy of the same byte volume (2 reads + 1 write per 16 B): the achievable-bandwidth reference on this box
__global__ void __launch_bounds__(256) k_copy(const float4* __restrict__ a, const float4* __restrict__ b, float4* __restrict__ o, size_t n16) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n16) return;
  float4 x = a[i], y = b[i];
  o[i] = make_float4(x.x + y.x, x.y + y.y, x.z + y.z, x.w + y.w);
}
template <class F> double timeit(F f) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  f(); hipDeviceSynchronize();
  std::vector<double> t;
  for (int r = 0; r < 7; ++r) {
    hipEventRecord(e0); for (int k = 0; k < 10; ++k) f(); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); t.push_back(ms / 10);
  }
  std::sort(t.begin(), t.end());
  return t[t.size() / 2];
}
int main() {
  const size_t n = 1 << 24;
  u64 *a, *b, *o; hipMalloc(&a, 32 * n); hipMalloc(&b, 32 * n); hipMalloc(&o, 32 * n);
  hipMemset(a, 1, 32 * n); hipMemset(b, 2, 32 * n);
  auto rep = [&](const char* name, double ms) { printf("%-44s %8.1f us  %6.2f TB/s (%.0f %% of 8 TB/s)\n", name, ms * 1e3, 96.0 * n / (ms * 1e-3) / 1e12, 96.0 * n / (ms * 1e-3) / 8e12 * 100); };
  rep("float4 copy-add (2 reads + 1 write)", timeit([&] { k_copy<<<(2 * n + 255) / 256, 256>>>((float4*)a, (float4*)b, (float4*)o, 2 * n); }));
  rep("E=2 block 256 (shipped shape)", timeit([&] { k_add<2, 256, false><<<(n / 2 + 255) / 256, 256>>>(a, b, o, n); }));
  rep("E=2 block 512", timeit([&] { k_add<2, 512, false><<<(n / 2 + 511) / 512, 512>>>(a, b, o, n); }));
  rep("E=2 block 1024", timeit([&] { k_add<2, 1024, false><<<(n / 2 + 1023) / 1024, 1024>>>(a, b, o, n); }));
  rep("E=2 block 256 non-temporal", timeit([&] { k_add<2, 256, true><<<(n / 2 + 255) / 256, 256>>>(a, b, o, n); }));
  rep("E=4 block 256", timeit([&] { k_add<4, 256, false><<<(n / 4 + 255) / 256, 256>>>(a, b, o, n); }));
  rep("E=4 block 256 non-temporal", timeit([&] { k_add<4, 256, true><<<(n / 4 + 255) / 256, 256>>>(a, b, o, n); }));
  rep("E=4 block 512", timeit([&] { k_add<4, 512, false><<<(n / 4 + 511) / 512, 512>>>(a, b, o, n); }));
  return 0;
}
