// leaf_loop.hip -- the shipped out-of-line product leaf (bn254_pair29.hpp: w2_mul_leaf) called back to back, two wavefronts per SIMD, nothing else:
// how much of a wavefront's time is parked (SQ_WAIT_ANY) when the ONLY events are the call, the return and the leaf's own s_nop / s_waitcnt?
// (round 6: the Miller-loop kernel parks 11.5 % of its wave cycles with nine memory accesses per pairing.)  Run under
//   rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU -- ./leaf_loop
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -I../../sylow_amd/csrc leaf_loop.hip -o leaf_loop
#include <hip/hip_runtime.h>
#include <cstdio>
#include "bn254_pair29.hpp"
using namespace bn254;
using namespace bn254::pl;

template <int MODE>
__global__ void __launch_bounds__(256, 2) k_leaf_loop(i32* out, int n, int seed) {
  W2 a, b;
  for (int i = 0; i < 9; ++i) { a.c.v[i] = (i32)((threadIdx.x * 2654435761u + i * 40503u + seed) & 0x0fffffff); b.c.v[i] = (i32)((threadIdx.x * 40503u + i * 7919u + seed) & 0x0fffffff); }
#pragma unroll 1
  for (int it = 0; it < n; ++it) {
    if (MODE == 0) a = w2_mul(a, b);                       // out-of-line leaf: call + return per product
    else if (MODE == 1) a = w2_mul_inl(a, b);              // the same product inlined: no control transfer but the loop branch
    else { a = w2_mul(a, b); a = w2_reduce(w2_add(a, b)); }  // leaf + one reduce pass (the tower's rhythm)
  }
  for (int i = 0; i < 9; ++i) out[(blockIdx.x * blockDim.x + threadIdx.x) * 9 + i] = a.c.v[i];
}
int main() {
  hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
  const int blocks = p.multiProcessorCount * 2, n = 20000;
  i32* out; hipMalloc(&out, (size_t)blocks * 256 * 9 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int mode = 0; mode < 3; ++mode) {
    for (int rep = 0; rep < 2; ++rep) {
      hipEventRecord(e0);
      if (mode == 0) k_leaf_loop<0><<<blocks, 256>>>(out, n, rep); else if (mode == 1) k_leaf_loop<1><<<blocks, 256>>>(out, n, rep); else k_leaf_loop<2><<<blocks, 256>>>(out, n, rep);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      if (rep) printf("mode %d (%s): %.3f ms for %d iterations = %.1f ns per iteration per wavefront pair\n", mode, mode == 0 ? "leaf call" : mode == 1 ? "leaf inlined" : "leaf call + reduce pass", ms, n, ms * 1e6 / n);
    }
  }
  return 0;
}
