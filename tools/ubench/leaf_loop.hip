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

namespace bn254 {
// TWO independent reduce passes as two INTERLEAVED chains: the same instructions as two calls of f29_reduce_terms, limb for limb, but the
// pins alternate between the chains, so that each chain's next multiply-add issues while the other's is in flight.  (Experiment of round 6, kept here and not in the tower: +3.7 % on the linear passes alone, mode 6 against mode 5.  One chain is serial by
// construction -- the running sum is the addend of the next multiply-add -- and the pins are volatile, so two passes written one after the other
// stay one after the other: on their own the linear passes reach 0.72 of ideal issue at two wavefronts per SIMD, tools/ubench/leaf_loop.)
template <int N0, int N1>
BN_DEV void f29_reduce_terms2(F29& r0, F29& r1, const F29* const (&x0)[N0], const i32 (&k0)[N0], const F29* const (&x1)[N1], const i32 (&k1)[N1]) {
  i32 p[9]; f29_p(p);
  i64 t8a = 0, t8b = 0;
#pragma unroll
  for (int j = 0; j < N0; ++j) t8a += (i64)x0[j]->v[8] * k0[j];
#pragma unroll
  for (int j = 0; j < N1; ++j) t8b += (i64)x1[j]->v[8] * k1[j];
  const i32 nqa = -(i32)((t8a * 5547168ll + (1ll << 43)) >> 44), nqb = -(i32)((t8b * 5547168ll + (1ll << 43)) >> 44);
  i64 acca = 0, accb = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    constexpr int NM = N0 > N1 ? N0 : N1;
#pragma unroll
    for (int j = 0; j < NM; ++j) {
      if (j < N0) { acca += (i64)x0[j]->v[i] * k0[j]; BN_CHAIN(acca); }
      if (j < N1) { accb += (i64)x1[j]->v[i] * k1[j]; BN_CHAIN(accb); }
    }
    acca += (i64)nqa * p[i]; BN_CHAIN(acca);
    accb += (i64)nqb * p[i]; BN_CHAIN(accb);
    r0.v[i] = (i32)((u32)acca & BN_M29);
    r1.v[i] = (i32)((u32)accb & BN_M29);
    acca >>= 29;
    accb >>= 29;
  }
  r0.v[8] = (i32)(acca + t8a + (i64)nqa * p[8]);
  r1.v[8] = (i32)(accb + t8b + (i64)nqb * p[8]);
}

}  // namespace bn254
// two products per call, one operand shared (27 argument registers, 18 result registers): half the calls and returns per product
// (a struct of 18 ints comes back through memory: a vector of 18 comes back in registers)
typedef i32 i32x18 __attribute__((ext_vector_type(18)));
struct F29x2 { F29 r0, r1; };
__device__ __noinline__ i32x18 w2_mul_dual_shared_leaf(i32 a0, i32 a1, i32 a2, i32 a3, i32 a4, i32 a5, i32 a6, i32 a7, i32 a8,
                                                      i32 b0, i32 b1, i32 b2, i32 b3, i32 b4, i32 b5, i32 b6, i32 b7, i32 b8,
                                                      i32 c0, i32 c1, i32 c2, i32 c3, i32 c4, i32 c5, i32 c6, i32 c7, i32 c8) {
  const W2 a{F29{{a0, a1, a2, a3, a4, a5, a6, a7, a8}}}, b{F29{{b0, b1, b2, b3, b4, b5, b6, b7, b8}}}, c{F29{{c0, c1, c2, c3, c4, c5, c6, c7, c8}}};
  const F29 x = w2_mul_inl(a, b).c, y = w2_mul_inl(a, c).c;
  return i32x18{x.v[0], x.v[1], x.v[2], x.v[3], x.v[4], x.v[5], x.v[6], x.v[7], x.v[8], y.v[0], y.v[1], y.v[2], y.v[3], y.v[4], y.v[5], y.v[6], y.v[7], y.v[8]};
}
template <int MODE>
__global__ void __launch_bounds__(256, 2) k_leaf_loop(i32* out, int n, int seed) {
  W2 a, b, c;
  for (int i = 0; i < 9; ++i) c.c.v[i] = (i32)((threadIdx.x * 7919u + i * 104729u + seed) & 0x0fffffff);
  for (int i = 0; i < 9; ++i) { a.c.v[i] = (i32)((threadIdx.x * 2654435761u + i * 40503u + seed) & 0x0fffffff); b.c.v[i] = (i32)((threadIdx.x * 40503u + i * 7919u + seed) & 0x0fffffff); }
#pragma unroll 1
  for (int it = 0; it < n; ++it) {
    if (MODE == 0) a = w2_mul(a, b);                       // out-of-line leaf: call + return per product
    else if (MODE == 1) a = w2_mul_inl(a, b);              // the same product inlined: no control transfer but the loop branch
    else if (MODE == 2) { a = w2_mul(a, b); a = w2_reduce(w2_add(a, b)); }  // leaf + one reduce pass (the tower's rhythm)
    else if (MODE == 3) { const W2 x = w2_mul(a, b), y = w2_mul(a, c); a = w2_add(x, y); a = w2_norm(a); c = w2_norm(w2_sub(x, y)); }   // two leaf calls per iteration
    else if (MODE == 5) {                                  // linear layer only: the passes the tower runs between leaf calls (xi-combination, lazy sums, carry normalisation), no product
      const W2 x = w2_xi_lin(a, 1, b, 1), y = w2_lin2(c, 3, a, -2);
      a = w2_norm(w2_add(x, y)); c = w2_norm(w2_sub(y, b));
    }
    else if (MODE == 6) {                                  // mode 5 with its two reduce passes as two interleaved chains (f29_reduce_terms2)
      const F29 ao = xchg9(a.c);
      const F29* const t0[3] = {&a.c, &ao, &b.c};
      const i32 c0[3] = {bn_keep(9), bn_keep_v(lane_odd() ? 1 : -1), bn_keep(1)};
      const F29* const t1[2] = {&c.c, &a.c};
      const i32 c1[2] = {bn_keep(3), bn_keep(-2)};
      W2 x, y;
      f29_reduce_terms2(x.c, y.c, t0, c0, t1, c1);
      a = w2_norm(w2_add(x, y)); c = w2_norm(w2_sub(y, b));
    }
    else {                                                                                                                             // the same two products in ONE call
      const i32x18 d = w2_mul_dual_shared_leaf(W_ARGS(a.c), W_ARGS(b.c), W_ARGS(c.c));
      const W2 x{F29{{d[0], d[1], d[2], d[3], d[4], d[5], d[6], d[7], d[8]}}}, y{F29{{d[9], d[10], d[11], d[12], d[13], d[14], d[15], d[16], d[17]}}};
      a = w2_norm(w2_add(x, y)); c = w2_norm(w2_sub(x, y));
    }
  }
  for (int i = 0; i < 9; ++i) out[(blockIdx.x * blockDim.x + threadIdx.x) * 9 + i] = a.c.v[i] ^ c.c.v[i];
}
int main() {
  hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
  const int blocks = p.multiProcessorCount * 2, n = 20000;
  i32* out; hipMalloc(&out, (size_t)blocks * 256 * 9 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int mode = 0; mode < 7; ++mode) {
    for (int rep = 0; rep < 2; ++rep) {
      hipEventRecord(e0);
      if (mode == 0) k_leaf_loop<0><<<blocks, 256>>>(out, n, rep); else if (mode == 1) k_leaf_loop<1><<<blocks, 256>>>(out, n, rep); else if (mode == 2) k_leaf_loop<2><<<blocks, 256>>>(out, n, rep);
      else if (mode == 3) k_leaf_loop<3><<<blocks, 256>>>(out, n, rep); else if (mode == 4) k_leaf_loop<4><<<blocks, 256>>>(out, n, rep); else if (mode == 5) k_leaf_loop<5><<<blocks, 256>>>(out, n, rep); else k_leaf_loop<6><<<blocks, 256>>>(out, n, rep);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      if (rep) printf("mode %d (%s): %.3f ms for %d iterations = %.1f ns per iteration per wavefront pair\n", mode, mode == 0 ? "leaf call" : mode == 1 ? "leaf inlined" : mode == 2 ? "leaf call + reduce pass" : mode == 3 ? "two leaf calls + norms" : mode == 4 ? "one dual-leaf call + norms" : mode == 5 ? "linear passes only" : "linear passes, reduce passes interleaved", ms, n, ms * 1e6 / n);
    }
  }
  return 0;
}
