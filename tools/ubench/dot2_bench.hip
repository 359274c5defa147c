// dot2_bench.hip -- the lane-pair Fp2 product leaf (bn254_pair29.hpp: w2_mul_leaf = DPP operand shuffle + f29_dot2) in isolation,
// at the occupancy of the pairing kernels (2 waves per SIMD), in several instruction schedules.  Every variant must produce the
// bit-identical result of the shipped leaf (checksums compared on the host).  Reports ns per leaf call per SIMD and, with the
// static instruction counts from the ISA, cycles per instruction at the clock the run sustained.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -I../../sylow_amd/csrc dot2_bench.hip -o dot2_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#include "bn254_pair29.hpp"
using namespace bn254;
using namespace bn254::pl;

// ---- V1: the shipped column form, but the two column accumulators are kept apart (an empty asm on y cuts the add tree that
// LLVM's reassociation otherwise re-linearises into ONE dependent v_mad chain per column)
BN_DEV F29 dot2_v1(const F29& a, const F29& b, const F29& c, const F29& d) {
  i32 p[9]; f29_p(p);
  i32 m[9];
  F29 r;
  i64 acc = 0;
#pragma unroll
  for (int k = 0; k < 17; ++k) {
    i64 x = acc, y = 0;
    const int lo = k > 8 ? k - 8 : 0, hi = k < 8 ? k : 8;
#pragma unroll
    for (int i = lo; i <= hi; ++i) { x += (i64)a.v[i] * b.v[k - i]; y += (i64)c.v[i] * d.v[k - i]; }
#pragma unroll
    for (int i = lo; i <= hi; ++i) {
      if (k < 9 && i == k) continue;
      if ((i - lo) & 1) x += (i64)m[i] * p[k - i]; else y += (i64)m[i] * p[k - i];
    }
    asm("" : "+v"(y));
    acc = x + y;
    if (k < 9) {
      m[k] = (i32)(((u32)acc * BN_PINV29) & BN_M29);
      acc += (i64)m[k] * p[0];
    } else {
      r.v[k - 9] = (i32)((u32)acc & BN_M29);
    }
    acc >>= 29;
  }
  r.v[8] = (i32)acc;
  return r;
}
// ---- V2: product scanning first (17 independent column sums), then the Montgomery reduction as 9 steps of 9 independent
// multiply-adds; the only serial chain left is col_k -> m_k -> carry -> col_{k+1}
BN_DEV F29 dot2_v2(const F29& a, const F29& b, const F29& c, const F29& d) {
  i32 p[9]; f29_p(p);
  i64 col[17];
#pragma unroll
  for (int k = 0; k < 17; ++k) {
    const int lo = k > 8 ? k - 8 : 0, hi = k < 8 ? k : 8;
    i64 s = 0;
#pragma unroll
    for (int i = lo; i <= hi; ++i) { s += (i64)a.v[i] * b.v[k - i]; s += (i64)c.v[i] * d.v[k - i]; }
    asm("" : "+v"(s));
    col[k] = s;
  }
  F29 r;
  i64 carry = 0;
#pragma unroll
  for (int k = 0; k < 9; ++k) {
    const i64 t = col[k] + carry;
    const i32 mk = (i32)(((u32)t * BN_PINV29) & BN_M29);
    carry = (t + (i64)mk * p[0]) >> 29;
#pragma unroll
    for (int j = 1; j < 9; ++j) col[k + j] += (i64)mk * p[j];
  }
#pragma unroll
  for (int k = 9; k < 17; ++k) {
    const i64 t = col[k] + carry;
    r.v[k - 9] = (i32)((u32)t & BN_M29);
    carry = t >> 29;
  }
  r.v[8] = (i32)carry;
  return r;
}
// ---- V3: like V1 with THREE accumulators per column (a*b, c*d, m*p kept apart)
BN_DEV F29 dot2_v3(const F29& a, const F29& b, const F29& c, const F29& d) {
  i32 p[9]; f29_p(p);
  i32 m[9];
  F29 r;
  i64 acc = 0;
#pragma unroll
  for (int k = 0; k < 17; ++k) {
    i64 x = acc, y = 0, z = 0;
    const int lo = k > 8 ? k - 8 : 0, hi = k < 8 ? k : 8;
#pragma unroll
    for (int i = lo; i <= hi; ++i) { x += (i64)a.v[i] * b.v[k - i]; y += (i64)c.v[i] * d.v[k - i]; }
#pragma unroll
    for (int i = lo; i <= hi; ++i) {
      if (k < 9 && i == k) continue;
      z += (i64)m[i] * p[k - i];
    }
    asm("" : "+v"(y));
    asm("" : "+v"(z));
    acc = x + y + z;
    if (k < 9) {
      m[k] = (i32)(((u32)acc * BN_PINV29) & BN_M29);
      acc += (i64)m[k] * p[0];
    } else {
      r.v[k - 9] = (i32)((u32)acc & BN_M29);
    }
    acc >>= 29;
  }
  r.v[8] = (i32)acc;
  return r;
}

// ---- V4: ONE accumulator carried through all 17 columns (no per-column merge add): every multiply-add is written as inline
// asm so that the compiler cannot restart each column from zero; 17 x v_lshl_add_u64 fewer, but one long dependent chain
#define MADI(acc, x, y) asm("v_mad_i64_i32 %0, vcc, %1, %2, %0" : "+v"(acc) : "v"(x), "v"(y) : "vcc")
#define MADU(acc, x, y) asm("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc) : "v"(x), "s"(y) : "vcc")
BN_DEV F29 dot2_v4(const F29& a, const F29& b, const F29& c, const F29& d) {
  i32 p[9]; f29_p(p);
  i32 m[9];
  F29 r;
  i64 acc = 0;
#pragma unroll
  for (int k = 0; k < 17; ++k) {
    const int lo = k > 8 ? k - 8 : 0, hi = k < 8 ? k : 8;
#pragma unroll
    for (int i = lo; i <= hi; ++i) { MADI(acc, a.v[i], b.v[k - i]); MADI(acc, c.v[i], d.v[k - i]); }
#pragma unroll
    for (int i = lo; i <= hi; ++i) {
      if (k < 9 && i == k) continue;
      MADU(acc, m[i], p[k - i]);
    }
    if (k < 9) {
      m[k] = (i32)(((u32)acc * BN_PINV29) & BN_M29);
      MADU(acc, m[k], p[0]);
    } else {
      r.v[k - 9] = (i32)((u32)acc & BN_M29);
    }
    acc >>= 29;
  }
  r.v[8] = (i32)acc;
  return r;
}
// ---- V5: two chains in asm (even / odd products), merged once per column: the shipped structure without relying on the compiler
BN_DEV F29 dot2_v5(const F29& a, const F29& b, const F29& c, const F29& d) {
  i32 p[9]; f29_p(p);
  i32 m[9];
  F29 r;
  i64 acc = 0;
#pragma unroll
  for (int k = 0; k < 17; ++k) {
    const int lo = k > 8 ? k - 8 : 0, hi = k < 8 ? k : 8;
    i64 y = 0;
#pragma unroll
    for (int i = lo; i <= hi; ++i) { MADI(acc, a.v[i], b.v[k - i]); MADI(y, c.v[i], d.v[k - i]); }
#pragma unroll
    for (int i = lo; i <= hi; ++i) {
      if (k < 9 && i == k) continue;
      if ((i - lo) & 1) MADU(acc, m[i], p[k - i]); else MADU(y, m[i], p[k - i]);
    }
    acc += y;
    if (k < 9) {
      m[k] = (i32)(((u32)acc * BN_PINV29) & BN_M29);
      MADU(acc, m[k], p[0]);
    } else {
      r.v[k - 9] = (i32)((u32)acc & BN_M29);
    }
    acc >>= 29;
  }
  r.v[8] = (i32)acc;
  return r;
}

// ---- V6: V4 with the multiply-add's carry-out sent to a scratch SGPR pair the compiler allocates (no vcc clobber: the compiler
// pads every vcc-clobbering asm statement with s_nop)
#define MADI6(acc, x, y) asm("v_mad_i64_i32 %0, %1, %2, %3, %0" : "+v"(acc), "=s"(sink) : "v"(x), "v"(y))
#define MADU6(acc, x, y) asm("v_mad_u64_u32 %0, %1, %2, %3, %0" : "+v"(acc), "=s"(sink) : "v"(x), "s"(y))
BN_DEV F29 dot2_v6(const F29& a, const F29& b, const F29& c, const F29& d) {
  i32 p[9]; f29_p(p);
  i32 m[9];
  F29 r;
  i64 acc = 0;
  u64 sink;
#pragma unroll
  for (int k = 0; k < 17; ++k) {
    const int lo = k > 8 ? k - 8 : 0, hi = k < 8 ? k : 8;
#pragma unroll
    for (int i = lo; i <= hi; ++i) { MADI6(acc, a.v[i], b.v[k - i]); MADI6(acc, c.v[i], d.v[k - i]); }
#pragma unroll
    for (int i = lo; i <= hi; ++i) {
      if (k < 9 && i == k) continue;
      MADU6(acc, m[i], p[k - i]);
    }
    if (k < 9) {
      m[k] = (i32)(((u32)acc * BN_PINV29) & BN_M29);
      MADU6(acc, m[k], p[0]);
    } else {
      r.v[k - 9] = (i32)((u32)acc & BN_M29);
    }
    acc >>= 29;
  }
  r.v[8] = (i32)acc;
  return r;
}

// ---- V7: one accumulator carried through all columns, plain C multiply-adds, an EMPTY input-only volatile asm after each (the
// value must exist at that point; an in/out operand instead makes the compiler pad every fence with s_nop) so that the compiler
// can neither re-associate the sum nor restart a column from zero (no instruction is emitted for a fence)
#define FENCE(x) asm volatile("" :: "v"(x))
BN_DEV F29 dot2_v7(const F29& a, const F29& b, const F29& c, const F29& d) {
  i32 p[9]; f29_p(p);
  i32 m[9];
  F29 r;
  i64 acc = 0;
#pragma unroll
  for (int k = 0; k < 17; ++k) {
    const int lo = k > 8 ? k - 8 : 0, hi = k < 8 ? k : 8;
#pragma unroll
    for (int i = lo; i <= hi; ++i) { acc += (i64)a.v[i] * b.v[k - i]; FENCE(acc); acc += (i64)c.v[i] * d.v[k - i]; FENCE(acc); }
#pragma unroll
    for (int i = lo; i <= hi; ++i) {
      if (k < 9 && i == k) continue;
      acc += (i64)m[i] * p[k - i]; FENCE(acc);
    }
    if (k < 9) {
      m[k] = (i32)(((u32)acc * BN_PINV29) & BN_M29);
      acc += (i64)m[k] * p[0]; FENCE(acc);
    } else {
      r.v[k - 9] = (i32)((u32)acc & BN_M29);
    }
    acc >>= 29;
  }
  r.v[8] = (i32)acc;
  return r;
}

template <int V>
BN_NOINLINE F29 leaf(i32 a0, i32 a1, i32 a2, i32 a3, i32 a4, i32 a5, i32 a6, i32 a7, i32 a8,
                     i32 b0, i32 b1, i32 b2, i32 b3, i32 b4, i32 b5, i32 b6, i32 b7, i32 b8) {
  const F29 a{{a0, a1, a2, a3, a4, a5, a6, a7, a8}}, b{{b0, b1, b2, b3, b4, b5, b6, b7, b8}};
  const i32 m = lane_odd() ? 0 : -1;
  const F29 B0 = dpp_pick9(b, false), B1 = dpp_pick9(b, true);
  F29 x2 = dpp_xor9(a, m);
#pragma unroll
  for (int i = 0; i < 9; ++i) x2.v[i] -= m;
  if (V == 0) return f29_dot2(a, B0, x2, B1);
  if (V == 1) return dot2_v1(a, B0, x2, B1);
  if (V == 2) return dot2_v2(a, B0, x2, B1);
  if (V == 4) return dot2_v4(a, B0, x2, B1);
  if (V == 5) return dot2_v5(a, B0, x2, B1);
  if (V == 6) return dot2_v6(a, B0, x2, B1);
  if (V == 7) return dot2_v7(a, B0, x2, B1);
  return dot2_v3(a, B0, x2, B1);
}

// DEP = 1: every call depends on the previous one (x = x * y);  DEP = 0: two independent products alternate (x = x*y; z = z*w)
template <int V, int DEP>
__global__ void __launch_bounds__(256, 2) k_bench(const u32* in, u32* out, int iters) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x, n = gridDim.x * blockDim.x;
  F29 x, y, z, w;
#pragma unroll
  for (int k = 0; k < 9; ++k) {
    x.v[k] = (i32)(in[k * n + i] & BN_M29); y.v[k] = (i32)(in[(k + 9) * n + i] & BN_M29);
    z.v[k] = x.v[k] ^ 0x155; w.v[k] = y.v[k] ^ 0x2aa;
  }
  x.v[8] &= 0xfffff; y.v[8] &= 0xfffff; z.v[8] &= 0xfffff; w.v[8] &= 0xfffff;
#pragma unroll 1
  for (int it = 0; it < iters; ++it) {
    x = leaf<V>(W_ARGS(x), W_ARGS(y));
    if (DEP) x = leaf<V>(W_ARGS(x), W_ARGS(w)); else z = leaf<V>(W_ARGS(z), W_ARGS(w));
  }
#pragma unroll
  for (int k = 0; k < 9; ++k) out[k * n + i] = (u32)(x.v[k] ^ z.v[k]);
}

template <int V, int DEP>
double run(const char* name, const u32* din, u32* dout, int n, std::vector<u32>& res) {
  const int iters = 2000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k_bench<V, DEP><<<n / 256, 256>>>(din, dout, 10); hipDeviceSynchronize();
  hipEventRecord(e0); k_bench<V, DEP><<<n / 256, 256>>>(din, dout, iters); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  res.resize(9 * (size_t)n); hipMemcpy(res.data(), dout, res.size() * 4, hipMemcpyDeviceToHost);
  const double calls_per_simd = 2.0 * iters * (n / 64) / 1024.0;        // wave-level leaf calls per SIMD
  const double ns = ms * 1e6 / calls_per_simd;
  printf("%-34s dep=%d  %8.3f ms  %7.1f ns per leaf call per SIMD\n", name, DEP, ms, ns);
  return ns;
}

int main() {
  const int n = 256 * 2 * 256;          // 256 CUs x 2 blocks x 256 threads = 2 waves per SIMD
  std::vector<u32> h(18 * (size_t)n); uint64_t s = 88172645463325252ull;
  for (auto& x : h) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; x = (u32)(s >> 16); }
  u32 *din, *dout; hipMalloc(&din, h.size() * 4); hipMalloc(&dout, 9 * (size_t)n * 4);
  hipMemcpy(din, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  std::vector<u32> r0, r;
  // correctness of every variant against the shipped form
  run<0, 1>("V0 shipped column form", din, dout, n, r0);
  run<1, 1>("V1 two accumulators kept apart", din, dout, n, r); printf("   equal=%d\n", (int)(r == r0));
  run<2, 1>("V2 products first, then reduction", din, dout, n, r); printf("   equal=%d\n", (int)(r == r0));
  run<3, 1>("V3 three accumulators", din, dout, n, r); printf("   equal=%d\n", (int)(r == r0));
  run<4, 1>("V4 one chained accumulator (asm)", din, dout, n, r); printf("   equal=%d\n", (int)(r == r0));
  run<7, 1>("V7 one chained accumulator, fences", din, dout, n, r); printf("   equal=%d\n", (int)(r == r0));
  // A/B: alternate the variants 12 times (the clock drifts with load: compare minima and medians, not single runs)
  const int R = 12;
  std::vector<double> t[2][4];
  for (int rep = 0; rep < R; ++rep) {
    t[1][0].push_back(run<0, 1>("V0", din, dout, n, r)); t[1][1].push_back(run<2, 1>("V2", din, dout, n, r));
    t[1][2].push_back(run<4, 1>("V4", din, dout, n, r)); t[1][3].push_back(run<7, 1>("V7", din, dout, n, r));
    t[0][0].push_back(run<0, 0>("V0", din, dout, n, r)); t[0][1].push_back(run<2, 0>("V2", din, dout, n, r));
    t[0][2].push_back(run<4, 0>("V4", din, dout, n, r)); t[0][3].push_back(run<7, 0>("V7", din, dout, n, r));
  }
  const char* nm[4] = {"V0 shipped", "V2 products first", "V4 chained (asm)", "V7 chained (fences)"};
  for (int dep = 1; dep >= 0; --dep) for (int v = 0; v < 4; ++v) {
    std::sort(t[dep][v].begin(), t[dep][v].end());
    printf("SUMMARY dep=%d %-22s min %7.1f  median %7.1f ns per leaf call per SIMD\n", dep, nm[v], t[dep][v][0], t[dep][v][R / 2]);
  }
  return 0;
}
