// plk_common.hpp -- helpers shared by the LANE-PAIR kernel units (bn254_pair.hpp, bn254_pair29.hpp).
// Thread t handles coordinate pair_role(t) of element pair_index(t) (bn254_pair.hpp), so a launch covers 2 n threads and a wavefront carries 32 elements.
// Both lanes of a pair always take the same branches.
#pragma once
#include "host.hpp"
#include "bn254_pair29.hpp"

namespace plk {
using namespace bn254;
using namespace bn254::pl;

// element geometry of the unit: one lane pair per element (32 per wavefront), or -- BN_QUAD 1, plk_quad.hip -- one lane quad (16 per wavefront),
// whose two lane pairs compute the same values: sub-pair 0 stores
#if BN_QUAD
BN_DEV size_t elem_index(size_t t) { return quad_index(t); }
BN_DEV bool elem_writer(size_t t) { return quad_sub(t) == 0; }
constexpr int ELEMS_PER_BLOCK = BLOCK / 4;
#else
BN_DEV size_t elem_index(size_t t) { return pair_index(t); }
BN_DEV bool elem_writer(size_t) { return true; }
constexpr int ELEMS_PER_BLOCK = BLOCK / 2;
#endif
BN_DEV S2 load_s2(const u64* base, size_t n, size_t i, int w0, int odd) { return S2{load_fp(base, n, i, w0 + 4 * odd)}; }
BN_DEV void store_s2(u64* base, size_t n, size_t i, int w0, int odd, const S2& a) { store_fp(base, n, i, w0 + 4 * odd, a.c); }
BN_DEV void load_s12(S12& r, const u64* base, size_t n, size_t i, int odd) {
  r.c0.c0 = load_s2(base, n, i, 0, odd); r.c0.c1 = load_s2(base, n, i, 8, odd); r.c0.c2 = load_s2(base, n, i, 16, odd);
  r.c1.c0 = load_s2(base, n, i, 24, odd); r.c1.c1 = load_s2(base, n, i, 32, odd); r.c1.c2 = load_s2(base, n, i, 40, odd);
}
BN_DEV void store_s12(u64* base, size_t n, size_t i, int odd, const S12& a) {
  store_s2(base, n, i, 0, odd, a.c0.c0); store_s2(base, n, i, 8, odd, a.c0.c1); store_s2(base, n, i, 16, odd, a.c0.c2);
  store_s2(base, n, i, 24, odd, a.c1.c0); store_s2(base, n, i, 32, odd, a.c1.c1); store_s2(base, n, i, 40, odd, a.c1.c2);
}
// ---- staggered launches (plk_pairing.hip: k_pairing; plk_verify.hip: k_bls_verify_fused) -------------------------------------------------
// Blocks [first, first + count) run only the first half of their element's work (the Miller loop) and park the value; blocks >= nblk finish
// the parked chunks (b - nblk + first).  See k_pairing for why.
struct Stagger {
  unsigned first, count, nblk;
  unsigned mute;                    // test mode (SYLOW_HIP_OPT_STAGGER = 2): parking blocks never publish, so every finishing block takes the recompute fallback
  u64* park;                        // [48][count * BLOCK / 2] raw Miller values
  unsigned* done;                   // [count] set by a parking block when its values are visible
  u64* clk;                         // sylow_hip_clock_probe: [64][4] accumulators, or NULL (the default)
};
// Live clock probe (include/sylow_hip.h: sylow_hip_clock_probe).  Every wavefront reads the shader-clock counter and the constant-rate counter
// when it starts and when it ends; lane 0 adds the two deltas, a wavefront count and the longest residency into slot blockIdx % 64.  Four scalar
// registers across the kernel and four fire-and-forget atomics per wavefront (32 768 wavefronts per 2^20 pairings): not measurable in the kernel's
// time (same-box A/B: profiles/r06_ab/r05_vs_r06.log).
struct ClockProbe { u64 c0, w0; };
BN_DEV void probe_begin(ClockProbe& pb, const u64* clk) {
  if (clk) { pb.c0 = __builtin_amdgcn_s_memtime(); pb.w0 = __builtin_amdgcn_s_memrealtime(); }
}
BN_DEV void probe_end(const ClockProbe& pb, u64* clk) {
  if (!clk) return;
  const u64 dc = __builtin_amdgcn_s_memtime() - pb.c0, dw = __builtin_amdgcn_s_memrealtime() - pb.w0;
  if ((threadIdx.x & 63) == 0) {
    u64* a = clk + 4 * (blockIdx.x & 63);
    __hip_atomic_fetch_add(a + 0, dc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_fetch_add(a + 1, dw, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_fetch_add(a + 2, (u64)1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_fetch_max(a + 3, dw, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}
// role of this block -- 0: whole element, 1: Miller loop only (park), 2: final exponentiation only -- and the chunk of elements it works on
BN_DEV int stagger_role(const Stagger& st, unsigned& chunk) {
  chunk = blockIdx.x;
  if (st.count) {
    if (chunk >= st.first && chunk < st.first + st.count) return 1;
    if (chunk >= st.nblk) { chunk = chunk - st.nblk + st.first; return 2; }
  }
  return 0;
}
BN_DEV void stagger_publish(const Stagger& st, unsigned chunk) {       // role 1, after the block's stores to st.park
  __threadfence();
  __syncthreads();
  if (threadIdx.x == 0 && !st.mute) __hip_atomic_store(&st.done[chunk - st.first], 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}
// role 2, before the block's loads from st.park: true when the parked values are visible.  The parking block was dispatched long before this
// one (in-order dispatch; it has normally finished whole rounds ago), so the flag is set on the first look; the wait is BOUNDED (~2 ms) all the
// same, and on false the caller recomputes the chunk from its inputs (role 0 on the same elements) -- no schedule of the dispatcher, no other
// grid sharing the GPU can turn the skew into a deadlock.
BN_DEV bool stagger_wait(const Stagger& st, unsigned chunk) {
  __shared__ int ready;
  if (threadIdx.x == 0) {
    int r = 0;
#pragma unroll 1
    for (int it = 0; it < 4096 && !r; ++it) {
      r = __hip_atomic_load(&st.done[chunk - st.first], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) != 0u;
      if (!r) __builtin_amdgcn_s_sleep(16);
    }
    ready = r;
  }
  __syncthreads();
  // every wavefront reads st.park next: the acquire above was thread 0's; the barrier orders the workgroup but is no agent-scope acquire for
  // the other wavefronts' vector loads in the memory model (one L1 per CU makes it hold in practice) -- one fence per wavefront makes it formal
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  return ready != 0;
}

BN_DEV S2 s2_g2gen_x() { return S2{sel(lane_odd(), fp_const(C_G2_GEN[0]), fp_const(C_G2_GEN[1]))}; }
BN_DEV S2 s2_g2gen_y() { return S2{sel(lane_odd(), fp_const(C_G2_GEN[2]), fp_const(C_G2_GEN[3]))}; }
BN_DEV W2 w2_select(const W2& a, const W2& b, bool c) { return W2{sel9(c, a.c, b.c)}; }     // c ? b : a


// Line table of one G2 point (G2Affine::precompute, pairing.rs:676-708) for the verification kernels, which only need the pairing
// up to factors the final exponentiation removes: every line (l0, l1, l2) is stored divided by l0 -- any Fp2 factor dies in the
// easy part -- as (l1 / l0, l2 / l0) plus a unit word (1; 0 for a line whose l0 is zero, stored undivided), so the accumulator
// update is a 10-product multiplication by (unit + x2 v^2 + x4 v w) instead of the 13-product mul_by_024.
// Layout: [87][2 coefficients][2 coordinates][9 limbs] int32 R-class digits, then [87] unit words.
constexpr int LINE_TABLE_LINES = 87;
constexpr int LINE_TABLE_WORDS = LINE_TABLE_LINES * 36 + LINE_TABLE_LINES;
}  // namespace plk

namespace plkh {
// plk_pairing.hip: fills `sg` for a staggered launch of `nblk` blocks, `full` of them whole chunks (count = 0: plain launch)
// `resident` = blocks of the kernel one CU holds (hipOccupancyMaxActiveBlocksPerMultiprocessor, queried by the caller for ITS kernel)
hipError_t stagger_setup(plk::Stagger& sg, host::Lease& ws, size_t nblk, size_t full, hipStream_t st, int resident);
template <class K> int blocks_per_cu(K kernel) {          // cached per kernel; 2 (what the kernels are built for) if the query fails
  static const int v = [&] {
    int b = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&b, kernel, BLOCK, 0) != hipSuccess || b < 1) { (void)hipGetLastError(); b = 2; }
    return b;
  }();
  return v;
}
}  // namespace plkh
