// call_cost.hip -- what does one out-of-line call + return cost a wavefront on gfx950?  (round 6: the metric's kernels make 4 690 product-leaf
// calls per wavefront, one every ~340 VALU instructions, and park 11 % of their cycles although the Miller-loop half touches memory nine times
// per pairing.)  A body of B dependent v_mad_u64_u32 is run N times (1) inline in a loop and (2) through a __noinline__ function (s_swappc_b64 +
// s_setpc_b64, the compiler's s_waitcnt at entry / exit included), at one and two wavefronts per SIMD; s_memtime ticks per iteration.  The
// difference is the control-transfer cost as the wavefront sees it (instruction-buffer refill from the instruction cache after each jump) and,
// at two wavefronts per SIMD, what of it the partner hides.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 call_cost.hip -o call_cost
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <algorithm>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 1; } } while (0)

template <int B> __device__ __forceinline__ uint64_t body(uint64_t acc, uint32_t a, uint32_t b) {
#pragma unroll
  for (int i = 0; i < B; ++i) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b) : "vcc");
  return acc;
}
template <int B> __device__ __noinline__ uint64_t body_nl(uint64_t acc, uint32_t a, uint32_t b) { return body<B>(acc, a, b); }

template <int B, bool CALL>
__global__ void __launch_bounds__(256) k_calls(uint64_t* out, uint64_t* ticks, uint32_t seed, int n) {
  uint32_t a = seed * 2654435761u + threadIdx.x * 40503u + 12345u, b = seed * 40503u + threadIdx.x + 7u;
  uint64_t acc = (uint64_t)a * 3 + b;
  const uint64_t t0 = __builtin_readcyclecounter();
#pragma unroll 1
  for (int it = 0; it < n; ++it) acc = CALL ? body_nl<B>(acc, a, b) : body<B>(acc, a, b);
  const uint64_t t1 = __builtin_readcyclecounter();
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
  if ((threadIdx.x & 63) == 0) ticks[blockIdx.x * 4 + threadIdx.x / 64] = t1 - t0;
}

template <int B> int run(int cus) {
  const int n = 4096;
  uint64_t *out, *ticks;
  for (int wps : {1, 2}) {
    const int blocks = cus * wps;                        // 4 wavefronts per block: one per SIMD; wps blocks per CU
    CHECK(hipMalloc(&out, (size_t)blocks * 256 * 8)); CHECK(hipMalloc(&ticks, (size_t)blocks * 4 * 8));
    double res[2];
    for (int call = 0; call < 2; ++call) {
      std::vector<double> meds;
      for (int rep = 0; rep < 5; ++rep) {
        if (call) k_calls<B, true><<<blocks, 256>>>(out, ticks, 17 + rep, n); else k_calls<B, false><<<blocks, 256>>>(out, ticks, 17 + rep, n);
        CHECK(hipDeviceSynchronize());
        std::vector<uint64_t> h((size_t)blocks * 4);
        CHECK(hipMemcpy(h.data(), ticks, h.size() * 8, hipMemcpyDeviceToHost));
        std::sort(h.begin(), h.end());
        meds.push_back((double)h[h.size() / 2] / n);
      }
      std::sort(meds.begin(), meds.end());
      res[call] = meds[2];
    }
    printf("body of %3d dependent v_mad_u64_u32, %d wavefront(s) per SIMD: inline %8.1f ticks / iteration, out of line %8.1f, call + return = %6.1f ticks (%.1f %% of the out-of-line iteration)\n",
           B, wps, res[0], res[1], res[1] - res[0], 100.0 * (res[1] - res[0]) / res[1]);
    CHECK(hipFree(out)); CHECK(hipFree(ticks));
  }
  return 0;
}
int main() {
  hipDeviceProp_t p; CHECK(hipGetDeviceProperties(&p, 0));
  printf("device %s CUs=%d\n", p.gcnArchName, p.multiProcessorCount);
  if (run<8>(p.multiProcessorCount)) return 1;
  if (run<81>(p.multiProcessorCount)) return 1;
  if (run<243>(p.multiProcessorCount)) return 1;
  return 0;
}
