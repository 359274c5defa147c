// mfma_overlap.hip -- does the MFMA pipe run beside the VALU multiply-add stream of the pairing's product leaf?
//
// The i8-MFMA offload of the Montgomery reduction's constant-operand product (docs/REJECTED.md "MFMA") was priced serially in round 4.  This
// probe measures the premise at the pairing kernel's occupancy (256 threads, two blocks per CU, 2 waves per SIMD), HIP-event timed.
//   part 1, ONE instruction stream:  valu = 8 x v_mad_i64_i32 per iteration (two independent chains, the leaf's instruction),
//           mfma = 1 x v_mfma_i32_32x32x16_i8 per iteration (four independent accumulators), both = the two interleaved in the same wave.
//   part 2, TWO wavefronts per SIMD with different roles: every wave is either a multiply-add wave or an MFMA wave; the role is
//           (wave in block + block group) & 1 under two guesses of which blocks share a CU (b, b + 256 | b, b + 1) -- the guess that mixes
//           roles on a SIMD shows the overlap, the other one is the control.  Each role also runs alone (the other role's waves exit).
// If the pipes overlap, t(both) ~ max(t(valu), t(mfma)); if they serialise, t(both) ~ t(valu) + t(mfma).
//   hipcc -O3 --offload-arch=gfx950 tools/ubench/mfma_overlap.hip -o tools/ubench/mfma_overlap
#include <hip/hip_runtime.h>
#include <cstdio>

typedef int v16i __attribute__((ext_vector_type(16)));

// roles: bit 0 = this wave runs the multiply-add stream, bit 1 = the MFMA stream
__device__ __forceinline__ long long work(int roles, int iters, int seed) {
  long long x = threadIdx.x + seed, y = blockIdx.x + 3;
  int a = (int)threadIdx.x * 7 + 1, b = seed | 1;
  v16i c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
  const long ma = 0x0102030405060708L + threadIdx.x, mb = 0x0807060504030201L + blockIdx.x;
#pragma unroll 1
  for (int i = 0; i < iters; i += 4) {
    if (roles & 1) {
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        asm volatile("v_mad_i64_i32 %0, vcc, %1, %2, %0" : "+v"(x) : "v"(a), "v"(b) : "vcc");
        asm volatile("v_mad_i64_i32 %0, vcc, %1, %2, %0" : "+v"(y) : "v"(b), "v"(a) : "vcc");
      }
    }
    if (roles & 2) {
      c0 = __builtin_amdgcn_mfma_i32_32x32x16_i8(ma, mb, c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_i32_32x32x16_i8(ma, mb, c1, 0, 0, 0);
      c2 = __builtin_amdgcn_mfma_i32_32x32x16_i8(ma, mb, c2, 0, 0, 0);
      c3 = __builtin_amdgcn_mfma_i32_32x32x16_i8(ma, mb, c3, 0, 0, 0);
    }
  }
  long long r = x ^ y;
  for (int j = 0; j < 16; ++j) r += c0[j] + c1[j] + c2[j] + c3[j];
  return r;
}
// mode 0: every wave takes `roles`.  mode 1 / 2: split roles, blocks (b, b + 256) / (b, b + 1) assumed to share a CU; `roles` masks which role works.
__global__ void __launch_bounds__(256, 2) k(long long* out, int iters, int seed, int mode, int roles) {
  int r = roles;
  if (mode) {
    const int wave = threadIdx.x >> 6, group = mode == 1 ? (int)(blockIdx.x >> 8) : (int)(blockIdx.x & 1);
    r = (((wave + group) & 1) ? 2 : 1) & roles;
  }
  r = __builtin_amdgcn_readfirstlane(r);
  out[blockIdx.x * 256 + threadIdx.x] = r ? work(r, iters, seed) : 0;
}

float run(long long* d, int iters, int mode, int roles) {
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  k<<<512, 256>>>(d, iters, 1, mode, roles); (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  for (int r = 0; r < 5; ++r) k<<<512, 256>>>(d, iters, r + 2, mode, roles);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
  return ms / 5;
}

int main() {
  long long* d; (void)hipMalloc(&d, 512 * 256 * 8);
  const int iters = 1 << 15;
  printf("512 blocks x 256 threads, %d iterations of (8 v_mad_i64_i32 | 1 v_mfma_i32_32x32x16_i8)\n", iters);
  {
    const float tv = run(d, iters, 0, 1), tm = run(d, iters, 0, 2), tb = run(d, iters, 0, 3);
    printf("one stream per wave:   valu %8.3f ms   mfma %8.3f ms   both %8.3f ms   (sum %.3f, max %.3f)   overlap = (sum - both) / min = %.2f\n",
           tv, tm, tb, tv + tm, tv > tm ? tv : tm, (tv + tm - tb) / (tv < tm ? tv : tm));
  }
  for (int mode = 1; mode <= 2; ++mode) {
    const float tv = run(d, iters, mode, 1), tm = run(d, iters, mode, 2), tb = run(d, iters, mode, 3);
    printf("roles split, guess %d:  valu waves alone %8.3f ms   mfma waves alone %8.3f ms   together %8.3f ms   (sum %.3f, max %.3f)   overlap = %.2f\n",
           mode, tv, tm, tb, tv + tm, tv > tm ? tv : tm, (tv + tm - tb) / (tv < tm ? tv : tm));
  }
  return 0;
}
