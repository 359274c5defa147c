// Probe: global -> LDS direct loads of 16 bytes per lane on gfx950 (__builtin_amdgcn_global_load_lds), as a prefetch of per-lane table rows.
// Each lane's 16 bytes land at lds_base + lane * 16 (the instruction's own lane striding); read back with a plain LDS load.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__global__ void k(const u32x4* __restrict__ in, u32x4* out, int n_chunks, size_t stride) {
  __shared__ u32x4 buf[7][256];
  const int t = threadIdx.x, wave = t >> 6;
  const size_t g = blockIdx.x * 256 + t;
  for (int c = 0; c < n_chunks; ++c)
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(in + (size_t)c * stride + g),
                                     (__attribute__((address_space(3))) void*)&buf[c][wave * 64], 16, 0, 0);
  __builtin_amdgcn_s_waitcnt(0);          // everything outstanding
  __syncthreads();
  u32x4 acc = {0, 0, 0, 0};
  for (int c = 0; c < n_chunks; ++c) acc += buf[c][t];
  out[g] = acc;
}
int main() {
  const int blocks = 64, n = blocks * 256, chunks = 7;
  u32x4 *in, *out;
  hipMalloc(&in, sizeof(u32x4) * n * chunks); hipMalloc(&out, sizeof(u32x4) * n);
  u32x4* h = (u32x4*)malloc(sizeof(u32x4) * n * chunks);
  for (int i = 0; i < n * chunks; ++i) h[i] = u32x4{(unsigned)i, (unsigned)(i * 3), 7u, (unsigned)(i ^ 0x55)};
  hipMemcpy(in, h, sizeof(u32x4) * n * chunks, hipMemcpyHostToDevice);
  k<<<blocks, 256>>>(in, out, chunks, n);
  u32x4* o = (u32x4*)malloc(sizeof(u32x4) * n);
  hipMemcpy(o, out, sizeof(u32x4) * n, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int g = 0; g < n; ++g) {
    u32x4 e = {0, 0, 0, 0};
    for (int c = 0; c < chunks; ++c) e += h[c * n + g];
    if (e.x != o[g].x || e.y != o[g].y || e.z != o[g].z || e.w != o[g].w) ++bad;
  }
  printf("lds dma probe: %d of %d lanes wrong\n", bad, n);
  return bad != 0;
}
