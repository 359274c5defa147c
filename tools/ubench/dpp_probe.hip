// dpp_probe.hip -- prints what the DPP controls used by the lane-pair layout do on this GPU: row_half_mirror as a lane permutation, and
// which lanes a bank_mask lets an instruction write.   hipcc --offload-arch=gfx950 dpp_probe.hip -o dpp_probe && ./dpp_probe
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(int* out) {
  const int l = threadIdx.x;
  int partner = __builtin_amdgcn_mov_dpp(l, 0x141, 0xF, 0xF, true);
  int masked = 1000 + l;
  const int zero = 0;
  asm volatile("s_nop 1\n\tv_subrev_u32_dpp %0, %0, %1 row_half_mirror row_mask:0xf bank_mask:0x5" : "+v"(masked) : "v"(zero));
  int sel;
  asm volatile("s_nop 1\n\ts_mov_b32 vcc_lo, 0x0f0f0f0f\n\ts_mov_b32 vcc_hi, 0x0f0f0f0f\n\tv_cndmask_b32_dpp %0, %1, %1, vcc row_half_mirror row_mask:0xf bank_mask:0xf"
               : "=&v"(sel) : "v"(l) : "vcc");
  int m2 = 5000 + l, m3 = 5000 + l, m4 = 5000 + l;
  asm volatile("s_nop 1\n\tv_subrev_u32_dpp %0, %1, %2 row_half_mirror row_mask:0xf bank_mask:0x5" : "+v"(m2) : "v"(l), "v"(zero));
  asm volatile("s_nop 1\n\tv_subrev_u32_dpp %0, %1, %2 row_half_mirror row_mask:0xf bank_mask:0x5 bound_ctrl:0" : "+v"(m3) : "v"(l), "v"(zero));
  asm volatile("s_nop 1\n\tv_mov_b32_dpp %0, %1 row_mirror row_mask:0xf bank_mask:0x3" : "+v"(m4) : "v"(l));
  out[l] = partner; out[64 + l] = masked; out[128 + l] = sel; out[192 + l] = m2; out[256 + l] = m3; out[320 + l] = m4;
}
int main() {
  int* d; hipMalloc(&d, 384 * 4);
  k<<<1, 64>>>(d);
  int h[384]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  printf("partner:"); for (int i = 0; i < 16; ++i) printf(" %d", h[i]); printf("\n");
  printf("masked :"); for (int i = 0; i < 16; ++i) printf(" %d", h[64 + i]); printf("\n");
  printf("select :"); for (int i = 0; i < 16; ++i) printf(" %d", h[128 + i]); printf("\n");
  printf("m2     :"); for (int i = 0; i < 16; ++i) printf(" %d", h[192 + i]); printf("\n");
  printf("m3     :"); for (int i = 0; i < 16; ++i) printf(" %d", h[256 + i]); printf("\n");
  printf("m4     :"); for (int i = 0; i < 16; ++i) printf(" %d", h[320 + i]); printf("\n");
  return 0;
}
