// dpp_probe2.hip -- which (instruction, DPP control, bank_mask) combinations read the partner lane while writing only half the banks
#include <hip/hip_runtime.h>
#include <cstdio>
#define PROBE(idx, text) { int m = 5000 + l; asm volatile("s_nop 1\n\t" text : "+v"(m) : "v"(l), "v"(zero)); out[(idx) * 64 + (int)threadIdx.x] = m; }
__global__ void k(int* out) {
  const int l = threadIdx.x + 100;
  const int zero = 0;
  PROBE(0, "v_mov_b32_dpp %0, %1 row_half_mirror row_mask:0xf bank_mask:0x5")
  PROBE(1, "v_mov_b32_dpp %0, %1 row_mirror row_mask:0xf bank_mask:0x3")
  PROBE(2, "v_subrev_u32_dpp %0, %1, %2 row_mirror row_mask:0xf bank_mask:0x3")
  PROBE(3, "v_sub_u32_dpp %0, %1, %2 row_half_mirror row_mask:0xf bank_mask:0x5")
  PROBE(4, "v_xor_b32_dpp %0, %1, %2 row_half_mirror row_mask:0xf bank_mask:0x5")
  PROBE(5, "v_subrev_u32_dpp %0, %1, %2 row_half_mirror row_mask:0xf bank_mask:0xf")
  PROBE(6, "v_subrev_u32_dpp %0, %1, %2 row_half_mirror row_mask:0xf bank_mask:0xa")
  PROBE(7, "v_add_u32_dpp %0, %1, %2 row_half_mirror row_mask:0xf bank_mask:0x5")
}
int main() {
  int* d; (void)hipMalloc(&d, 8 * 64 * 4);
  k<<<1, 64>>>(d);
  int h[8 * 64]; (void)hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  for (int p = 0; p < 8; ++p) { printf("probe %d:", p); for (int i = 0; i < 16; ++i) printf(" %d", h[p * 64 + i]); printf("\n"); }
  return 0;
}
