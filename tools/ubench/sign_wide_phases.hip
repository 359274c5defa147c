// sign_wide_phases.hip -- where one wide signature (sylow_amd/csrc/sign_wide.hip, eight lanes per signature) spends its time: the kernel
// itself, instantiated with clock64() stamps at its phase boundaries, launched for ONE signature.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/ubench/sign_wide_phases.hip -o tools/ubench/sign_wide_phases -Lsylow_amd -lsylow_hip -Wl,-rpath,$PWD/sylow_amd
// (host::dst_arg and the error plumbing come from the library)
#include "../../sylow_amd/csrc/sign_wide.hip"

int main() {
  const size_t n = 1;
  u64 hsk[4] = {0x1234567890abcdefull, 0x0fedcba987654321ull, 0x1111111122222222ull, 0x0123456789abcdefull}, hoff[2] = {0, 32};
  uint8_t hmsg[32]; for (int i = 0; i < 32; ++i) hmsg[i] = (uint8_t)(7 * i + 1);
  u64 *sk, *off, *oxy, *stamps; uint8_t *msg, *oinf;
  (void)hipMalloc(&sk, 32); (void)hipMalloc(&off, 16); (void)hipMalloc(&oxy, 64); (void)hipMalloc(&stamps, 64); (void)hipMalloc(&msg, 32); (void)hipMalloc(&oinf, 1);
  (void)hipMemcpy(sk, hsk, 32, hipMemcpyHostToDevice); (void)hipMemcpy(off, hoff, 16, hipMemcpyHostToDevice); (void)hipMemcpy(msg, hmsg, 32, hipMemcpyHostToDevice);
  DstPrime dp; host::dst_arg(dp, nullptr, 0);
  const char* names[7] = {"expand_message", "front + shared inversion", "svdw_back (3 root candidates)", "exchange + add", "GLV + table", "window loop", "combine + affine + store"};
  for (int rep = 0; rep < 3; ++rep) {
    wsign::k_bls_sign_wide<true><<<1, 64>>>(sk, msg, off, dp, oxy, oinf, n, stamps);
    u64 h[8]; (void)hipMemcpy(h, stamps, 64, hipMemcpyDeviceToHost);
    printf("rep %d: total %.1f us (100 MHz clock ticks: %llu)\n", rep, (h[7] - h[0]) / 100.0, (unsigned long long)(h[7] - h[0]));
    for (int k = 0; k < 7; ++k) printf("   %-28s %8.1f us\n", names[k], (h[k + 1] - h[k]) / 100.0);
  }
  return 0;
}
