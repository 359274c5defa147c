// Issue-rate microbenchmark for the integer / fp64 instructions a 256-bit
// Montgomery multiplier can be built from on gfx950.  Each kernel runs a long
// unrolled stream of ONE instruction over NACC independent accumulators and
// reports shader cycles (s_memtime) per wave-instruction, at 1/2/4/8 waves per
// SIMD.  The result decides limb width and is quoted in DESIGN.md.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <string>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { \
  fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 1; } } while (0)

constexpr int ITERS = 2048;   // loop trips
constexpr int UNROLL = 16;    // instructions per accumulator set per trip

template <int OP, int NACC>
__global__ void __launch_bounds__(256) k_rate(uint64_t* out, uint32_t seed) {
  uint64_t acc[NACC];
  uint32_t a = seed * 2654435761u + threadIdx.x * 40503u + 12345u;
  uint32_t b = seed * 40503u + threadIdx.x * 2654435761u + 6789u;
  double fa = 1.0 + (double)(a & 0xffff) * 1e-9, fb = 1.0 + (double)(b & 0xffff) * 1e-9;
  for (int i = 0; i < NACC; ++i) acc[i] = (uint64_t)a * (i + 3) + b;
  uint64_t t0 = __builtin_readcyclecounter();
  for (int it = 0; it < ITERS; ++it) {
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
#pragma unroll
      for (int i = 0; i < NACC; ++i) {
        if constexpr (OP == 0) {        // v_mad_u64_u32  d64 = a*b + c64
          asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b) : "vcc");
        } else if constexpr (OP == 1) { // v_mul_lo_u32
          uint32_t x = (uint32_t)acc[i];
          asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(x) : "v"(a));
          acc[i] = x;
        } else if constexpr (OP == 2) { // v_mul_hi_u32
          uint32_t x = (uint32_t)acc[i];
          asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(x) : "v"(a));
          acc[i] = x;
        } else if constexpr (OP == 3) { // v_fma_f64
          double x = __builtin_bit_cast(double, acc[i]);
          asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(x) : "v"(fa), "v"(fb));
          acc[i] = __builtin_bit_cast(uint64_t, x);
        } else if constexpr (OP == 4) { // v_mad_u32_u24
          uint32_t x = (uint32_t)acc[i];
          asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(x) : "v"(a), "v"(b));
          acc[i] = x;
        } else if constexpr (OP == 5) { // 64-bit add as v_add_co + v_addc_co
          uint32_t lo = (uint32_t)acc[i], hi = (uint32_t)(acc[i] >> 32);
          asm volatile("v_add_co_u32 %0, vcc, %0, %2\n\tv_addc_co_u32 %1, vcc, %1, %3, vcc"
                       : "+v"(lo), "+v"(hi) : "v"(a), "v"(b) : "vcc");
          acc[i] = ((uint64_t)hi << 32) | lo;
        } else if constexpr (OP == 6) { // v_lshl_add_u64
          asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(acc[i]) : "v"(((uint64_t)b << 32) | a));
        } else if constexpr (OP == 7) { // v_mad_u32_u16
          uint32_t x = (uint32_t)acc[i];
          asm volatile("v_mad_u32_u16 %0, %0, %1, %2" : "+v"(x) : "v"(a), "v"(b));
          acc[i] = x;
        } else if constexpr (OP == 8) { // v_add_u32 (full-rate reference)
          uint32_t x = (uint32_t)acc[i];
          asm volatile("v_add_u32 %0, %0, %1" : "+v"(x) : "v"(a));
          acc[i] = x;
        } else if constexpr (OP == 9) { // v_mul_hi_u32_u24
          uint32_t x = (uint32_t)acc[i];
          asm volatile("v_mul_hi_u32_u24 %0, %0, %1" : "+v"(x) : "v"(a));
          acc[i] = x;
        } else if constexpr (OP == 10) { // v_mul_f64
          double x = __builtin_bit_cast(double, acc[i]);
          asm volatile("v_mul_f64 %0, %0, %1" : "+v"(x) : "v"(fa));
          acc[i] = __builtin_bit_cast(uint64_t, x);
        } else if constexpr (OP == 11) { // v_mad_u64_u32 with SGPR carry-out pair, followed by addc on a 3rd word
          uint32_t top = (uint32_t)(acc[(i + 1) % NACC]);
          asm volatile("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc"
                       : "+v"(acc[i]), "+v"(top) : "v"(a), "v"(b) : "vcc");
          acc[(i + 1) % NACC] = (acc[(i + 1) % NACC] & 0xffffffff00000000ull) | top;
        } else if constexpr (OP == 12) { // v_mad_i64_i32
          asm volatile("v_mad_i64_i32 %0, vcc, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b) : "vcc");
        }
      }
    }
  }
  uint64_t t1 = __builtin_readcyclecounter();
  uint64_t s = 0;
  for (int i = 0; i < NACC; ++i) s ^= acc[i];
  // one record per wave: cycles, plus a checksum so nothing is dead
  if ((threadIdx.x & 63) == 0) {
    out[(blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64) * 2 + 0] = t1 - t0;
    out[(blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64) * 2 + 1] = s;
  }
}

template <int OP, int NACC>
int run(const char* name, uint64_t* d_out, int instr_per_step) {
  hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
  int cus = prop.multiProcessorCount;
  for (int wps : {1, 2, 4, 8}) {           // waves per SIMD
    int blocks = cus * wps;                 // 256 threads = 4 waves = 1 per SIMD
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    k_rate<OP, NACC><<<blocks, 256>>>(d_out, 1);  // warm
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    k_rate<OP, NACC><<<blocks, 256>>>(d_out, 2);
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<uint64_t> h(blocks * 4 * 2);
    CHECK(hipMemcpy(h.data(), d_out, h.size() * 8, hipMemcpyDeviceToHost));
    double cyc = 0; for (int i = 0; i < blocks * 4; ++i) cyc += (double)h[2 * i];
    cyc /= (blocks * 4);
    double n_instr = (double)ITERS * UNROLL * NACC * instr_per_step;
    // cycles one SIMD spends per wave-instruction = wave cycles / instrs / waves sharing the SIMD
    double total_wave_instr = n_instr * blocks * 4;
    double ginstr_s = total_wave_instr / (ms * 1e-3) / 1e9;
    printf("%-28s nacc=%d waves/SIMD=%d  wave_cycles/instr=%7.2f  SIMD_cycles/instr=%6.2f  chip Gwave-instr/s=%8.1f  (%.3f ms)\n",
           name, NACC, wps, cyc / n_instr, cyc / n_instr / wps, ginstr_s, ms);
  }
  return 0;
}

int main() {
  uint64_t* d_out; CHECK(hipMalloc(&d_out, 256 * 8 * 4 * 2 * 8 * 4));
  hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
  printf("device %s CUs=%d clock=%d kHz\n", prop.gcnArchName, prop.multiProcessorCount, prop.clockRate);
  run<8, 8>("v_add_u32", d_out, 1);
  run<0, 8>("v_mad_u64_u32", d_out, 1);
  run<0, 1>("v_mad_u64_u32 (dependent)", d_out, 1);
  run<12, 8>("v_mad_i64_i32", d_out, 1);
  run<11, 8>("v_mad_u64_u32+v_addc", d_out, 2);
  run<1, 8>("v_mul_lo_u32", d_out, 1);
  run<2, 8>("v_mul_hi_u32", d_out, 1);
  run<4, 8>("v_mad_u32_u24", d_out, 1);
  run<9, 8>("v_mul_hi_u32_u24", d_out, 1);
  run<7, 8>("v_mad_u32_u16", d_out, 1);
  run<5, 8>("v_add_co+v_addc_co", d_out, 2);
  run<6, 8>("v_lshl_add_u64", d_out, 1);
  run<3, 8>("v_fma_f64", d_out, 1);
  run<3, 1>("v_fma_f64 (dependent)", d_out, 1);
  run<10, 8>("v_mul_f64", d_out, 1);
  return 0;
}
