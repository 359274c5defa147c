// runtime.hip -- libsylow_hip.so runtime: device selection, memory helpers, the scratch workspace, per-device generator tables,
// and the HBM-bound Fp / Fr micro-batch kernels.  gfx950 only.  No CPU fallback: every entry point launches HIP kernels or fails.
#include "host.hpp"

#include <atomic>
#include <mutex>
#include <vector>

// ------------------------------------------------------------------ Fp kernels ----------------

// HBM-bound kernels (96 B per element): each lane handles TWO adjacent elements so that every limb
// plane is read / written with one 16-byte access per lane (1 KiB per wavefront instruction).
// Inputs are reduced like Fp::new by conditional subtraction (no multiplications); a*b is then ONE Barrett
// multiplication in the canonical domain (fp_mulmod_plain) -- no round trip through Montgomery form.
BN_DEV void load_plain2(Fp& e0, Fp& e1, const u64* __restrict__ base, size_t n, size_t i) {
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
    const u64x2 w = __builtin_nontemporal_load(reinterpret_cast<const u64x2*>(base + (size_t)k * n + i));
    e0.v[2 * k] = (u32)w.x; e0.v[2 * k + 1] = (u32)(w.x >> 32);
    e1.v[2 * k] = (u32)w.y; e1.v[2 * k + 1] = (u32)(w.y >> 32);
  }
}
BN_DEV void store_plain2(u64* __restrict__ base, size_t n, size_t i, const Fp& e0, const Fp& e1) {
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
    u64x2 w;
    w.x = (u64)e0.v[2 * k] | ((u64)e0.v[2 * k + 1] << 32);
    w.y = (u64)e1.v[2 * k] | ((u64)e1.v[2 * k + 1] << 32);
    __builtin_nontemporal_store(w, reinterpret_cast<u64x2*>(base + (size_t)k * n + i));
  }
}
template <int OP, int FR>
BN_DEV Fp fp_binop_one(const Fp& x, const Fp& y) {
  if (FR) {                                            // the scalar field, same macro-generated API (fp.rs:556-565)
    if (OP == OP_MUL) return fr_mulmod_inline(x, y);
    Fp xr = fr_reduce_plain(x), yr = fr_reduce_plain(y);
    return (OP == OP_ADD) ? fr_add(xr, yr) : fr_sub(xr, yr);
  }
  if (OP == OP_MUL) return fp_mulmod_plain(x, y);      // Barrett takes any 256-bit operands
  Fp xr = fp_reduce_plain(x), yr = fp_reduce_plain(y);
  return (OP == OP_ADD) ? fp_add(xr, yr) : fp_sub(xr, yr);
}
template <int OP, int FR>
__global__ void __launch_bounds__(BLOCK) k_fp_binop(const u64* __restrict__ a, const u64* __restrict__ b, u64* __restrict__ out, size_t n) {
  size_t i = 2 * TID;
  if (i >= n) return;
  const bool vec = ((n & 1) == 0);          // planes stay 16-byte aligned only for even n
  if (vec) {
    Fp x0, x1, y0, y1;
    load_plain2(x0, x1, a, n, i);
    load_plain2(y0, y1, b, n, i);
    store_plain2(out, n, i, fp_binop_one<OP, FR>(x0, y0), fp_binop_one<OP, FR>(x1, y1));
  } else {
    for (size_t j = i; j < n && j < i + 2; ++j)
      store_plain(out, n, j, 0, fp_binop_one<OP, FR>(load_plain(a, n, j, 0), load_plain(b, n, j, 0)));
  }
}
template <int OP, int FR>
BN_DEV Fp fp_unop_one(const Fp& x) {
  if (FR) {
    if (OP == OP_SQR) return fr_mulmod_inline(x, x);
    if (OP == OP_NEG) return fr_neg(fr_reduce_plain(x));
    return fr_inv(fr_reduce_plain(x));
  }
  if (OP == OP_SQR) return fp_mulmod_plain(x, x);
  if (OP == OP_NEG) return fp_neg(fp_reduce_plain(x));
  return fp_from_mont(fp_inv(fp_to_mont(x)));
}
template <int OP, int FR>
__global__ void __launch_bounds__(BLOCK) k_fp_unop(const u64* __restrict__ a, u64* __restrict__ out, size_t n) {
  size_t i = 2 * TID;
  if (i >= n) return;
  const bool vec = ((n & 1) == 0);
  if (vec) {
    Fp x0, x1;
    load_plain2(x0, x1, a, n, i);
    store_plain2(out, n, i, fp_unop_one<OP, FR>(x0), fp_unop_one<OP, FR>(x1));
  } else {
    for (size_t j = i; j < n && j < i + 2; ++j) store_plain(out, n, j, 0, fp_unop_one<OP, FR>(load_plain(a, n, j, 0)));
  }
}

// Fp::pow(U256) with a per-element exponent (fp.rs:451-457): uniform 256-step square-and-multiply (a lane whose bit is
// clear multiplies by one); sqrt (fp.rs:611-616); is_square (fp.rs:625-631, as a Jacobi symbol); sgn0 is bit 0 of the value
__global__ void __launch_bounds__(BLOCK) k_fp_pow(const u64* a, const u64* e, u64* out, size_t n) {
  size_t i = TID;
  if (i >= n) return;
  const Fp x = load_fp(a, n, i, 0), one = fp_one();
  const Fp ev = load_plain(e, n, i, 0);
  Fp r = one;
#pragma unroll 1
  for (int b = 255; b >= 0; --b) {
    r = fp_mul(r, r);
    r = fp_mul(r, fp_select(one, x, (ev.v[b >> 5] >> (b & 31)) & 1));
  }
  store_fp(out, n, i, 0, r);
}
__global__ void __launch_bounds__(BLOCK) k_fp_sqrt(const u64* a, u64* out, uint8_t* ok, uint8_t* sq, size_t n) {
  size_t i = TID;
  if (i >= n) return;
  const Fp x = load_fp(a, n, i, 0);
  if (out) {
    const Fp r = fp_mul(x, fp_pow_pm3_quarter(x));            // x^((p+1)/4)
    store_fp(out, n, i, 0, r);
    if (ok) ok[i] = fp_eq(fp_mul(r, r), x) ? 1 : 0;
  }
  if (sq) sq[i] = fp_is_square(x) ? 1 : 0;
}

// test hook for the carry-free core (bn254_f29.hpp): op 0: to_fp(from_fp(a)) (must be a); 1: product through
// f29_mul; 2: a*b + b*a through f29_dot2; 3: lazy (a + b) - b + a normalised then * 1 ... all compared with the
// saturated core by tests/test_gpu_fields.py
__global__ void __launch_bounds__(BLOCK) k_f29_hook(int op, const u64* a, const u64* b, u64* out, size_t n) {
  size_t i = TID;
  if (i >= n) return;
  Fp x = load_fp(a, n, i, 0), y = load_fp(b, n, i, 0), r;
  F29 fx = f29_from_fp(x), fy = f29_from_fp(y);
  if (op == 0) r = f29_to_fp(fx);
  else if (op == 1) r = f29_to_fp(f29_mul(fx, fy));
  else if (op == 2) r = f29_to_fp(f29_dot2(fx, fy, fy, fx));
  else if (op == 4) r = fp_inv(x);                       // safegcd
  else if (op == 5) r = fp_inv_fermat(x);                // x^(p-2) on the carry-free exponentiation chain
  else if (op == 6) r = f29_to_fp(f29_sqr(f29_reduce_from([&](int k) { return (i64)fx.v[k]; })));
  else {
    F29 t = f29_norm(f29_sub(f29_add(f29_add(fx, fy), fx), fy));     // 2x as a lazy value (L <= 3), normalised
    r = f29_to_fp(f29_mul(t, f29_sub(fy, fx)));                        // 2x * (y - x)
  }
  store_fp(out, n, i, 0, r);
}

// ------------------------------------------------------------------ layout helpers --------------
__global__ void __launch_bounds__(BLOCK) k_aos_to_soa(const u64* __restrict__ aos, u64* __restrict__ soa, size_t words, size_t n) {
  size_t t = TID;
  if (t >= words * n) return;
  size_t w = t / n, i = t % n;
  soa[t] = aos[i * words + w];
}
__global__ void __launch_bounds__(BLOCK) k_soa_to_aos(const u64* __restrict__ soa, u64* __restrict__ aos, size_t words, size_t n) {
  size_t t = TID;
  if (t >= words * n) return;
  size_t w = t / n, i = t % n;
  aos[i * words + w] = soa[t];
}
__global__ void __launch_bounds__(BLOCK) k_flags_all(const uint8_t* flags, size_t n, int32_t* out) {
  // out pre-set to 1; any zero flag clears it
  size_t i = TID;
  bool bad = (i < n) && (flags[i] == 0);
  if (__ballot(bad) != 0 && (threadIdx.x & 63) == 0) atomicAnd(out, 0);
}


// ------------------------------------------------------------------ Fp / Fr byte codecs ----------
// Fp::from_be_bytes / Fr::from_be_bytes (fp.rs:686-719, 746-778): 32 big-endian bytes -> CtOption::new(Self::new(v), v < modulus).
// Both halves of the CtOption are produced: the value (v mod modulus, exactly Self::new) and the flag (status DECODE_ERROR for
// v >= modulus).  Fp::to_be_bytes (fp.rs:727-737) is the inverse on the canonical value.
template <int FR>
__global__ void __launch_bounds__(BLOCK) k_fe_from_bytes(const uint8_t* in, u64* out, uint8_t* status, size_t n) {
  const size_t i = TID;
  if (i >= n) return;
  Fp x;
  read_be_fp(x, in + 32 * i);
  const u32 ml[8] = {FR ? 0xf0000001u : BN_P0, FR ? 0x43e1f593u : BN_P1, FR ? 0x79b97091u : BN_P2, FR ? 0x2833e848u : BN_P3,
                     FR ? 0x8181585du : BN_P4, FR ? 0xb85045b6u : BN_P5, FR ? 0xe131a029u : BN_P6, FR ? 0x30644e72u : BN_P7};
  bool lt = false, decided = false;
#pragma unroll
  for (int j = 7; j >= 0; --j) {
    if (!decided && x.v[j] != ml[j]) { lt = x.v[j] < ml[j]; decided = true; }
  }
  store_plain(out, n, i, 0, FR ? fr_reduce_plain(x) : fp_reduce_plain(x));
  status[i] = lt ? SYLOW_HIP_ST_OK : SYLOW_HIP_ST_DECODE_ERROR;
}
template <int FR>
__global__ void __launch_bounds__(BLOCK) k_fe_to_bytes(const u64* a, uint8_t* out, size_t n) {
  const size_t i = TID;
  if (i >= n) return;
  const Fp x = load_plain(a, n, i, 0);
  write_be_fp(out + 32 * i, FR ? fr_reduce_plain(x) : fp_reduce_plain(x));
}

// ================================================================== host state ======================
thread_local char sylow_g_err[256] = "";
namespace host {
int32_t fail(hipError_t e, const char* what) {
  snprintf(sylow_g_err, sizeof(sylow_g_err), "%s: %s", what, hipGetErrorString(e));
  (void)hipGetLastError();          // the failure is reported through the return code: do not leave it sticky for the next launch check
  return SYLOW_HIP_E_HIP;
}
static const uint8_t SYLOW_DST[] = "WARLOCK-CHAOS-V01-CS01-SHA-256";   // lib.rs:90 (30 bytes)
static std::atomic<size_t> g_scratch_limit{0};
size_t scratch_limit() { return g_scratch_limit.load(std::memory_order_relaxed); }
static std::atomic<long long> g_option[SYLOW_HIP_OPT_COUNT];            // 0 = default, else value + 1
long long option(int opt) { return (opt < 0 || opt >= SYLOW_HIP_OPT_COUNT) ? -1 : g_option[opt].load(std::memory_order_relaxed) - 1; }
static std::atomic<uint64_t*> g_clock_probe{nullptr};
uint64_t* clock_probe() { return g_clock_probe.load(std::memory_order_relaxed); }
unsigned compute_units() {
  static std::atomic<unsigned> cache[64];
  int d = 0;
  if (hipGetDevice(&d) != hipSuccess || d < 0 || d >= 64) return 0;
  unsigned v = cache[d].load(std::memory_order_relaxed);
  if (!v) {
    int cu = 0;
    if (hipDeviceGetAttribute(&cu, hipDeviceAttributeMultiprocessorCount, d) != hipSuccess || cu <= 0) return 0;
    v = (unsigned)cu;
    cache[d].store(v, std::memory_order_relaxed);
  }
  return v;
}
void dst_arg(DstPrime& dp, const uint8_t* dst, size_t len) {
  if (!dst) { dst = SYLOW_DST; len = 30; }
  make_dst_prime(dp, dst, len);
}

namespace {
constexpr int MAX_DEV = 64;
struct Block { void* p = nullptr; size_t cap = 0; hipEvent_t done = nullptr; hipStream_t last = nullptr; bool recorded = false, leased = false; };
struct DevState { std::vector<Block> blocks; bn254::i32* gen29 = nullptr; bn254::i32* g2comb = nullptr; bn254::i32* g1comb = nullptr; };
std::mutex g_mu;               // guards g_dev (bookkeeping + one-time table construction); never held across a user kernel
DevState g_dev[MAX_DEV];
int32_t current_device(int& d) {
  HIPCHK(hipGetDevice(&d));
  if (d < 0 || d >= MAX_DEV) { snprintf(sylow_g_err, sizeof(sylow_g_err), "device index out of range"); return SYLOW_HIP_E_ARG; }
  return SYLOW_HIP_OK;
}
}  // namespace

// Block choice: (1) an idle block last used on this stream, (2) an idle block whose completion event has fired, (3) any idle
// block (the new stream then waits for its event on the device, the host does not block), (4) a new block.  Reuse is always
// ordered with hipStreamWaitEvent on the block's own event -- also on the "same" stream, since a destroyed stream's handle
// value can come back for a different stream -- so no stale stream handle is ever passed to HIP.
int32_t Lease::acquire(size_t bytes, hipStream_t stream) {
  std::lock_guard<std::mutex> lock(g_mu);
  int d = 0;
  int32_t rc = current_device(d);
  if (rc != SYLOW_HIP_OK) return rc;
  std::vector<Block>& bl = g_dev[d].blocks;
  int pick = -1;
  for (size_t i = 0; i < bl.size() && pick < 0; ++i) if (!bl[i].leased && bl[i].recorded && bl[i].last == stream) pick = (int)i;
  for (size_t i = 0; i < bl.size() && pick < 0; ++i) if (!bl[i].leased && (!bl[i].recorded || hipEventQuery(bl[i].done) == hipSuccess)) pick = (int)i;
  for (size_t i = 0; i < bl.size() && pick < 0; ++i) if (!bl[i].leased) pick = (int)i;
  (void)hipGetLastError();          // hipEventQuery reports hipErrorNotReady through the sticky error too
  if (pick < 0) {
    Block b;
    HIPCHK(hipEventCreateWithFlags(&b.done, hipEventDisableTiming));
    bl.push_back(b);
    pick = (int)bl.size() - 1;
  }
  Block& b = bl[pick];
  if (b.cap < bytes) {
    if (b.recorded) HIPCHK(hipEventSynchronize(b.done));      // growing frees the old block: its last user must be done
    if (b.p) { HIPCHK(hipFree(b.p)); b.p = nullptr; b.cap = 0; }
    const size_t cap = bytes + (bytes >> 2) + 4096;
    HIPCHK(hipMalloc(&b.p, cap));
    b.cap = cap;
    b.recorded = false;
  } else if (b.recorded) {
    HIPCHK(hipStreamWaitEvent(stream, b.done, 0));
  }
  b.leased = true;
  p = b.p; dev = d; slot = pick; st = stream;
  return SYLOW_HIP_OK;
}
int32_t Lease::release() {
  if (slot < 0) return SYLOW_HIP_OK;
  std::lock_guard<std::mutex> lock(g_mu);
  if ((size_t)slot >= g_dev[dev].blocks.size()) { slot = -1; return SYLOW_HIP_OK; }    // defensive: shutdown refuses while a block is leased
  Block& b = g_dev[dev].blocks[slot];
  slot = -1;
  b.leased = false;
  b.last = st;
  // the caller's thread is still on `dev` (entry points never switch devices)
  hipError_t e = hipEventRecord(b.done, st);
  b.recorded = (e == hipSuccess);
  if (e != hipSuccess) {
    // without an event the block cannot be ordered: drain the stream now so that the next user is safe anyway
    (void)hipStreamSynchronize(st);
    return fail(e, "hipEventRecord(workspace)");
  }
  return SYLOW_HIP_OK;
}

int32_t gen_lines29(const bn254::i32** out, hipStream_t st) {
  std::lock_guard<std::mutex> lock(g_mu);
  int d = 0;
  int32_t rc = current_device(d);
  if (rc != SYLOW_HIP_OK) return rc;
  DevState& D = g_dev[d];
  if (!D.gen29) {
    bn254::i32* t = nullptr;
    HIPCHK(hipMalloc((void**)&t, plkh::line_table_bytes()));
    rc = plkh::build_lines29(nullptr, 0, 0, t, st);
    // one-time: later calls may run on other streams, so the table must be complete before it is published
    hipError_t e = (rc == SYLOW_HIP_OK) ? hipStreamSynchronize(st) : hipSuccess;
    if (rc != SYLOW_HIP_OK || e != hipSuccess) { (void)hipFree(t); return rc != SYLOW_HIP_OK ? rc : fail(e, "generator line table"); }
    D.gen29 = t;
  }
  *out = D.gen29;
  return SYLOW_HIP_OK;
}
int32_t g1_gen_comb(const bn254::i32** out, hipStream_t st) {
  std::lock_guard<std::mutex> lock(g_mu);
  int d = 0;
  int32_t rc = current_device(d);
  if (rc != SYLOW_HIP_OK) return rc;
  DevState& D = g_dev[d];
  if (!D.g1comb) {
    bn254::i32* t = nullptr;
    HIPCHK(hipMalloc((void**)&t, g1h::g1_comb_bytes()));
    rc = g1h::build_g1_comb(t, st);
    hipError_t e = (rc == SYLOW_HIP_OK) ? hipStreamSynchronize(st) : hipSuccess;
    if (rc != SYLOW_HIP_OK || e != hipSuccess) { (void)hipFree(t); return rc != SYLOW_HIP_OK ? rc : fail(e, "G1 generator comb table"); }
    D.g1comb = t;
  }
  *out = D.g1comb;
  return SYLOW_HIP_OK;
}
int32_t g2_gen_comb(const bn254::i32** out, hipStream_t st) {
  std::lock_guard<std::mutex> lock(g_mu);
  int d = 0;
  int32_t rc = current_device(d);
  if (rc != SYLOW_HIP_OK) return rc;
  DevState& D = g_dev[d];
  if (!D.g2comb) {
    bn254::i32* t = nullptr;
    HIPCHK(hipMalloc((void**)&t, plkh::g2_comb_bytes()));
    rc = plkh::build_g2_comb(t, st);
    hipError_t e = (rc == SYLOW_HIP_OK) ? hipStreamSynchronize(st) : hipSuccess;          // published only when complete (see gen_lines29)
    if (rc != SYLOW_HIP_OK || e != hipSuccess) { (void)hipFree(t); return rc != SYLOW_HIP_OK ? rc : fail(e, "G2 generator comb table"); }
    D.g2comb = t;
  }
  *out = D.g2comb;
  return SYLOW_HIP_OK;
}
}  // namespace host

// ================================================================== C ABI ======================
extern "C" {

const char* sylow_hip_last_error(void) { return sylow_g_err; }
int32_t sylow_hip_device_count(void) {
  int c = 0;
  if (hipGetDeviceCount(&c) != hipSuccess) return 0;
  return c;
}
static int32_t check_device(int32_t device) {
  int c = 0;
  if (hipGetDeviceCount(&c) != hipSuccess || c == 0) { snprintf(sylow_g_err, sizeof(sylow_g_err), "no HIP device"); return SYLOW_HIP_E_NO_DEVICE; }
  ARGCHK(device >= 0 && device < c && device < host::MAX_DEV);
  hipDeviceProp_t prop;
  HIPCHK(hipGetDeviceProperties(&prop, device));
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
    snprintf(sylow_g_err, sizeof(sylow_g_err), "device %d is %s; this library is built for gfx950 only", device, prop.gcnArchName);
    return SYLOW_HIP_E_NO_DEVICE;
  }
  return SYLOW_HIP_OK;
}
int32_t sylow_hip_init(int32_t device) {
  int32_t rc = check_device(device);
  if (rc != SYLOW_HIP_OK) return rc;
  HIPCHK(hipSetDevice(device));
  return SYLOW_HIP_OK;
}
int32_t sylow_hip_init_devices(const int32_t* device_ids, int32_t n_dev) {
  ARGCHK(device_ids && n_dev > 0);
  for (int32_t i = 0; i < n_dev; ++i) {
    int32_t rc = check_device(device_ids[i]);
    if (rc != SYLOW_HIP_OK) return rc;
  }
  HIPCHK(hipSetDevice(device_ids[0]));
  return SYLOW_HIP_OK;
}
int32_t sylow_hip_set_device(int32_t device) {
  ARGCHK(device >= 0 && device < host::MAX_DEV);
  HIPCHK(hipSetDevice(device));
  return SYLOW_HIP_OK;
}
int32_t sylow_hip_shutdown(void) {
  std::lock_guard<std::mutex> lock(host::g_mu);
  // another host thread inside an entry point holds a Lease: freeing its block under it would leave its queued kernels on freed
  // memory -- shutdown is refused (nothing is freed) until every entry point has returned
  for (int d = 0; d < host::MAX_DEV; ++d)
    for (const host::Block& b : host::g_dev[d].blocks)
      if (b.leased) { snprintf(sylow_g_err, sizeof(sylow_g_err), "shutdown while an entry point is still running on device %d", d); return SYLOW_HIP_E_ARG; }
  int prev = 0;
  const bool have_prev = hipGetDevice(&prev) == hipSuccess;
  int32_t rc = SYLOW_HIP_OK;
  for (int d = 0; d < host::MAX_DEV; ++d) {
    host::DevState& D = host::g_dev[d];
    if (D.blocks.empty() && !D.gen29 && !D.g2comb && !D.g1comb) continue;
    if (hipSetDevice(d) != hipSuccess) { rc = SYLOW_HIP_E_HIP; continue; }
    hipError_t e = hipDeviceSynchronize();          // nothing may still be reading a block or a table
    if (e != hipSuccess) rc = host::fail(e, "hipDeviceSynchronize(shutdown)");
    for (host::Block& b : D.blocks) {
      if (b.p) (void)hipFree(b.p);
      if (b.done) (void)hipEventDestroy(b.done);
    }
    D.blocks.clear();
    if (D.gen29) { (void)hipFree(D.gen29); D.gen29 = nullptr; }
    if (D.g2comb) { (void)hipFree(D.g2comb); D.g2comb = nullptr; }
    if (D.g1comb) { (void)hipFree(D.g1comb); D.g1comb = nullptr; }
  }
  if (have_prev) (void)hipSetDevice(prev);
  return rc;
}
// Scratch blocks grow with the largest batch a call has seen (the line tables of a 2^20-pair product take ~10 GB) and are kept for reuse.
// trim frees every block that is idle AND whose last user has completed (its event has fired) and is larger than keep_bytes; blocks in
// use, or still referenced by queued work, stay.  Cheap; hosts that share the GPU call it after a large batch.
// hipFree synchronises the device, so the blocks are only UNLINKED under the pool's mutex and freed after it is dropped: other host
// threads keep acquiring / releasing leases while the driver waits for unrelated streams.
int32_t sylow_hip_trim(size_t keep_bytes) {
  std::vector<void*> victims;
  {
    std::lock_guard<std::mutex> lock(host::g_mu);
    int d = 0;
    int32_t rc = host::current_device(d);
    if (rc != SYLOW_HIP_OK) return rc;
    for (host::Block& b : host::g_dev[d].blocks) {
      if (b.leased || !b.p || b.cap <= keep_bytes) continue;
      if (b.recorded && hipEventQuery(b.done) != hipSuccess) continue;
      victims.push_back(b.p);
      b.p = nullptr; b.cap = 0; b.recorded = false;
    }
    (void)hipGetLastError();        // hipEventQuery reports hipErrorNotReady through the sticky error too
  }
  for (void* p : victims) (void)hipFree(p);
  return SYLOW_HIP_OK;
}
// Upper bound for the line tables of the multi-pair routes (plk_multi.hip), the one scratch user whose size is not proportional to its
// input: 0 = the default (12 GB).  Process-wide; read at the start of each call.
int32_t sylow_hip_set_scratch_limit(size_t bytes) { host::g_scratch_limit.store(bytes, std::memory_order_relaxed); return SYLOW_HIP_OK; }
// Route selectors and thresholds (include/sylow_hip.h): the library reads no environment variable
int32_t sylow_hip_set_option(int32_t option, int64_t value) {
  ARGCHK(option >= 0 && option < SYLOW_HIP_OPT_COUNT);
  host::g_option[option].store(value < 0 ? 0 : (long long)value + 1, std::memory_order_relaxed);
  return SYLOW_HIP_OK;
}
int32_t sylow_hip_get_option(int32_t option, int64_t* value_host) {
  ARGCHK(option >= 0 && option < SYLOW_HIP_OPT_COUNT && value_host);
  *value_host = (int64_t)host::option(option);
  return SYLOW_HIP_OK;
}
int32_t sylow_hip_clock_probe(uint64_t* acc) { host::g_clock_probe.store(acc, std::memory_order_relaxed); return SYLOW_HIP_OK; }
int32_t sylow_hip_wall_clock_khz(int32_t* khz_host) {
  ARGCHK(khz_host);
  int d = 0, khz = 0;
  HIPCHK(hipGetDevice(&d));
  HIPCHK(hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, d));
  *khz_host = khz;
  return SYLOW_HIP_OK;
}
int32_t sylow_hip_malloc(void** dptr, size_t bytes) { ARGCHK(dptr); HIPCHK(hipMalloc(dptr, bytes ? bytes : 1)); return SYLOW_HIP_OK; }
int32_t sylow_hip_free(void* dptr) { HIPCHK(hipFree(dptr)); return SYLOW_HIP_OK; }
int32_t sylow_hip_memcpy_h2d(void* dst, const void* src, size_t bytes, void* stream) {
  HIPCHK(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, (hipStream_t)stream)); return SYLOW_HIP_OK;
}
int32_t sylow_hip_memcpy_d2h(void* dst, const void* src, size_t bytes, void* stream) {
  HIPCHK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, (hipStream_t)stream)); return SYLOW_HIP_OK;
}
int32_t sylow_hip_stream_sync(void* stream) { HIPCHK(hipStreamSynchronize((hipStream_t)stream)); return SYLOW_HIP_OK; }
int32_t sylow_hip_aos_to_soa(const uint64_t* aos, uint64_t* soa, size_t words, size_t n, void* stream) {
  ARGCHK(aos && soa); if (!n || !words) return SYLOW_HIP_OK;
  k_aos_to_soa<<<GRID(words * n)>>>(aos, soa, words, n); LAUNCHED();
}
int32_t sylow_hip_soa_to_aos(const uint64_t* soa, uint64_t* aos, size_t words, size_t n, void* stream) {
  ARGCHK(aos && soa); if (!n || !words) return SYLOW_HIP_OK;
  k_soa_to_aos<<<GRID(words * n)>>>(soa, aos, words, n); LAUNCHED();
}

#define FP_BIN(field, FR, name, OP)                                                                                 \
  int32_t sylow_hip_##field##_##name##_batch(const uint64_t* a, const uint64_t* b, uint64_t* out, size_t n, void* stream) { \
    ARGCHK(a && b && out); if (!n) return SYLOW_HIP_OK;                                                           \
    k_fp_binop<OP, FR><<<GRID((n + 1) / 2)>>>(a, b, out, n); LAUNCHED();                                             \
  }
FP_BIN(fp, 0, add, OP_ADD) FP_BIN(fp, 0, sub, OP_SUB) FP_BIN(fp, 0, mul, OP_MUL)
FP_BIN(fr, 1, add, OP_ADD) FP_BIN(fr, 1, sub, OP_SUB) FP_BIN(fr, 1, mul, OP_MUL)
#define FP_UN(field, FR, name, OP)                                                                          \
  int32_t sylow_hip_##field##_##name##_batch(const uint64_t* a, uint64_t* out, size_t n, void* stream) {      \
    ARGCHK(a && out); if (!n) return SYLOW_HIP_OK;                                                        \
    k_fp_unop<OP, FR><<<GRID((n + 1) / 2)>>>(a, out, n); LAUNCHED();                                         \
  }
FP_UN(fp, 0, sqr, OP_SQR) FP_UN(fp, 0, neg, OP_NEG) FP_UN(fp, 0, inv, OP_INV)
FP_UN(fr, 1, sqr, OP_SQR) FP_UN(fr, 1, neg, OP_NEG) FP_UN(fr, 1, inv, OP_INV)

// FieldExtension<D, N, F> component-wise operators (extensions.rs:67-238: Add / Sub / Neg and scale by a base-field element) for
// Fp2 / Fp6 / Fp12 batches: an extension value is `degree` Fp coefficients, a batch [4 * degree][n] is `degree` consecutive Fp
// planes of [4][n], so each operator is the Fp kernel once per coefficient (HBM-bound like the Fp micro-batches).
#define FEXT_CHECK() ARGCHK(a && out && (degree == 2 || degree == 6 || degree == 12)); if (!n) return SYLOW_HIP_OK
int32_t sylow_hip_fext_add_batch(const uint64_t* a, const uint64_t* b, uint64_t* out, int32_t degree, size_t n, void* stream) {
  FEXT_CHECK(); ARGCHK(b);
  for (int c = 0; c < degree; ++c) k_fp_binop<OP_ADD, 0><<<GRID((n + 1) / 2)>>>(a + 4 * n * c, b + 4 * n * c, out + 4 * n * c, n);
  LAUNCHED();
}
int32_t sylow_hip_fext_sub_batch(const uint64_t* a, const uint64_t* b, uint64_t* out, int32_t degree, size_t n, void* stream) {
  FEXT_CHECK(); ARGCHK(b);
  for (int c = 0; c < degree; ++c) k_fp_binop<OP_SUB, 0><<<GRID((n + 1) / 2)>>>(a + 4 * n * c, b + 4 * n * c, out + 4 * n * c, n);
  LAUNCHED();
}
int32_t sylow_hip_fext_neg_batch(const uint64_t* a, uint64_t* out, int32_t degree, size_t n, void* stream) {
  FEXT_CHECK();
  for (int c = 0; c < degree; ++c) k_fp_unop<OP_NEG, 0><<<GRID((n + 1) / 2)>>>(a + 4 * n * c, out + 4 * n * c, n);
  LAUNCHED();
}
// scale(&self, factor: F) with a base-field factor k_i [4][n] per element (extensions.rs:121-139 applied down to Fp)
int32_t sylow_hip_fext_scale_batch(const uint64_t* a, const uint64_t* k, uint64_t* out, int32_t degree, size_t n, void* stream) {
  FEXT_CHECK(); ARGCHK(k);
  for (int c = 0; c < degree; ++c) k_fp_binop<OP_MUL, 0><<<GRID((n + 1) / 2)>>>(a + 4 * n * c, k, out + 4 * n * c, n);
  LAUNCHED();
}
int32_t sylow_hip_fp_pow_batch(const uint64_t* a, const uint64_t* e, uint64_t* out, size_t n, void* stream) {
  ARGCHK(a && e && out); if (!n) return SYLOW_HIP_OK; k_fp_pow<<<GRID(n)>>>(a, e, out, n); LAUNCHED();
}
int32_t sylow_hip_fp_sqrt_batch(const uint64_t* a, uint64_t* out, uint8_t* is_some, size_t n, void* stream) {
  ARGCHK(a && out && is_some); if (!n) return SYLOW_HIP_OK; k_fp_sqrt<<<GRID(n)>>>(a, out, is_some, nullptr, n); LAUNCHED();
}
int32_t sylow_hip_fp_is_square_batch(const uint64_t* a, uint8_t* flags, size_t n, void* stream) {
  ARGCHK(a && flags); if (!n) return SYLOW_HIP_OK; k_fp_sqrt<<<GRID(n)>>>(a, nullptr, nullptr, flags, n); LAUNCHED();
}

int32_t sylow_hip_fp_from_be_bytes_batch(const uint8_t* in, uint64_t* out, uint8_t* status, size_t n, void* stream) {
  ARGCHK(in && out && status); if (!n) return SYLOW_HIP_OK; k_fe_from_bytes<0><<<GRID(n)>>>(in, out, status, n); LAUNCHED();
}
int32_t sylow_hip_fr_from_be_bytes_batch(const uint8_t* in, uint64_t* out, uint8_t* status, size_t n, void* stream) {
  ARGCHK(in && out && status); if (!n) return SYLOW_HIP_OK; k_fe_from_bytes<1><<<GRID(n)>>>(in, out, status, n); LAUNCHED();
}
int32_t sylow_hip_fp_to_be_bytes_batch(const uint64_t* a, uint8_t* out, size_t n, void* stream) {
  ARGCHK(a && out); if (!n) return SYLOW_HIP_OK; k_fe_to_bytes<0><<<GRID(n)>>>(a, out, n); LAUNCHED();
}
int32_t sylow_hip_fr_to_be_bytes_batch(const uint64_t* a, uint8_t* out, size_t n, void* stream) {
  ARGCHK(a && out); if (!n) return SYLOW_HIP_OK; k_fe_to_bytes<1><<<GRID(n)>>>(a, out, n); LAUNCHED();
}

// Synthetic inputs (BASELINE.md §3 / SURVEY.md §8 d1): the SplitMix64-seeded xoshiro256** stream, 256-bit draws (four outputs,
// least-significant word first) masked to 254 bits and rejection-sampled to < p -- the generator the tests use
// (tests/helpers.py), so bench inputs and oracle inputs are the same stream.  HOST function: out_host is a host array in the
// SoA layout [4][stride], element i of the stream at out_host[w * stride + i].
int32_t sylow_hip_host_xoshiro_fp(uint64_t seed, uint64_t* out_host, size_t n, size_t stride) {
  ARGCHK(out_host && stride >= n);
  uint64_t st[4], x = seed;
  for (int i = 0; i < 4; ++i) {
    x += 0x9E3779B97F4A7C15ull;
    uint64_t z = x;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    st[i] = z ^ (z >> 31);
  }
  auto rotl = [](uint64_t v, int k) { return (v << k) | (v >> (64 - k)); };
  auto next = [&]() {
    const uint64_t r = rotl(st[1] * 5, 7) * 9, t = st[1] << 17;
    st[2] ^= st[0]; st[3] ^= st[1]; st[1] ^= st[2]; st[0] ^= st[3];
    st[2] ^= t;
    st[3] = rotl(st[3], 45);
    return r;
  };
  const uint64_t pw[4] = {0x3c208c16d87cfd47ull, 0x97816a916871ca8dull, 0xb85045b68181585dull, 0x30644e72e131a029ull};
  for (size_t i = 0; i < n;) {
    uint64_t v[4] = {next(), next(), next(), next()};
    v[3] &= (1ull << 62) - 1;
    bool lt = false;
    for (int k = 3; k >= 0; --k) if (v[k] != pw[k]) { lt = v[k] < pw[k]; break; }
    if (!lt) continue;
    for (int k = 0; k < 4; ++k) out_host[(size_t)k * stride + i] = v[k];
    ++i;
  }
  return SYLOW_HIP_OK;
}

// test hook (see k_f29_hook)
int32_t sylow_hip_f29_hook_batch(int32_t op, const uint64_t* a, const uint64_t* b, uint64_t* out, size_t n, void* stream) {
  ARGCHK(a && b && out); if (!n) return SYLOW_HIP_OK;
  k_f29_hook<<<GRID(n)>>>(op, a, b, out, n); LAUNCHED();
}

int32_t sylow_hip_flags_all(const uint8_t* flags, size_t n, int32_t* out_dev, void* stream) {
  ARGCHK(out_dev && (flags || !n));
  HIPCHK(hipMemsetD32Async((hipDeviceptr_t)out_dev, 1, 1, (hipStream_t)stream));
  if (!n) return SYLOW_HIP_OK;
  k_flags_all<<<GRID(n)>>>(flags, n, out_dev); LAUNCHED();
}

}  // extern "C"
