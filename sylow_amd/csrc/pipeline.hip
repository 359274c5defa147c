// pipeline.hip -- the value-typed form of the two headline operations: HOST arrays in, HOST results out.
//
// The reference's API takes and returns values (pairing(&G1Projective, &G2Projective) -> Gt, pairing.rs:870-893;
// verify(&G2Projective, &[u8], &G1Projective) -> bool, lib.rs:223-236), i.e. a host that switches to this library holds
// arrays of structs in host memory (a Rust Vec<G1Affine> is [n][8] words, element-major).  Upload -> compute -> download run
// serially would leave the GPU idle for the transfers (2^20 pairings: 201 MB up + 403 MB down, ~8 % of the 119 ms of compute on
// PCIe Gen5).  Here a batch is cut into a few chunks of growing size (2^16 elements -- one resident set of lane pairs -- then 2^18,
// then the rest; a short last chunk where the results are large: `schedule`) that alternate between TWO streams, each with its own device block:
//
//     stream k & 1:   H2D(chunk k) -> AoS->SoA -> kernels -> SoA->AoS -> D2H(chunk k)
//
// and the host issues them software-pipelined -- H2D + launches of chunk k + 1 BEFORE the D2H of chunk k -- so that whatever the
// copy calls do on pageable memory (the runtime stages them and blocks the calling thread until the stream reaches the copy), the
// next chunk's work is already queued behind the running one: the copy engines move chunk k - 1 out and chunk k + 1 in while the
// compute units run chunk k.  Exposed: the first chunk's upload and the last chunk's download.  With pinned host memory
// (sylow_hip_host_malloc) every call is asynchronous and the same order holds.
// Same kernels, same values as the device-pointer entry points: outputs are bit-identical to the unpipelined call.
#include "host.hpp"
#include "pipeline_schedule.hpp"

#include <algorithm>
#include <vector>

namespace {
constexpr size_t DEFAULT_CHUNK = size_t(1) << 16;

struct Pipe {
  hipStream_t st[2] = {nullptr, nullptr};
  host::Lease blk[2];
  int32_t open(size_t bytes_per_block) {
    for (int i = 0; i < 2; ++i) HIPCHK(hipStreamCreateWithFlags(&st[i], hipStreamNonBlocking));
    for (int i = 0; i < 2; ++i) { const int32_t rc = blk[i].acquire(bytes_per_block, st[i]); if (rc != SYLOW_HIP_OK) return rc; }
    return SYLOW_HIP_OK;
  }
  // drain both streams, hand the blocks back (their completion events are recorded on streams that are about to go away: the
  // streams are synchronised first, so the events have fired), destroy the streams
  int32_t close() {
    hipError_t e = hipSuccess;
    for (int i = 0; i < 2; ++i) if (st[i]) { const hipError_t x = hipStreamSynchronize(st[i]); if (e == hipSuccess) e = x; }
    for (int i = 0; i < 2; ++i) blk[i].release();
    for (int i = 0; i < 2; ++i) if (st[i]) { (void)hipStreamSynchronize(st[i]); (void)hipStreamDestroy(st[i]); st[i] = nullptr; }
    return e == hipSuccess ? SYLOW_HIP_OK : host::fail(e, "pipeline drain");
  }
  ~Pipe() { (void)close(); }
};
inline size_t align256(size_t x) { return (x + 255) & ~size_t(255); }
using pipeline::schedule;
#define RCCHK(x) do { const int32_t rc_ = (x); if (rc_ != SYLOW_HIP_OK) return rc_; } while (0)
}  // namespace

// `WIRE` = false: element-major canonical words ([n][8] / [n][16]) + optional flag arrays.
// `WIRE` = true: the reference's wire format (G1Affine::to_be_bytes 64 bytes, G2Affine::to_be_bytes 128 bytes: g1.rs:151-180,
// g2.rs:319-359), decoded and validated on the device (from_be_bytes + curve + r-torsion checks, what G1Affine::from_be_bytes /
// G2Projective::new do upstream); st_p / st_q [n] receive the per-element status, failed elements enter the pairing as the identity.
template <bool WIRE>
static int32_t pairing_pipeline(const void* p_in, const uint8_t* p_inf, const void* q_in, const uint8_t* q_inf, uint64_t* gt_aos,
                                uint8_t* st_p, uint8_t* st_q, size_t n, size_t chunk) {
  size_t c = 0;
  const std::vector<size_t> cut = schedule(n, chunk ? chunk : DEFAULT_CHUNK, /*large_results=*/true, &c);
  // block layout (bytes), c = the largest chunk: p_in 64 c | q_in 128 c | p_soa 64 c | q_soa 128 c | gt_soa 384 c | gt_aos 384 c | p_inf c | q_inf c | st_p c | st_q c
  const size_t o_pa = 0, o_qa = o_pa + 64 * c, o_ps = o_qa + 128 * c, o_qs = o_ps + 64 * c, o_gs = o_qs + 128 * c, o_ga = o_gs + 384 * c,
               o_pi = o_ga + 384 * c, o_qi = align256(o_pi + c), o_sp = align256(o_qi + c), o_sq = align256(o_sp + c), total = align256(o_sq + c);
  Pipe pp;
  RCCHK(pp.open(total));
  const size_t nchunks = cut.size() - 1;
  auto enqueue = [&](size_t k) -> int32_t {
    const size_t lo = cut[k], m = cut[k + 1] - lo;
    hipStream_t s = pp.st[k & 1];
    char* b = (char*)pp.blk[k & 1].p;
    HIPCHK(hipMemcpyAsync(b + o_pa, (const char*)p_in + 64 * lo, 64 * m, hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(b + o_qa, (const char*)q_in + 128 * lo, 128 * m, hipMemcpyHostToDevice, s));
    const uint8_t *dpi = nullptr, *dqi = nullptr;
    if (WIRE) {
      RCCHK(sylow_hip_g1_from_be_bytes_batch((const uint8_t*)(b + o_pa), (uint64_t*)(b + o_ps), (uint8_t*)(b + o_pi), (uint8_t*)(b + o_sp), m, s));
      RCCHK(sylow_hip_g2_from_be_bytes_batch((const uint8_t*)(b + o_qa), (uint64_t*)(b + o_qs), (uint8_t*)(b + o_qi), (uint8_t*)(b + o_sq), m, s));
      dpi = (const uint8_t*)(b + o_pi); dqi = (const uint8_t*)(b + o_qi);
    } else {
      if (p_inf) { HIPCHK(hipMemcpyAsync(b + o_pi, p_inf + lo, m, hipMemcpyHostToDevice, s)); dpi = (const uint8_t*)(b + o_pi); }
      if (q_inf) { HIPCHK(hipMemcpyAsync(b + o_qi, q_inf + lo, m, hipMemcpyHostToDevice, s)); dqi = (const uint8_t*)(b + o_qi); }
      RCCHK(sylow_hip_aos_to_soa((const uint64_t*)(b + o_pa), (uint64_t*)(b + o_ps), 8, m, s));
      RCCHK(sylow_hip_aos_to_soa((const uint64_t*)(b + o_qa), (uint64_t*)(b + o_qs), 16, m, s));
    }
    RCCHK(sylow_hip_pairing_batch((const uint64_t*)(b + o_ps), dpi, (const uint64_t*)(b + o_qs), dqi, (uint64_t*)(b + o_gs), m, s));
    RCCHK(sylow_hip_soa_to_aos((const uint64_t*)(b + o_gs), (uint64_t*)(b + o_ga), 48, m, s));
    return SYLOW_HIP_OK;
  };
  RCCHK(enqueue(0));
  for (size_t k = 0; k < nchunks; ++k) {
    if (k + 1 < nchunks) RCCHK(enqueue(k + 1));
    const size_t lo = cut[k], m = cut[k + 1] - lo;
    char* b = (char*)pp.blk[k & 1].p;
    if (WIRE) {
      HIPCHK(hipMemcpyAsync(st_p + lo, b + o_sp, m, hipMemcpyDeviceToHost, pp.st[k & 1]));
      HIPCHK(hipMemcpyAsync(st_q + lo, b + o_sq, m, hipMemcpyDeviceToHost, pp.st[k & 1]));
    }
    HIPCHK(hipMemcpyAsync(gt_aos + 48 * lo, b + o_ga, 384 * m, hipMemcpyDeviceToHost, pp.st[k & 1]));
  }
  return pp.close();
}

template <bool WIRE>
static int32_t verify_pipeline(const void* pk_in, const uint8_t* pk_inf, const uint8_t* msgs, const uint64_t* msg_offsets, const void* sig_in,
                               const uint8_t* sig_inf, uint8_t* ok, uint8_t* st_pk, uint8_t* st_sig, size_t n, size_t chunk) {
  size_t c = 0;
  const std::vector<size_t> cut = schedule(n, chunk ? chunk : DEFAULT_CHUNK, /*large_results=*/false, &c);     // one flag byte per element comes back
  const size_t nchunks = cut.size() - 1;
  // the offsets are host memory: check ALL of them (a kernel reads msgs + offsets[i] .. msgs + offsets[i + 1] of a window that holds only
  // [offsets[lo], offsets[hi]) of the blob, so one interior offset out of order would read outside it)
  for (size_t i = 0; i < n; ++i) ARGCHK(msg_offsets[i] <= msg_offsets[i + 1]);
  size_t max_msg = 0;
  for (size_t k = 0; k < nchunks; ++k) max_msg = std::max(max_msg, (size_t)(msg_offsets[cut[k + 1]] - msg_offsets[cut[k]]));
  ARGCHK(max_msg <= (size_t(1) << 46) && c <= (size_t(1) << 32));          // the block layout below cannot overflow size_t
  // block layout: pk_in 128 c | sig_in 64 c | pk_soa 128 c | sig_soa 64 c | offsets 8 (c + 1) | msgs | ok c | pk_inf c | sig_inf c | st_pk c | st_sig c
  const size_t o_ka = 0, o_sa = o_ka + 128 * c, o_ks = o_sa + 64 * c, o_ss = o_ks + 128 * c, o_of = o_ss + 64 * c, o_ms = align256(o_of + 8 * (c + 1)),
               o_ok = align256(o_ms + max_msg + 1), o_ki = align256(o_ok + c), o_si = align256(o_ki + c), o_tk = align256(o_si + c),
               o_ts = align256(o_tk + c), total = align256(o_ts + c);
  Pipe pp;
  RCCHK(pp.open(total));
  auto enqueue = [&](size_t k) -> int32_t {
    const size_t lo = cut[k], m = cut[k + 1] - lo;
    hipStream_t s = pp.st[k & 1];
    char* b = (char*)pp.blk[k & 1].p;
    const size_t m0 = msg_offsets[lo], mb = msg_offsets[lo + m] - m0;
    HIPCHK(hipMemcpyAsync(b + o_ka, (const char*)pk_in + 128 * lo, 128 * m, hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(b + o_sa, (const char*)sig_in + 64 * lo, 64 * m, hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(b + o_of, msg_offsets + lo, 8 * (m + 1), hipMemcpyHostToDevice, s));
    if (mb) HIPCHK(hipMemcpyAsync(b + o_ms, msgs + m0, mb, hipMemcpyHostToDevice, s));
    const uint8_t *dki = nullptr, *dsi = nullptr;
    if (WIRE) {
      RCCHK(sylow_hip_g2_from_be_bytes_batch((const uint8_t*)(b + o_ka), (uint64_t*)(b + o_ks), (uint8_t*)(b + o_ki), (uint8_t*)(b + o_tk), m, s));
      RCCHK(sylow_hip_g1_from_be_bytes_batch((const uint8_t*)(b + o_sa), (uint64_t*)(b + o_ss), (uint8_t*)(b + o_si), (uint8_t*)(b + o_ts), m, s));
      dki = (const uint8_t*)(b + o_ki); dsi = (const uint8_t*)(b + o_si);
    } else {
      if (pk_inf) { HIPCHK(hipMemcpyAsync(b + o_ki, pk_inf + lo, m, hipMemcpyHostToDevice, s)); dki = (const uint8_t*)(b + o_ki); }
      if (sig_inf) { HIPCHK(hipMemcpyAsync(b + o_si, sig_inf + lo, m, hipMemcpyHostToDevice, s)); dsi = (const uint8_t*)(b + o_si); }
      RCCHK(sylow_hip_aos_to_soa((const uint64_t*)(b + o_ka), (uint64_t*)(b + o_ks), 16, m, s));
      RCCHK(sylow_hip_aos_to_soa((const uint64_t*)(b + o_sa), (uint64_t*)(b + o_ss), 8, m, s));
    }
    // the offsets stay the caller's absolute ones: the message pointer is moved back by the chunk's first offset instead (the kernels
    // only ever read msgs + offsets[i] .. msgs + offsets[i + 1])
    RCCHK(sylow_hip_bls_verify_batch((const uint64_t*)(b + o_ks), dki, (const uint8_t*)(b + o_ms) - m0, (const uint64_t*)(b + o_of),
                                     (const uint64_t*)(b + o_ss), dsi, (uint8_t*)(b + o_ok), m, s));
    return SYLOW_HIP_OK;
  };
  RCCHK(enqueue(0));
  for (size_t k = 0; k < nchunks; ++k) {
    if (k + 1 < nchunks) RCCHK(enqueue(k + 1));
    const size_t lo = cut[k], m = cut[k + 1] - lo;
    char* b = (char*)pp.blk[k & 1].p;
    if (WIRE) {
      HIPCHK(hipMemcpyAsync(st_pk + lo, b + o_tk, m, hipMemcpyDeviceToHost, pp.st[k & 1]));
      HIPCHK(hipMemcpyAsync(st_sig + lo, b + o_ts, m, hipMemcpyDeviceToHost, pp.st[k & 1]));
    }
    HIPCHK(hipMemcpyAsync(ok + lo, b + o_ok, m, hipMemcpyDeviceToHost, pp.st[k & 1]));
  }
  RCCHK(pp.close());
  // A blob that failed decoding / the curve test / the r-torsion test entered the check as the identity, and identity inputs can
  // satisfy the pairing equation (rejected key + all-zero signature: e(O, g2) e(-H(m), O) = 1).  The reference never gets that far --
  // from_be_bytes / G2Projective::new return Err (g1.rs:204-280, g2.rs:460-525) -- so a rejected element is NOT verified, whatever its flag says.
  if (WIRE)
    for (size_t i = 0; i < n; ++i)
      if (st_pk[i] | st_sig[i]) ok[i] = 0;
  return SYLOW_HIP_OK;
}

extern "C" {

int32_t sylow_hip_host_malloc(void** hptr, size_t bytes) { ARGCHK(hptr); HIPCHK(hipHostMalloc(hptr, bytes ? bytes : 1, hipHostMallocDefault)); return SYLOW_HIP_OK; }
int32_t sylow_hip_host_free(void* hptr) { HIPCHK(hipHostFree(hptr)); return SYLOW_HIP_OK; }

int32_t sylow_hip_pairing_host(const uint64_t* p_aos, const uint8_t* p_inf, const uint64_t* q_aos, const uint8_t* q_inf, uint64_t* gt_aos, size_t n, size_t chunk) {
  if (!n) return SYLOW_HIP_OK;
  ARGCHK(p_aos && q_aos && gt_aos);
  return pairing_pipeline<false>(p_aos, p_inf, q_aos, q_inf, gt_aos, nullptr, nullptr, n, chunk);
}
int32_t sylow_hip_pairing_host_bytes(const uint8_t* p_be, const uint8_t* q_be, uint64_t* gt_aos, uint8_t* status_p, uint8_t* status_q, size_t n, size_t chunk) {
  if (!n) return SYLOW_HIP_OK;
  ARGCHK(p_be && q_be && gt_aos && status_p && status_q);
  return pairing_pipeline<true>(p_be, nullptr, q_be, nullptr, gt_aos, status_p, status_q, n, chunk);
}
int32_t sylow_hip_bls_verify_host(const uint64_t* pk_aos, const uint8_t* pk_inf, const uint8_t* msgs, const uint64_t* msg_offsets,
                                  const uint64_t* sig_aos, const uint8_t* sig_inf, uint8_t* ok, size_t n, size_t chunk) {
  if (!n) return SYLOW_HIP_OK;
  ARGCHK(pk_aos && msg_offsets && sig_aos && ok);
  ARGCHK(msgs || msg_offsets[n] == msg_offsets[0]);
  return verify_pipeline<false>(pk_aos, pk_inf, msgs, msg_offsets, sig_aos, sig_inf, ok, nullptr, nullptr, n, chunk);
}
int32_t sylow_hip_bls_verify_host_bytes(const uint8_t* pk_be, const uint8_t* msgs, const uint64_t* msg_offsets, const uint8_t* sig_be,
                                        uint8_t* ok, uint8_t* status_pk, uint8_t* status_sig, size_t n, size_t chunk) {
  if (!n) return SYLOW_HIP_OK;
  ARGCHK(pk_be && msg_offsets && sig_be && ok && status_pk && status_sig);
  ARGCHK(msgs || msg_offsets[n] == msg_offsets[0]);
  return verify_pipeline<true>(pk_be, nullptr, msgs, msg_offsets, sig_be, nullptr, ok, status_pk, status_sig, n, chunk);
}

}  // extern "C"
