// collective.hip -- the only exchange steps of the path, for hosts that are NOT Python: one boolean / one Gt
// from all GPUs of a node through RCCL over xGMI, one process per GPU.
//   * aggregate verify:  AND of every rank's flag vector = MIN all-reduce of one int32 (RCCL has no bit-AND; min over {0,1} is AND)
//   * aggregate product: all-gather of the ranks' 384-byte partial Miller products, then product + ONE final exponentiation on
//     every rank (an Fp12 product is not an RCCL reduction operator)
// `comm` is the caller's ncclComm_t passed as void*.  RCCL is bound at run time (dlopen of librccl.so.1, which resolves to the
// copy the host process already loaded -- e.g. torch's -- because the soname matches), so the library has no link-time
// dependency on it and single-GPU hosts never load it.  comm == NULL means "one rank": the local result is the global result.
#include "host.hpp"

#include <dlfcn.h>

#include <mutex>

#include <rccl/rccl.h>

namespace {
struct Rccl {
  ncclResult_t (*all_reduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*all_gather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*comm_count)(const ncclComm_t, int*) = nullptr;
  const char* (*error_string)(ncclResult_t) = nullptr;
  bool ok = false;
};
const Rccl* rccl() {
  static Rccl r;
  static std::once_flag once;
  std::call_once(once, [] {
    void* h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!h) return;
    r.all_reduce = (decltype(r.all_reduce))dlsym(h, "ncclAllReduce");
    r.all_gather = (decltype(r.all_gather))dlsym(h, "ncclAllGather");
    r.comm_count = (decltype(r.comm_count))dlsym(h, "ncclCommCount");
    r.error_string = (decltype(r.error_string))dlsym(h, "ncclGetErrorString");
    r.ok = r.all_reduce && r.all_gather && r.comm_count;
  });
  return r.ok ? &r : nullptr;
}
int32_t nccl_fail(const Rccl* r, ncclResult_t e, const char* what) {
  snprintf(sylow_g_err, sizeof(sylow_g_err), "%s: %s", what, (r && r->error_string) ? r->error_string(e) : "RCCL error");
  return SYLOW_HIP_E_HIP;
}
}  // namespace

extern "C" {

int32_t sylow_hip_all_valid(const uint8_t* flags, size_t n, void* comm, int32_t* out_dev, void* stream) {
  ARGCHK(out_dev && (flags || !n));
  int32_t rc = sylow_hip_flags_all(flags, n, out_dev, stream);
  if (rc != SYLOW_HIP_OK || !comm) return rc;
  const Rccl* r = rccl();
  if (!r) { snprintf(sylow_g_err, sizeof(sylow_g_err), "librccl.so.1 not found: %s", dlerror()); return SYLOW_HIP_E_HIP; }
  ncclResult_t e = r->all_reduce(out_dev, out_dev, 1, ncclInt32, ncclMin, (ncclComm_t)comm, (hipStream_t)stream);   // 4 bytes over xGMI
  return e == ncclSuccess ? SYLOW_HIP_OK : nccl_fail(r, e, "ncclAllReduce(min)");
}

int32_t sylow_hip_pairing_product_all(const uint64_t* p_xy, const uint8_t* p_inf, const uint64_t* q_xy, const uint8_t* q_inf, size_t n_pairs,
                                      int32_t skip_infinity, void* comm, uint64_t* gt_out, uint8_t* is_one, void* stream) {
  ARGCHK((gt_out || is_one) && (n_pairs == 0 || (p_xy && q_xy)));
  if (!comm) return sylow_hip_pairing_product_batch(p_xy, p_inf, q_xy, q_inf, n_pairs, skip_infinity, gt_out, is_one, stream);
  const Rccl* r = rccl();
  if (!r) { snprintf(sylow_g_err, sizeof(sylow_g_err), "librccl.so.1 not found: %s", dlerror()); return SYLOW_HIP_E_HIP; }
  int world = 0;
  ncclResult_t e = r->comm_count((ncclComm_t)comm, &world);
  if (e != ncclSuccess || world < 1) return nccl_fail(r, e, "ncclCommCount");
  hipStream_t st = (hipStream_t)stream;
  // scratch: this rank's partial [48], the gathered partials rank-major [world][48], and the same as SoA [48][world]
  host::Lease ws;
  int32_t rc = ws.acquire((size_t)(48 + 96 * (size_t)world) * sizeof(u64), st);
  if (rc != SYLOW_HIP_OK) return rc;
  u64 *mine = (u64*)ws.p, *all = mine + 48, *soa = all + 48 * (size_t)world;
  rc = sylow_hip_pairing_product_partial_batch(p_xy, p_inf, q_xy, q_inf, n_pairs, skip_infinity, mine, stream);
  if (rc == SYLOW_HIP_OK) {
    e = r->all_gather(mine, all, 48, ncclUint64, (ncclComm_t)comm, st);                 // 384 bytes per rank
    if (e != ncclSuccess) rc = nccl_fail(r, e, "ncclAllGather");
  }
  if (rc == SYLOW_HIP_OK) rc = sylow_hip_aos_to_soa(all, soa, 48, (size_t)world, stream);
  if (rc == SYLOW_HIP_OK) rc = sylow_hip_fp12_product_final_exp(soa, (size_t)world, gt_out, is_one, stream);
  const int32_t rc2 = ws.release();
  return rc != SYLOW_HIP_OK ? rc : rc2;
}

// Aggregate verification over a batch sharded across the ranks of `comm` (NULL = this process alone): every rank reduces its shard to
// one raw Miller product (sylow_hip_bls_aggregate_partial_batch), the 384-byte partials are all-gathered, and each rank finishes
// product + final exponentiation: one boolean on every rank.
static int32_t aggregate_verify(const uint64_t* pk_xy, const uint8_t* pk_inf, size_t n_pk, const uint8_t* msgs, const uint64_t* msg_offsets,
                                const uint64_t* sig_xy, const uint8_t* sig_inf, const uint64_t* weights, size_t n, void* comm, uint64_t* gt_out, uint8_t* is_one, void* stream) {
  ARGCHK(gt_out || is_one);
  hipStream_t st = (hipStream_t)stream;
  int world = 1;
  const Rccl* r = nullptr;
  if (comm) {
    r = rccl();
    if (!r) { snprintf(sylow_g_err, sizeof(sylow_g_err), "librccl.so.1 not found: %s", dlerror()); return SYLOW_HIP_E_HIP; }
    ncclResult_t e = r->comm_count((ncclComm_t)comm, &world);
    if (e != ncclSuccess || world < 1) return nccl_fail(r, e, "ncclCommCount");
  }
  host::Lease ws;
  int32_t rc = ws.acquire((size_t)(48 + 96 * (size_t)world) * sizeof(u64), st);
  if (rc != SYLOW_HIP_OK) return rc;
  u64 *mine = (u64*)ws.p, *all = mine + 48, *soa = all + 48 * (size_t)world;
  rc = weights ? sylow_hip_bls_weighted_partial_batch(pk_xy, pk_inf, n_pk, msgs, msg_offsets, sig_xy, sig_inf, weights, n, mine, stream)
               : sylow_hip_bls_aggregate_partial_batch(pk_xy, pk_inf, n_pk, msgs, msg_offsets, sig_xy, sig_inf, n, mine, stream);
  if (rc == SYLOW_HIP_OK && comm) {
    ncclResult_t e = r->all_gather(mine, all, 48, ncclUint64, (ncclComm_t)comm, st);                 // 384 bytes per rank
    if (e != ncclSuccess) rc = nccl_fail(r, e, "ncclAllGather");
    if (rc == SYLOW_HIP_OK) rc = sylow_hip_aos_to_soa(all, soa, 48, (size_t)world, stream);
    if (rc == SYLOW_HIP_OK) rc = sylow_hip_fp12_product_final_exp(soa, (size_t)world, gt_out, is_one, stream);
  } else if (rc == SYLOW_HIP_OK) {
    rc = sylow_hip_fp12_product_final_exp(mine, 1, gt_out, is_one, stream);
  }
  const int32_t rc2 = ws.release();
  return rc != SYLOW_HIP_OK ? rc : rc2;
}
int32_t sylow_hip_bls_aggregate_verify_batch(const uint64_t* pk_xy, const uint8_t* pk_inf, size_t n_pk, const uint8_t* msgs, const uint64_t* msg_offsets,
                                             const uint64_t* sig_xy, const uint8_t* sig_inf, size_t n, void* comm, uint64_t* gt_out, uint8_t* is_one, void* stream) {
  return aggregate_verify(pk_xy, pk_inf, n_pk, msgs, msg_offsets, sig_xy, sig_inf, nullptr, n, comm, gt_out, is_one, stream);
}
// The small-exponent batch test: prod_i [e(sig_i, G2gen) e(-H(m_i), pk_i)]^(w_i) == identity with caller-supplied weights
// (sylow_hip_bls_weighted_partial_batch); same sharding as the unweighted aggregate.
int32_t sylow_hip_bls_batch_verify_weighted(const uint64_t* pk_xy, const uint8_t* pk_inf, size_t n_pk, const uint8_t* msgs, const uint64_t* msg_offsets,
                                            const uint64_t* sig_xy, const uint8_t* sig_inf, const uint64_t* weights, size_t n, void* comm, uint64_t* gt_out, uint8_t* is_one, void* stream) {
  ARGCHK(weights || !n);
  return aggregate_verify(pk_xy, pk_inf, n_pk, msgs, msg_offsets, sig_xy, sig_inf, weights, n, comm, gt_out, is_one, stream);
}
}  // extern "C"
