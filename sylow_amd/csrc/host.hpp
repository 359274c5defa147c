// host.hpp -- host-side plumbing shared by the translation units of libsylow_hip.so: error reporting, launch macros,
// the per-(device, stream) scratch workspace, the per-device G2-generator line tables, and the launchers each unit exports
// to the others.  Every unit compiles its own kernels (no relocatable device code): a kernel and the host function that
// launches it always live in the same .hip file.
#pragma once
#include "common.hpp"

#include <cstdio>
#include <cstdlib>
#include <cstring>

extern thread_local char sylow_g_err[256];
namespace host {
int32_t fail(hipError_t e, const char* what);

// Scratch workspace (runtime.hip).  A Lease hands out one device block for the duration of ONE entry-point call: blocks are
// keyed per device, a block whose previous user ran on the same stream is reused in stream order, a block last used on another
// stream is reused only after that stream's completion event (the new stream waits on the event; no stored stream handle is
// ever dereferenced), and two host threads never hold the same block.  release() records the completion event.
struct Lease {
  void* p = nullptr;
  int dev = -1, slot = -1;
  hipStream_t st = nullptr;
  int32_t acquire(size_t bytes, hipStream_t stream);
  int32_t release();                 // records the block's completion event on the stream; idempotent
  ~Lease() { release(); }
};
// per-device line tables of the G2 generator (G2Affine::precompute of the constant, pairing.rs:676-708), built on first use
int32_t gen_lines29(const bn254::i32** out, hipStream_t st);    // carry-free lane-pair line table (plk_common.hpp: LINE_TABLE_WORDS)
int32_t g1_gen_comb(const bn254::i32** out, hipStream_t st);     // fixed-base table of the G1 generator (g1.hip), built on first use
int32_t g2_gen_comb(const bn254::i32** out, hipStream_t st);     // fixed-base table of the G2 generator (plk_group.hip), built on first use
void dst_arg(DstPrime& dp, const uint8_t* dst, size_t len);     // NULL -> sylow's DST (lib.rs:90)
size_t scratch_limit();                                         // sylow_hip_set_scratch_limit: 0 = default
unsigned compute_units();                                       // CUs of the calling thread's current device (0 if unknown)
long long option(int opt);                                      // sylow_hip_set_option: the value in force, -1 = the default (SYLOW_HIP_OPT_*)
inline long long option_or(int opt, long long dflt) { const long long v = option(opt); return v < 0 ? dflt : v; }
uint64_t* clock_probe();                                        // sylow_hip_clock_probe: the accumulator the metric's kernels add to, or NULL
// A short-lived side stream for work that may run BESIDE what the caller's stream holds next (the aggregate verifiers' signature half beside the
// hashing; the lane-quad tail of a batch beside its full rounds).  open(): the side stream waits for everything the caller's stream holds at this
// point; join(): the caller's stream waits for the side work.  Any failure to create the stream or its events degrades to the caller's stream
// (same results, no overlap), and so does enable = false.
struct Fork {
  hipStream_t side = nullptr;
  hipEvent_t ev_fork = nullptr, ev_join = nullptr;
  hipStream_t open(hipStream_t main, bool enable = true) {
    if (!enable) return main;
    if (hipStreamCreateWithFlags(&side, hipStreamNonBlocking) != hipSuccess) { side = nullptr; (void)hipGetLastError(); return main; }
    if (hipEventCreateWithFlags(&ev_fork, hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&ev_join, hipEventDisableTiming) != hipSuccess ||
        hipEventRecord(ev_fork, main) != hipSuccess || hipStreamWaitEvent(side, ev_fork, 0) != hipSuccess) {
      (void)hipGetLastError();
      close();
      return main;
    }
    return side;
  }
  int32_t join(hipStream_t main) {
    if (!side) return SYLOW_HIP_OK;
    hipError_t e = hipEventRecord(ev_join, side);
    if (e == hipSuccess) e = hipStreamWaitEvent(main, ev_join, 0);
    if (e != hipSuccess) { (void)hipStreamSynchronize(side); return fail(e, "join of the side stream"); }
    return SYLOW_HIP_OK;
  }
  void close() {     // destroying a stream / an event with work in flight is deferred by the runtime until that work completes
    if (ev_fork) (void)hipEventDestroy(ev_fork);
    if (ev_join) (void)hipEventDestroy(ev_join);
    if (side) (void)hipStreamDestroy(side);
    ev_fork = ev_join = nullptr; side = nullptr;
  }
  ~Fork() { close(); }
};
}  // namespace host

#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return host::fail(e_, #x); } while (0)
#define ARGCHK(c) do { if (!(c)) { snprintf(sylow_g_err, sizeof(sylow_g_err), "bad argument: %s", #c); return SYLOW_HIP_E_ARG; } } while (0)
#define GRID(n) dim3((unsigned)(((n) + BLOCK - 1) / BLOCK)), dim3(BLOCK), 0, (hipStream_t)stream
#define LAUNCH_RC() (hipGetLastError() == hipSuccess ? SYLOW_HIP_OK : host::fail(hipErrorLaunchFailure, "kernel launch"))
#define LAUNCHED() do { hipError_t e_ = hipGetLastError(); if (e_ != hipSuccess) return host::fail(e_, "kernel launch"); return SYLOW_HIP_OK; } while (0)

// ---- launchers exported between units (argument lists as the C entry points of include/sylow_hip.h) ----------------------
namespace towerh {      // tower.hip: the one-element-per-lane Fp12 selector behind sylow_hip_fp12_hook_batch (ops 0..11)
int32_t fp12_hook(int32_t op, const uint64_t* a, const uint64_t* b, uint64_t* out, size_t n, void* stream);
}  // namespace towerh
namespace g1h {         // g1.hip
size_t g1_comb_bytes();
int32_t build_g1_comb(bn254::i32* table, void* stream);
// hash.hip: H(m_i) (or -H(m_i)) affine, SoA stride n, with the library DST / a caller's DST
int32_t hash_to_g1(const uint8_t* msgs, const uint64_t* msg_offsets, uint64_t* out_xy, uint8_t* out_inf, size_t n, int negate, void* stream);
int32_t hash_to_g1_proj(const uint8_t* msgs, const uint64_t* msg_offsets, uint64_t* acc, size_t n, void* stream);   // projective, into sum_tree's array
// sign_wide.hip: sk_i * H(m_i) on eight lanes per signature (single calls and small batches)
int32_t sign_wide(const uint64_t* sk, const uint8_t* msgs, const uint64_t* msg_offsets, uint64_t* sig_xy, uint8_t* sig_inf, size_t n, void* stream);
int32_t g1_scalar_mul_wide(const uint64_t* p_xy, const uint8_t* p_inf, const uint64_t* k, uint64_t* out_xy, uint8_t* out_inf, size_t n, void* stream);
size_t sign_wide_max();
int32_t evm_ecmul_wide(const uint8_t* in, uint8_t* out, uint8_t* status, size_t n, void* stream);
// sign_wide.hip: the same on eight lanes per message (small batches)
int32_t hash_to_g1_wide(const uint8_t* msgs, const uint64_t* msg_offsets, const DstPrime& dp, uint64_t* out_xy, uint8_t* out_inf, size_t n, int negate, void* stream);
int32_t hash_to_g1_dst(const uint8_t* msgs, const uint64_t* msg_offsets, const DstPrime& dp, uint64_t* out_xy, uint8_t* out_inf, size_t n, int negate, void* stream);
// sum_i P_i (or its negative) -> column `col` of an affine SoA array of stride `stride`; acc: scratch of 12 n words
int32_t sum(const uint64_t* p_xy, const uint8_t* p_inf, size_t n, uint64_t* acc, uint64_t* out_xy, uint8_t* out_inf, size_t stride, size_t col, int negate, void* stream);
// the same from an acc [12][n] that already holds the n projective points
int32_t sum_tree(uint64_t* acc, size_t n, uint64_t* out_xy, uint8_t* out_inf, size_t stride, size_t col, int negate, void* stream);
// the same for the first m elements of an acc whose SoA stride is acc_stride
int32_t sum_tree_strided(uint64_t* acc, size_t acc_stride, size_t m, uint64_t* out_xy, uint8_t* out_inf, size_t stride, size_t col, int negate, void* stream);
}  // namespace g1h
namespace plkh {        // lane-pair units
// selectors of plk_pairing.hip's Fp12 kernel (sylow_hip_fp12_hook_batch, and the Fp12 entry points of tower.hip)
enum { OPW_MUL = 16, OPW_SQR = 17, OPW_SPARSE = 18, OPW_CYCSQR = 19, OPW_FROB1 = 20, OPW_FROB2 = 21, OPW_FROB3 = 22, OPW_EXPZ = 23,
       OPW_S_MUL = 24, OPW_S_SQR = 25, OPW_S_INV = 26, OPW_S_CYCSQR = 27, OPW_CONJ = 28,
       OPW_SPARSE_UNIT = 29, OPW_LAST = 29 };   // 29: first line coefficient = (element index & 1), the other two from `b` as for 18
// plk_pairing.hip: one Fp12 operation on the lane-pair layer (op = an OPW_* selector; OPW_SPARSE: b = 24 words)
int32_t fp12_op(int32_t op, const uint64_t* a, const uint64_t* b, uint64_t* out, size_t n, void* stream);
int32_t build_lines29(const uint64_t* q_xy, size_t n, size_t idx, bn254::i32* table, void* stream);   // plk_verify.hip; q_xy NULL = generator
size_t line_table_bytes();                                                                             // plk_verify.hip
size_t g2_comb_bytes();                                                                                // plk_group.hip
int32_t build_g2_comb(bn254::i32* table, void* stream);                                                // plk_group.hip
// plk_multi.hip: small batches on one wavefront per element (0 from wide_batch_max = route disabled)
size_t wide_batch_max();
size_t wide_verify_max();
int32_t miller_raw_wide_batch(const uint64_t* p_xy, const uint64_t* q_xy, uint64_t* f_out, size_t n, void* stream);
int32_t final_exp_wide_batch(const uint64_t* f, uint64_t* gt_out, size_t n, void* stream);
int32_t verify_two_pairings_wide_batch(const uint64_t* pk_xy, const uint8_t* pk_inf, const uint64_t* h, const uint8_t* h_inf, const uint64_t* sig_xy, const uint8_t* sig_inf,
                                       uint64_t* scratch, uint8_t* ok, size_t n, void* stream);
int32_t pairing_wide_batch(const uint64_t* p_xy, const uint8_t* p_inf, const uint64_t* q_xy, const uint8_t* q_inf, uint64_t* scratch, uint64_t* gt_out, size_t n, void* stream);
int32_t verify_wide_batch(const uint64_t* pk_xy, const uint8_t* pk_inf, const uint64_t* hneg, const uint8_t* hneg_inf, const uint64_t* sig_xy, const uint8_t* sig_inf,
                          uint64_t* scratch, uint8_t* ok, size_t n, void* stream, int one_key = 0);
// plk_quad.hip: mid-size batches on one lane QUAD per element (the lane-pair tower compiled with BN_QUAD 1); 0 from quad_batch_max = route off
size_t quad_batch_max();
size_t tail_split(size_t n);       // the remainder of a batch of whole rounds + a short tail that runs on quads beside the rounds, or 0
int32_t pairing_quad_range(const uint64_t* p_xy, const uint8_t* p_inf, const uint64_t* q_xy, const uint8_t* q_inf, uint64_t* gt_out, size_t n, size_t m, void* stream);
int32_t miller_loop_quad_batch(const uint64_t* p_xy, const uint64_t* q_xy, uint64_t* f_out, size_t n, void* stream);
int32_t final_exp_quad_batch(const uint64_t* f, uint64_t* gt_out, size_t n, void* stream);
int32_t verify_fused_quad(int pk_is_table, const uint64_t* pk_xy, const uint8_t* pk_inf, const bn254::i32* pk_table, const uint64_t* hneg, const uint8_t* hneg_inf,
                          const uint64_t* sig_xy, const uint8_t* sig_inf, const bn254::i32* gen, uint8_t* ok, size_t n, size_t m, void* stream);
// plk_group.hip: EIP-197 pair decoding + validation into SoA arrays (one lane pair per 192-byte pair)
int32_t evm_decode_pairs(const uint8_t* in, size_t n_pairs, uint64_t* pxy, uint8_t* pinf, uint64_t* qxy, uint8_t* qinf, uint8_t* pst, void* stream);
}  // namespace plkh
