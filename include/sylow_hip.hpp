// sylow_hip.hpp -- header-only C++17 host layer over the C ABI (sylow_hip.h).
//
// The reference is a compiled (Rust) library and no Rust toolchain exists in the build image, so
// this is the host side "in the reference's shape" that CAN be compiled here: the same item names
// and argument meaning as sylow's public API for the hot path (src/lib.rs:71-84,179-236;
// src/pairing.rs:870-893,1029-1037), batch-first.  A Rust shim binding the same C entry points
// is listed in INTEGRATION.md.  Host containers are array-of-structs (what a Rust
// Vec<G1Affine> would convert to); the device-side transposition to struct-of-arrays is done by
// sylow_hip_aos_to_soa / _soa_to_aos, so no host loop touches the limbs.
#pragma once
#include <cstdint>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "sylow_hip.h"

namespace sylow {

struct Fp { uint64_t w[4]; };                       // canonical value, little-endian limbs (Fp::value().to_words())
struct Fp2 { Fp c0, c1; };
struct Fp12 { Fp c[12]; };
struct G1Affine { Fp x, y; };                       // 8 words; infinity flags travel separately
struct G2Affine { Fp2 x, y; };                      // 16 words
struct Gt { Fp12 v; };
inline bool operator==(const Gt& a, const Gt& b) {
  for (int i = 0; i < 12; ++i) for (int k = 0; k < 4; ++k) if (a.v.c[i].w[k] != b.v.c[i].w[k]) return false;
  return true;
}
enum class GroupError { NotOnCurve = 1, NotInSubgroup = 2, CannotHashToGroup = 3, DecodeError = 4 };   // groups/group.rs:38-47

struct Error : std::runtime_error { using std::runtime_error::runtime_error; };
inline void check(int32_t rc, const char* what) {
  if (rc != SYLOW_HIP_OK) throw Error(std::string(what) + ": " + sylow_hip_last_error());
}

class DeviceBuffer {                                 // RAII hipMalloc'ed bytes
 public:
  explicit DeviceBuffer(size_t bytes) : bytes_(bytes) { check(sylow_hip_malloc(&p_, bytes), "sylow_hip_malloc"); }
  ~DeviceBuffer() { if (p_) sylow_hip_free(p_); }
  DeviceBuffer(const DeviceBuffer&) = delete;
  DeviceBuffer& operator=(const DeviceBuffer&) = delete;
  DeviceBuffer(DeviceBuffer&& o) noexcept : p_(o.p_), bytes_(o.bytes_) { o.p_ = nullptr; }
  template <class T> T* as() const { return static_cast<T*>(p_); }
  size_t bytes() const { return bytes_; }
 private:
  void* p_ = nullptr;
  size_t bytes_;
};

// host AoS vector of W-word structs -> device SoA [W][n]
template <class T>
DeviceBuffer to_device_soa(const std::vector<T>& v, void* stream = nullptr) {
  constexpr size_t W = sizeof(T) / 8;
  const size_t n = v.size();
  DeviceBuffer aos(n * sizeof(T) + 8), soa(n * sizeof(T) + 8);
  if (n) {
    check(sylow_hip_memcpy_h2d(aos.as<void>(), v.data(), n * sizeof(T), stream), "h2d");
    check(sylow_hip_aos_to_soa(aos.as<uint64_t>(), soa.as<uint64_t>(), W, n, stream), "aos_to_soa");
    check(sylow_hip_stream_sync(stream), "sync");
  }
  return soa;
}
template <class T>
std::vector<T> from_device_soa(const DeviceBuffer& soa, size_t n, void* stream = nullptr) {
  constexpr size_t W = sizeof(T) / 8;
  std::vector<T> out(n);
  if (n) {
    DeviceBuffer aos(n * sizeof(T));
    check(sylow_hip_soa_to_aos(soa.as<uint64_t>(), aos.as<uint64_t>(), W, n, stream), "soa_to_aos");
    check(sylow_hip_memcpy_d2h(out.data(), aos.as<void>(), n * sizeof(T), stream), "d2h");
    check(sylow_hip_stream_sync(stream), "sync");
  }
  return out;
}
inline DeviceBuffer to_device_bytes(const std::vector<uint8_t>& v) {
  DeviceBuffer d(v.size() + 8);
  if (!v.empty()) { check(sylow_hip_memcpy_h2d(d.as<void>(), v.data(), v.size(), nullptr), "h2d"); check(sylow_hip_stream_sync(nullptr), "sync"); }
  return d;
}

// optional identity flags (one byte per element; nullptr = no element is the identity)
struct Flags {
  DeviceBuffer d;
  const uint8_t* ptr;
  Flags(const std::vector<uint8_t>* v, size_t n) : d(n + 8), ptr(nullptr) {
    if (v) {
      if (v->size() != n) throw Error("identity flags: length mismatch");
      if (n) { check(sylow_hip_memcpy_h2d(d.as<void>(), v->data(), n, nullptr), "h2d"); check(sylow_hip_stream_sync(nullptr), "sync"); }
      ptr = d.as<uint8_t>();
    }
  }
};
inline void fetch_flags(std::vector<uint8_t>* out, const DeviceBuffer& d, size_t n) {
  if (!out) return;
  out->resize(n);
  if (n) { check(sylow_hip_memcpy_d2h(out->data(), d.as<void>(), n, nullptr), "d2h"); check(sylow_hip_stream_sync(nullptr), "sync"); }
}

inline G1Affine g1_generator() { return G1Affine{Fp{{1, 0, 0, 0}}, Fp{{2, 0, 0, 0}}}; }                 // g1.rs:54-60
inline G2Affine g2_generator() {                                                                       // g2.rs:47-77
  return G2Affine{Fp2{Fp{{0x46DEBD5CD992F6EDull, 0x674322D4F75EDADDull, 0x426A00665E5C4479ull, 0x1800DEEF121F1E76ull}},
                      Fp{{0x97E485B7AEF312C2ull, 0xF1AA493335A9E712ull, 0x7260BFB731FB5D25ull, 0x198E9393920D483Aull}}},
                  Fp2{Fp{{0x4CE6CC0166FA7DAAull, 0xE3D1E7690C43D37Bull, 0x4AAB71808DCB408Full, 0x12C85EA5DB8C6DEBull}},
                      Fp{{0x55ACDADCD122975Bull, 0xBC4B313370B38EF3ull, 0xEC9E99AD690C3395ull, 0x090689D0585FF075ull}}}};
}

// pairing(&G1Projective, &G2Projective) -> Gt (pairing.rs:870-893), elementwise over the batch
inline std::vector<Gt> pairing(const std::vector<G1Affine>& p, const std::vector<G2Affine>& q,
                               const std::vector<uint8_t>* p_inf = nullptr, const std::vector<uint8_t>* q_inf = nullptr) {
  if (p.size() != q.size()) throw Error("pairing: length mismatch");
  const size_t n = p.size();
  if ((p_inf && p_inf->size() != n) || (q_inf && q_inf->size() != n)) throw Error("identity flags: length mismatch");
  // host vectors in, host vector out: the chunked, double-buffered pipeline (copies of chunk k - 1 / k + 1 beside the kernels of chunk k)
  std::vector<Gt> out(n);
  check(sylow_hip_pairing_host(reinterpret_cast<const uint64_t*>(p.data()), p_inf ? p_inf->data() : nullptr, reinterpret_cast<const uint64_t*>(q.data()),
                               q_inf ? q_inf->data() : nullptr, reinterpret_cast<uint64_t*>(out.data()), n, 0), "sylow_hip_pairing_host");
  return out;
}
// the same through explicit upload -> sylow_hip_pairing_batch -> download on one stream (what the pipeline must equal bit for bit)
inline std::vector<Gt> pairing_unpipelined(const std::vector<G1Affine>& p, const std::vector<G2Affine>& q,
                                           const std::vector<uint8_t>* p_inf = nullptr, const std::vector<uint8_t>* q_inf = nullptr) {
  if (p.size() != q.size()) throw Error("pairing: length mismatch");
  const size_t n = p.size();
  auto dp = to_device_soa(p); auto dq = to_device_soa(q);
  DeviceBuffer dgt(n * sizeof(Gt) + 8);
  Flags dpi(p_inf, n), dqi(q_inf, n);
  check(sylow_hip_pairing_batch(dp.as<uint64_t>(), dpi.ptr, dq.as<uint64_t>(), dqi.ptr, dgt.as<uint64_t>(), n, nullptr), "sylow_hip_pairing_batch");
  return from_device_soa<Gt>(dgt, n);
}
// glued_pairing(&[G1Projective], &[G2Projective]) -> Gt (pairing.rs:1029-1037): ONE product.  Identity flags as in the reference:
// skip_infinity = false replays it (flags only shape Z of the G2 point, a G2 identity zeroes the product),
// true drops pairs with an identity on either side (EIP-197).
inline Gt glued_pairing(const std::vector<G1Affine>& g1s, const std::vector<G2Affine>& g2s,
                        const std::vector<uint8_t>* g1_inf = nullptr, const std::vector<uint8_t>* g2_inf = nullptr, bool skip_infinity = false) {
  const size_t k = g1s.size() < g2s.size() ? g1s.size() : g2s.size();        // zip truncates (pairing.rs:975)
  std::vector<G1Affine> a(g1s.begin(), g1s.begin() + k);
  std::vector<G2Affine> b(g2s.begin(), g2s.begin() + k);
  std::vector<uint8_t> ai, bi;
  if ((g1_inf && g1_inf->size() < k) || (g2_inf && g2_inf->size() < k)) throw Error("glued_pairing: identity flags shorter than the point list");
  if (g1_inf) ai.assign(g1_inf->begin(), g1_inf->begin() + k);
  if (g2_inf) bi.assign(g2_inf->begin(), g2_inf->begin() + k);
  auto dp = to_device_soa(a); auto dq = to_device_soa(b);
  Flags dpi(g1_inf ? &ai : nullptr, k), dqi(g2_inf ? &bi : nullptr, k);
  DeviceBuffer dgt(sizeof(Gt));
  // the whole batch as one product, spread over the GPU (chunked Miller loops, product tree, one final exponentiation)
  check(sylow_hip_pairing_product_batch(dp.as<uint64_t>(), dpi.ptr, dq.as<uint64_t>(), dqi.ptr, k, skip_infinity ? 1 : 0, dgt.as<uint64_t>(), nullptr, nullptr),
        "sylow_hip_pairing_product_batch");
  return from_device_soa<Gt>(dgt, 1)[0];
}
// Mul<&Fp> for G1 / G2 (group.rs:639-667), elementwise
// inf_out receives the identity flags of the results (k = 0, k = r, identity in -> identity out); p_inf marks identity inputs
inline std::vector<G1Affine> mul(const std::vector<G1Affine>& p, const std::vector<Fp>& k, std::vector<uint8_t>* inf_out = nullptr,
                                 const std::vector<uint8_t>* p_inf = nullptr) {
  if (p.size() != k.size()) throw Error("G1 * Fp: length mismatch");
  const size_t n = p.size();
  auto dp = to_device_soa(p); auto dk = to_device_soa(k);
  Flags dpi(p_inf, n);
  DeviceBuffer dout(n * sizeof(G1Affine) + 8), dinf(n + 8);
  check(sylow_hip_g1_scalar_mul_batch(dp.as<uint64_t>(), dpi.ptr, dk.as<uint64_t>(), dout.as<uint64_t>(), dinf.as<uint8_t>(), n, nullptr), "g1_scalar_mul");
  fetch_flags(inf_out, dinf, n);
  return from_device_soa<G1Affine>(dout, n);
}
// in_subgroup: every p[i] is in the r-torsion (what G2Projective::new guarantees upstream): the endomorphism-split product
inline std::vector<G2Affine> mul(const std::vector<G2Affine>& p, const std::vector<Fp>& k, std::vector<uint8_t>* inf_out = nullptr,
                                 const std::vector<uint8_t>* p_inf = nullptr, bool in_subgroup = false) {
  if (p.size() != k.size()) throw Error("G2 * Fp: length mismatch");
  const size_t n = p.size();
  auto dp = to_device_soa(p); auto dk = to_device_soa(k);
  Flags dpi(p_inf, n);
  DeviceBuffer dout(n * sizeof(G2Affine) + 8), dinf(n + 8);
  check((in_subgroup ? sylow_hip_g2_scalar_mul_subgroup_batch : sylow_hip_g2_scalar_mul_batch)(dp.as<uint64_t>(), dpi.ptr, dk.as<uint64_t>(), dout.as<uint64_t>(), dinf.as<uint8_t>(), n, nullptr), "g2_scalar_mul");
  fetch_flags(inf_out, dinf, n);
  return from_device_soa<G2Affine>(dout, n);
}
// Mul<&Fr> for &Gt (groups/gt.rs:161-187), elementwise: gt[i] "times" k[i]
inline std::vector<Gt> mul(const std::vector<Gt>& gt, const std::vector<Fp>& k) {
  if (gt.size() != k.size()) throw Error("Gt * Fr: length mismatch");
  const size_t n = gt.size();
  auto dg = to_device_soa(gt); auto dk = to_device_soa(k);
  DeviceBuffer dout(n * sizeof(Gt) + 8);
  check(sylow_hip_gt_pow_batch(dg.as<uint64_t>(), dk.as<uint64_t>(), dout.as<uint64_t>(), n, nullptr), "sylow_hip_gt_pow_batch");
  return from_device_soa<Gt>(dout, n);
}
// Fr arithmetic (fields/fp.rs:556-565), elementwise on canonical values carried in the Fp container
namespace fr {
inline std::vector<Fp> binop(int32_t (*fn)(const uint64_t*, const uint64_t*, uint64_t*, size_t, void*), const std::vector<Fp>& a, const std::vector<Fp>& b) {
  if (a.size() != b.size()) throw Error("Fr: length mismatch");
  auto da = to_device_soa(a); auto db = to_device_soa(b);
  DeviceBuffer dout(a.size() * sizeof(Fp) + 8);
  check(fn(da.as<uint64_t>(), db.as<uint64_t>(), dout.as<uint64_t>(), a.size(), nullptr), "fr binop");
  return from_device_soa<Fp>(dout, a.size());
}
inline std::vector<Fp> add(const std::vector<Fp>& a, const std::vector<Fp>& b) { return binop(sylow_hip_fr_add_batch, a, b); }
inline std::vector<Fp> sub(const std::vector<Fp>& a, const std::vector<Fp>& b) { return binop(sylow_hip_fr_sub_batch, a, b); }
inline std::vector<Fp> mul(const std::vector<Fp>& a, const std::vector<Fp>& b) { return binop(sylow_hip_fr_mul_batch, a, b); }
inline std::vector<Fp> inv(const std::vector<Fp>& a) {
  auto da = to_device_soa(a);
  DeviceBuffer dout(a.size() * sizeof(Fp) + 8);
  check(sylow_hip_fr_inv_batch(da.as<uint64_t>(), dout.as<uint64_t>(), a.size(), nullptr), "sylow_hip_fr_inv_batch");
  return from_device_soa<Fp>(dout, a.size());
}
}  // namespace fr
// sum_i k[j][i] * P[j][i] per job (examples/threshold_signing.rs:124-143); rows term-major: row i*n_jobs + j
inline std::vector<G1Affine> aggregate(const std::vector<G1Affine>& p, const std::vector<Fp>& k, size_t n_jobs, size_t n_terms) {
  if (p.size() != n_jobs * n_terms || k.size() != p.size()) throw Error("aggregate: shape mismatch");
  auto dp = to_device_soa(p); auto dk = to_device_soa(k);
  DeviceBuffer dout(n_jobs * sizeof(G1Affine) + 8), dinf(n_jobs + 8);
  check(sylow_hip_g1_lincomb_batch(dp.as<uint64_t>(), nullptr, dk.as<uint64_t>(), dout.as<uint64_t>(), dinf.as<uint8_t>(), n_jobs, n_terms, nullptr), "sylow_hip_g1_lincomb_batch");
  return from_device_soa<G1Affine>(dout, n_jobs);
}
struct Messages {                                    // concatenated bytes + offsets on the device
  DeviceBuffer bytes, offsets; size_t n;
  explicit Messages(const std::vector<std::vector<uint8_t>>& msgs) : bytes(total(msgs) + 8), offsets((msgs.size() + 1) * 8), n(msgs.size()) {
    std::vector<uint8_t> blob; std::vector<uint64_t> off(1, 0);
    for (auto& m : msgs) { blob.insert(blob.end(), m.begin(), m.end()); off.push_back(blob.size()); }
    if (!blob.empty()) check(sylow_hip_memcpy_h2d(bytes.as<void>(), blob.data(), blob.size(), nullptr), "h2d");
    check(sylow_hip_memcpy_h2d(offsets.as<void>(), off.data(), off.size() * 8, nullptr), "h2d");
    check(sylow_hip_stream_sync(nullptr), "sync");
  }
  static size_t total(const std::vector<std::vector<uint8_t>>& msgs) { size_t t = 0; for (auto& m : msgs) t += m.size(); return t; }
};
// sign(&Fp, &[u8]) -> Result<G1Projective, GroupError> (lib.rs:179-187), elementwise; sig_inf receives the identity flags of
// the signatures (k = 0 or a multiple of r signs to the identity)
inline std::vector<G1Affine> sign(const std::vector<Fp>& k, const std::vector<std::vector<uint8_t>>& msgs, std::vector<uint8_t>* sig_inf = nullptr) {
  if (k.size() != msgs.size()) throw Error("sign: length mismatch");
  Messages m(msgs);
  auto dk = to_device_soa(k);
  DeviceBuffer dsig(m.n * sizeof(G1Affine) + 8), dinf(m.n + 8);
  check(sylow_hip_bls_sign_batch(dk.as<uint64_t>(), m.bytes.as<uint8_t>(), m.offsets.as<uint64_t>(), dsig.as<uint64_t>(), dinf.as<uint8_t>(), m.n, nullptr), "sylow_hip_bls_sign_batch");
  fetch_flags(sig_inf, dinf, m.n);
  return from_device_soa<G1Affine>(dsig, m.n);
}
// verify(&G2Projective, &[u8], &G1Projective) -> Result<bool, GroupError> (lib.rs:223-236), elementwise.  An identity key or
// signature is a flag, not a coordinate pair: pairing() maps it to Gt::identity() (pairing.rs:876-886)
inline std::vector<uint8_t> verify(const std::vector<G2Affine>& pubkey, const std::vector<std::vector<uint8_t>>& msgs, const std::vector<G1Affine>& sig,
                                   const std::vector<uint8_t>* pk_inf = nullptr, const std::vector<uint8_t>* sig_inf = nullptr) {
  if (pubkey.size() != msgs.size() || sig.size() != msgs.size()) throw Error("verify: length mismatch");
  const size_t n = msgs.size();
  if ((pk_inf && pk_inf->size() != n) || (sig_inf && sig_inf->size() != n)) throw Error("identity flags: length mismatch");
  std::vector<uint8_t> blob; std::vector<uint64_t> off(1, 0);
  for (auto& m : msgs) { blob.insert(blob.end(), m.begin(), m.end()); off.push_back(blob.size()); }
  if (blob.empty()) blob.push_back(0);
  std::vector<uint8_t> ok(n);
  check(sylow_hip_bls_verify_host(reinterpret_cast<const uint64_t*>(pubkey.data()), pk_inf ? pk_inf->data() : nullptr, blob.data(), off.data(),
                                  reinterpret_cast<const uint64_t*>(sig.data()), sig_inf ? sig_inf->data() : nullptr, ok.data(), n, 0), "sylow_hip_bls_verify_host");
  return ok;
}
inline std::vector<uint8_t> verify_unpipelined(const std::vector<G2Affine>& pubkey, const std::vector<std::vector<uint8_t>>& msgs, const std::vector<G1Affine>& sig,
                                               const std::vector<uint8_t>* pk_inf = nullptr, const std::vector<uint8_t>* sig_inf = nullptr) {
  if (pubkey.size() != msgs.size() || sig.size() != msgs.size()) throw Error("verify: length mismatch");
  Messages m(msgs);
  auto dpk = to_device_soa(pubkey); auto dsig = to_device_soa(sig);
  Flags dpi(pk_inf, m.n), dsi(sig_inf, m.n);
  DeviceBuffer dok(m.n + 8);
  check(sylow_hip_bls_verify_batch(dpk.as<uint64_t>(), dpi.ptr, m.bytes.as<uint8_t>(), m.offsets.as<uint64_t>(), dsig.as<uint64_t>(), dsi.ptr, dok.as<uint8_t>(), m.n, nullptr), "sylow_hip_bls_verify_batch");
  std::vector<uint8_t> ok(m.n);
  if (m.n) { check(sylow_hip_memcpy_d2h(ok.data(), dok.as<void>(), m.n, nullptr), "d2h"); check(sylow_hip_stream_sync(nullptr), "sync"); }
  return ok;
}
// page-locked host storage for the pipeline's staging side (every copy asynchronous): a minimal allocator for std::vector
template <class T> struct PinnedAllocator {
  using value_type = T;
  PinnedAllocator() = default;
  template <class U> PinnedAllocator(const PinnedAllocator<U>&) {}
  T* allocate(size_t n) { void* p = nullptr; check(sylow_hip_host_malloc(&p, n * sizeof(T)), "sylow_hip_host_malloc"); return static_cast<T*>(p); }
  void deallocate(T* p, size_t) { sylow_hip_host_free(p); }
  template <class U> bool operator==(const PinnedAllocator<U>&) const { return true; }
  template <class U> bool operator!=(const PinnedAllocator<U>&) const { return false; }
};

// "are ALL of them valid?" as ONE boolean: the glued product of examples/verify_multiple_messages_same_signer.rs:41-60 (weights == nullptr)
// or the sound small-exponent test with the caller's random weights (sylow_hip_bls_batch_verify_weighted).  One key per message, or one key.
inline bool verify_all(const std::vector<G2Affine>& pubkey, const std::vector<std::vector<uint8_t>>& msgs, const std::vector<G1Affine>& sig,
                       const std::vector<Fp>* weights = nullptr, Gt* product = nullptr) {
  if (sig.size() != msgs.size() || (pubkey.size() != msgs.size() && pubkey.size() != 1)) throw Error("verify_all: length mismatch");
  if (weights && weights->size() != msgs.size()) throw Error("verify_all: one weight per signature");
  Messages m(msgs);
  auto dpk = to_device_soa(pubkey); auto dsig = to_device_soa(sig);
  DeviceBuffer dgt(sizeof(Gt) + 8), done(8);
  if (weights) {
    auto dw = to_device_soa(*weights);
    check(sylow_hip_bls_batch_verify_weighted(dpk.as<uint64_t>(), nullptr, pubkey.size(), m.bytes.as<uint8_t>(), m.offsets.as<uint64_t>(), dsig.as<uint64_t>(), nullptr,
                                              dw.as<uint64_t>(), m.n, nullptr, dgt.as<uint64_t>(), done.as<uint8_t>(), nullptr), "sylow_hip_bls_batch_verify_weighted");
  } else {
    check(sylow_hip_bls_aggregate_verify_batch(dpk.as<uint64_t>(), nullptr, pubkey.size(), m.bytes.as<uint8_t>(), m.offsets.as<uint64_t>(), dsig.as<uint64_t>(), nullptr,
                                               m.n, nullptr, dgt.as<uint64_t>(), done.as<uint8_t>(), nullptr), "sylow_hip_bls_aggregate_verify_batch");
  }
  uint8_t one = 0;
  check(sylow_hip_memcpy_d2h(&one, done.as<void>(), 1, nullptr), "d2h"); check(sylow_hip_stream_sync(nullptr), "sync");
  if (product) *product = from_device_soa<Gt>(dgt, 1)[0];
  return one != 0;
}
// sum_i p[i] as ONE point (the `+` fold of examples/verify_multiple_messages_same_signer.rs:41-60); *is_identity receives the flag of the result
inline G1Affine sum(const std::vector<G1Affine>& p, bool* is_identity = nullptr, const std::vector<uint8_t>* p_inf = nullptr) {
  const size_t n = p.size();
  auto dp = to_device_soa(p);
  Flags dpi(p_inf, n);
  DeviceBuffer dout(sizeof(G1Affine) + 8), dinf(8);
  check(sylow_hip_g1_sum_batch(dp.as<uint64_t>(), dpi.ptr, n, dout.as<uint64_t>(), dinf.as<uint8_t>(), nullptr), "sylow_hip_g1_sum_batch");
  std::vector<uint8_t> f;
  fetch_flags(&f, dinf, 1);
  if (is_identity) *is_identity = f[0] != 0;
  return from_device_soa<G1Affine>(dout, 1)[0];
}
// Sub for &G1Projective (group.rs:614-624), elementwise on affine inputs
inline std::vector<G1Affine> sub(const std::vector<G1Affine>& a, const std::vector<G1Affine>& b, std::vector<uint8_t>* inf_out = nullptr) {
  if (a.size() != b.size()) throw Error("G1 - G1: length mismatch");
  const size_t n = a.size();
  auto da = to_device_soa(a); auto db = to_device_soa(b);
  DeviceBuffer dout(n * sizeof(G1Affine) + 8), dinf(n + 8);
  check(sylow_hip_g1_sub_batch(da.as<uint64_t>(), nullptr, db.as<uint64_t>(), nullptr, dout.as<uint64_t>(), dinf.as<uint8_t>(), n, nullptr), "sylow_hip_g1_sub_batch");
  fetch_flags(inf_out, dinf, n);
  return from_device_soa<G1Affine>(dout, n);
}

// ---- runtime knobs of the library (no reference counterpart: the reference is one element on one thread) -----------------------------------
// Route selectors and thresholds (SYLOW_HIP_OPT_*): process-wide, results identical under every setting; value < 0 restores the default.
inline void set_option(int32_t option, int64_t value) { check(sylow_hip_set_option(option, value), "sylow_hip_set_option"); }
inline int64_t get_option(int32_t option) {
  int64_t v = -1;
  check(sylow_hip_get_option(option, &v), "sylow_hip_get_option");
  return v;
}
// Live clock probe of the metric's kernels: arm() zeroes 256 device words and hands them to the library, read() returns the engine clock (MHz)
// the wavefronts launched since arm() ran at (0 if none ran) and disarms.
class ClockProbe {
 public:
  ClockProbe() : acc_(256 * sizeof(uint64_t)) {}
  ~ClockProbe() { sylow_hip_clock_probe(nullptr); }
  void arm() {
    const std::vector<uint64_t> zero(256, 0);
    check(sylow_hip_memcpy_h2d(acc_.as<void>(), zero.data(), zero.size() * 8, nullptr), "h2d"); check(sylow_hip_stream_sync(nullptr), "sync");
    check(sylow_hip_clock_probe(acc_.as<uint64_t>()), "sylow_hip_clock_probe");
  }
  double read_mhz(uint64_t* wavefronts = nullptr) {
    check(sylow_hip_stream_sync(nullptr), "sync");
    check(sylow_hip_clock_probe(nullptr), "sylow_hip_clock_probe");
    std::vector<uint64_t> w(256);
    check(sylow_hip_memcpy_d2h(w.data(), acc_.as<void>(), w.size() * 8, nullptr), "d2h"); check(sylow_hip_stream_sync(nullptr), "sync");
    int32_t khz = 0;
    check(sylow_hip_wall_clock_khz(&khz), "sylow_hip_wall_clock_khz");
    uint64_t clk = 0, wall = 0, waves = 0;
    for (int s = 0; s < 64; ++s) { clk += w[4 * s]; wall += w[4 * s + 1]; waves += w[4 * s + 2]; }
    if (wavefronts) *wavefronts = waves;
    return wall ? (double)clk / (double)wall * khz / 1e3 : 0.0;
  }
 private:
  DeviceBuffer acc_;
};

}  // namespace sylow
