/*
 * sylow_hip.h -- C ABI of the MI355X-native batched BN254 pairing / BLS-verify engine.
 *
 * This is the drop-in boundary for the one data-parallel hot path of warlock-labs/sylow
 * (src/fields -> src/groups -> src/pairing.rs -> src/svdw.rs / src/hasher.rs -> lib.rs
 * sign/verify).  The reference has no FFI of its own (100 % safe Rust, `deny(unsafe_code)`,
 * src/lib.rs:63); each entry point below names the reference item whose *batched* form it
 * computes, i.e. what a `sylow-hip` Rust shim binds with `extern "C"`.
 *
 * Conventions
 *  - Every array argument is a DEVICE pointer (hipMalloc / sylow_hip_malloc / a torch tensor's
 *    data_ptr()).  Nothing here takes torch types.
 *  - Field elements cross the boundary as canonical integers in [0, p) (what sylow's
 *    `Fp::value().to_words()` yields, src/fields/fp.rs:232-234): 4 little-endian uint64 limbs.
 *    Inputs >= p are reduced mod p exactly like `Fp::new` (fp.rs:199-201).
 *  - Struct-of-arrays, word-major: a batch of n objects made of W 64-bit words is a uint64
 *    array of shape [W][n]; word w of element i is at base[w * n + i].  Word order inside an
 *    object is the reference's nesting order, least-significant limb first:
 *      Fp   : W =  4  (limb0..limb3)
 *      Fp2  : W =  8  (c0 limbs, c1 limbs)                      fields/fp2.rs
 *      Fp6  : W = 24  (c0.c0, c0.c1, c1.c0, c1.c1, c2.c0, c2.c1) fields/fp6.rs
 *      Fp12 : W = 48  (c0 as Fp6, c1 as Fp6)  == Gt             fields/fp12.rs, groups/gt.rs
 *      G1 affine    : W =  8 (x, y)      + uint8 infinity flag array (may be NULL = none)
 *      G2 affine    : W = 16 (x.c0, x.c1, y.c0, y.c1) + uint8 infinity flag array
 *      G1 projective: W = 12 (x, y, z);  G2 projective: W = 24
 *    The canonical affine encoding of the identity is (0, 1, inf=1) (groups/group.rs:271-277).
 *  - `stream` is a hipStream_t passed as void* (NULL = the default stream).  Calls are
 *    asynchronous on that stream; use sylow_hip_stream_sync or your own events.
 *  - Threads: entry points may be called concurrently from several host threads and on several streams.  The
 *    few calls that need device scratch (pairing_product*, fp12_product_final_exp, evm_ecpairing,
 *    bls_verify_same_signer, *_all) lease a private block per call; blocks are recycled in stream order through
 *    HIP events, never by synchronising a stored stream handle.
 *  - Return value: 0 on success, negative SYLOW_HIP_E_* otherwise.  No call throws or aborts.
 *    Per-element failures are reported through `status` byte arrays using codes that mirror
 *    sylow's GroupError (groups/group.rs:38-47).
 */
#ifndef SYLOW_HIP_H
#define SYLOW_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SYLOW_HIP_OK 0
#define SYLOW_HIP_E_HIP (-1)      /* a HIP runtime call failed; see sylow_hip_last_error() */
#define SYLOW_HIP_E_ARG (-2)      /* bad argument (NULL pointer, n == 0 where not allowed, ...) */
#define SYLOW_HIP_E_NO_DEVICE (-3)

/* per-element status codes (mirror GroupError, groups/group.rs:38-47) */
#define SYLOW_HIP_ST_OK 0
#define SYLOW_HIP_ST_NOT_ON_CURVE 1
#define SYLOW_HIP_ST_NOT_IN_SUBGROUP 2
#define SYLOW_HIP_ST_CANNOT_HASH 3
#define SYLOW_HIP_ST_DECODE_ERROR 4

/* ---- runtime ------------------------------------------------------------------------------ */
int32_t sylow_hip_init(int32_t device);                 /* hipSetDevice + arch check (gfx950) */
/* One process driving several GPUs: checks every listed device (gfx950) and makes device_ids[0] current.  All per-device
 * state (scratch workspace, generator line tables) is created lazily on the device that is current when a call needs it. */
int32_t sylow_hip_init_devices(const int32_t* device_ids, int32_t n_dev);
/* Launches go to the CALLING THREAD's current HIP device.  Hosts that switch devices (or share the process with code that
 * does) call this before a batch of entry points; pointers passed to a call must belong to that device. */
int32_t sylow_hip_set_device(int32_t device);
/* Frees every scratch block, completion event and generator table on every device (after a device synchronise).  The library
 * stays usable: state is rebuilt on demand.  EXCLUSIVE: while another host thread is inside an entry point that holds a scratch
 * block the call frees nothing and returns SYLOW_HIP_E_ARG. */
int32_t sylow_hip_shutdown(void);
/* Frees the current device's idle scratch blocks larger than keep_bytes whose last user has completed (blocks grow with the largest
 * batch seen and are otherwise kept for reuse until sylow_hip_shutdown).  Never touches a block that is in use and does not hold the
 * library's lock while the driver frees (hipFree may wait for the device: the CALLER can block, other host threads' entry points do not). */
int32_t sylow_hip_trim(size_t keep_bytes);
/* Bounds the one scratch user whose size is NOT proportional to its input: the line tables of the multi-pair routes (multi_pairing_batch,
 * glued_miller_loop_batch, evm_ecpairing_batch with >= 2 pairs per job, pairing_product*, the aggregate verifiers; 19.5 KB per pair, by
 * default as many whole rounds of 2^16 jobs as fit 12 GB).  With a limit the job slices shrink to what fits (slower below one round of
 * k-slot jobs = 1.28 GB * k: the table-driven loop then runs under-filled), and a batch-wide product whose chunks no longer fit takes the
 * in-register schedule (no table at all, ~25 % more Miller-loop work).  Results are identical for every limit.  0 restores the default.
 * Process-wide; takes effect at the next call. */
int32_t sylow_hip_set_scratch_limit(size_t bytes);
/* Route selectors and thresholds (A/B measurements, crossover runs, forcing a route in a test).  Process-wide, read at every call; a value
 * < 0 restores the default.  The LIBRARY reads no environment variable (rounds 1-5 did: eight getenv switches behind this ABI); a host
 * that wants SYLOW_HIP_* variables reads them itself and calls this (sylow_amd/_lib.py does).  Results are identical
 * under every setting.
 *   STAGGER          1 (default) skewed launches of pairing_batch / bls_verify_batch at >= 2 rounds, 0 plain launches, 2 the skew with the
 *                    parking blocks' flags muted (every finishing block takes its recompute fallback)
 *   MULTI_TABLES     default: jobs of >= 2 pairs on average go through line tables in HBM; 0 never, 1 always
 *   WIDE_TAIL        1 (default) one-wavefront-per-element kernels for small batches and single tails, 0 never
 *   WIDE_PACK        two elements per wavefront in those kernels above this many elements (default: the CU count; 0 never, 1 always)
 *   AGG_FORK         1 (default) the aggregate verifiers fork the signature half onto a side stream, 0 one stream
 *   SIGN_WIDE_MAX    largest batch signed / hashed / multiplied on eight lanes per element (default 16384)
 *   WIDE_MAX         largest batch of pairings on the one-wavefront route (default 6144)
 *   WIDE_VERIFY_MAX  largest batch of verifications on it (default 4096)
 *   QUAD_MAX         largest batch of pairings / Miller loops / final exponentiations / verifications on one lane QUAD per element
 *                    (plk_quad.hip; default: 64 x the CU count = 16384, one wavefront per SIMD; 0 never)
 *   TAIL_SPLIT       1 (default) a batch of one or two whole rounds of one wavefront per SIMD (128 x the CU count = 32768 elements each) plus a
 *                    tail that fits the quad route runs the tail on quads on a side stream beside the rounds, 0 one launch */
#define SYLOW_HIP_OPT_STAGGER 0
#define SYLOW_HIP_OPT_MULTI_TABLES 1
#define SYLOW_HIP_OPT_WIDE_TAIL 2
#define SYLOW_HIP_OPT_WIDE_PACK 3
#define SYLOW_HIP_OPT_AGG_FORK 4
#define SYLOW_HIP_OPT_SIGN_WIDE_MAX 5
#define SYLOW_HIP_OPT_WIDE_MAX 6
#define SYLOW_HIP_OPT_WIDE_VERIFY_MAX 7
#define SYLOW_HIP_OPT_QUAD_MAX 8
#define SYLOW_HIP_OPT_TAIL_SPLIT 9
#define SYLOW_HIP_OPT_COUNT 10
int32_t sylow_hip_set_option(int32_t option, int64_t value);
/* @shape value_host=i64[1] */
int32_t sylow_hip_get_option(int32_t option, int64_t* value_host);      /* HOST pointer; -1 = the default is in force */
/* Live clock probe of the metric's kernels (plk::k_pairing, plk::k_bls_verify_fused, and their lane-quad forms for mid-size batches).  `acc` = 256 uint64 words of DEVICE memory, zeroed
 * by the caller (NULL switches the probe off; the default).  While set, every wavefront of those kernels reads the shader-clock counter
 * (s_memtime) and the constant-rate counter (s_memrealtime) when it starts and when it ends and adds, with relaxed device-scope atomics, into
 * slot s = blockIdx % 64:  acc[4 s] += shader-clock ticks, acc[4 s + 1] += constant-rate ticks, acc[4 s + 2] += 1 (wavefronts),
 * acc[4 s + 3] = max(constant-rate ticks of one wavefront).  sum(acc[4 s]) / sum(acc[4 s + 1]) x the constant rate
 * (sylow_hip_wall_clock_khz) is the engine clock those wavefronts ran at, weighted by residency: what bench.py reports as
 * roofline.sustained_mhz.  Process-wide: one accumulator for whichever launches follow, so it must be memory of the device those launches run
 * on (a host that drives several GPUs from one process probes one device at a time); the pointer must stay valid until the probe is switched off
 * and the stream is drained. */
/* @shape acc=u64[256]? */
int32_t sylow_hip_clock_probe(uint64_t* acc);
/* @shape khz_host=i32[1] */
int32_t sylow_hip_wall_clock_khz(int32_t* khz_host);                    /* HOST pointer: rate of s_memrealtime on the current device */
const char* sylow_hip_last_error(void);
int32_t sylow_hip_device_count(void);
int32_t sylow_hip_malloc(void** dptr, size_t bytes);
int32_t sylow_hip_free(void* dptr);
int32_t sylow_hip_memcpy_h2d(void* dst, const void* src, size_t bytes, void* stream);
int32_t sylow_hip_memcpy_d2h(void* dst, const void* src, size_t bytes, void* stream);
int32_t sylow_hip_stream_sync(void* stream);
/* Synthetic inputs: n draws of the SplitMix64-seeded xoshiro256** stream, each 256 bits (four outputs, least-significant
 * word first) masked to 254 bits and rejection-sampled to < p (Fp::rand's role for benches and tests).
 * HOST function: out_host is a HOST array in the SoA layout [4][stride] (stride >= n). */
/* @shape out_host=u64[3*stride+n] */
int32_t sylow_hip_host_xoshiro_fp(uint64_t seed, uint64_t* out_host, size_t n, size_t stride);
/* layout helpers for hosts that hold array-of-structs ([n][W], e.g. a Rust Vec<[u64; 4]>) */
/* @shape aos=u64[words*n] soa=u64[words*n] */
int32_t sylow_hip_aos_to_soa(const uint64_t* aos, uint64_t* soa, size_t words, size_t n, void* stream);
/* @shape soa=u64[words*n] aos=u64[words*n] */
int32_t sylow_hip_soa_to_aos(const uint64_t* soa, uint64_t* aos, size_t words, size_t n, void* stream);

/* ---- Fp: src/fields/fp.rs ----------------------------------------------------------------- */
/* Add / Sub / Mul / Neg / square / Inv for &Fp (fp.rs:304-310, 340-347, 387-393, 442-449,
 * 620-622, 418-433).  inv(0) = 0, no error (fp.rs:1126-1132). */
/* @shape a=u64[4*n] b=u64[4*n] out=u64[4*n] */
int32_t sylow_hip_fp_add_batch(const uint64_t* a, const uint64_t* b, uint64_t* out, size_t n, void* stream);
/* @shape a=u64[4*n] b=u64[4*n] out=u64[4*n] */
int32_t sylow_hip_fp_sub_batch(const uint64_t* a, const uint64_t* b, uint64_t* out, size_t n, void* stream);
/* @shape a=u64[4*n] b=u64[4*n] out=u64[4*n] */
int32_t sylow_hip_fp_mul_batch(const uint64_t* a, const uint64_t* b, uint64_t* out, size_t n, void* stream);
/* @shape a=u64[4*n] out=u64[4*n] */
int32_t sylow_hip_fp_sqr_batch(const uint64_t* a, uint64_t* out, size_t n, void* stream);
/* @shape a=u64[4*n] out=u64[4*n] */
int32_t sylow_hip_fp_neg_batch(const uint64_t* a, uint64_t* out, size_t n, void* stream);
/* @shape a=u64[4*n] out=u64[4*n] */
int32_t sylow_hip_fp_inv_batch(const uint64_t* a, uint64_t* out, size_t n, void* stream);
/* Fp::pow(U256) (fp.rs:451-457), per-element exponents e [4][n] */
/* @shape a=u64[4*n] e=u64[4*n] out=u64[4*n] */
int32_t sylow_hip_fp_pow_batch(const uint64_t* a, const uint64_t* e, uint64_t* out, size_t n, void* stream);
/* Fp::sqrt (fp.rs:611-616): out = a^((p+1)/4), is_some[i] = (out_i^2 == a_i), i.e. the CtOption's value and flag */
/* @shape a=u64[4*n] out=u64[4*n] is_some=u8[n] */
int32_t sylow_hip_fp_sqrt_batch(const uint64_t* a, uint64_t* out, uint8_t* is_some, size_t n, void* stream);
/* Fp::is_square (fp.rs:625-631): 1 for squares and for 0 */
/* @shape a=u64[4*n] flags=u8[n] */
int32_t sylow_hip_fp_is_square_batch(const uint64_t* a, uint8_t* flags, size_t n, void* stream);
/* Fp::from_be_bytes / Fr::from_be_bytes (fp.rs:686-719, 746-778) = CtOption::new(Self::new(v), v < modulus) on 32 big-endian
 * bytes per element, in [n][32]: out [4][n] receives the value (v mod modulus, like Self::new) and status[i] the flag --
 * OK, or DECODE_ERROR where the reference's CtOption is none (v >= p resp. v >= r).  to_be_bytes (fp.rs:727-737) writes the
 * canonical value, out [n][32]. */
/* @shape in=u8[32*n] out=u64[4*n] status=u8[n] */
int32_t sylow_hip_fp_from_be_bytes_batch(const uint8_t* in, uint64_t* out, uint8_t* status, size_t n, void* stream);
/* @shape in=u8[32*n] out=u64[4*n] status=u8[n] */
int32_t sylow_hip_fr_from_be_bytes_batch(const uint8_t* in, uint64_t* out, uint8_t* status, size_t n, void* stream);
/* @shape a=u64[4*n] out=u8[32*n] */
int32_t sylow_hip_fp_to_be_bytes_batch(const uint64_t* a, uint8_t* out, size_t n, void* stream);
/* @shape a=u64[4*n] out=u8[32*n] */
int32_t sylow_hip_fr_to_be_bytes_batch(const uint64_t* a, uint8_t* out, size_t n, void* stream);

/* ---- Fr, the r-torsion scalar field (fields/fp.rs:556-565: the same macro-generated API as Fp, modulus r) -----------
 * Same contract as the Fp calls: [4][n] canonical limbs in and out, any 256-bit input accepted like Fr::new,
 * inv(0) = 0.  Used by Lagrange interpolation / polynomial evaluation (examples/threshold_signing.rs:64-70,146-155). */
/* @shape a=u64[4*n] b=u64[4*n] out=u64[4*n] */
int32_t sylow_hip_fr_add_batch(const uint64_t* a, const uint64_t* b, uint64_t* out, size_t n, void* stream);
/* @shape a=u64[4*n] b=u64[4*n] out=u64[4*n] */
int32_t sylow_hip_fr_sub_batch(const uint64_t* a, const uint64_t* b, uint64_t* out, size_t n, void* stream);
/* @shape a=u64[4*n] b=u64[4*n] out=u64[4*n] */
int32_t sylow_hip_fr_mul_batch(const uint64_t* a, const uint64_t* b, uint64_t* out, size_t n, void* stream);
/* @shape a=u64[4*n] out=u64[4*n] */
int32_t sylow_hip_fr_sqr_batch(const uint64_t* a, uint64_t* out, size_t n, void* stream);
/* @shape a=u64[4*n] out=u64[4*n] */
int32_t sylow_hip_fr_neg_batch(const uint64_t* a, uint64_t* out, size_t n, void* stream);
/* @shape a=u64[4*n] out=u64[4*n] */
int32_t sylow_hip_fr_inv_batch(const uint64_t* a, uint64_t* out, size_t n, void* stream);

/* ---- extension tower (test hooks): fields/fp2.rs:285-306,164-171,355-360; fp6.rs:283-367,
 * 415-423; fp12.rs:229-238,536-550,281-286,515-522,426-503 ------------------------------------ */
/* FieldExtension<D, N, F> component-wise operators (fields/extensions.rs:67-238): Add / Sub / Neg and scale by a base-field
 * element, for Fp2 / Fp6 / Fp12 batches: degree = 2, 6 or 12 Fp coefficients, arrays [4 * degree][n]; the scale factor k is one Fp
 * per element, [4][n]. */
/* @shape a=u64[4*degree*n] b=u64[4*degree*n] out=u64[4*degree*n] */
int32_t sylow_hip_fext_add_batch(const uint64_t* a, const uint64_t* b, uint64_t* out, int32_t degree, size_t n, void* stream);
/* @shape a=u64[4*degree*n] b=u64[4*degree*n] out=u64[4*degree*n] */
int32_t sylow_hip_fext_sub_batch(const uint64_t* a, const uint64_t* b, uint64_t* out, int32_t degree, size_t n, void* stream);
/* @shape a=u64[4*degree*n] out=u64[4*degree*n] */
int32_t sylow_hip_fext_neg_batch(const uint64_t* a, uint64_t* out, int32_t degree, size_t n, void* stream);
/* @shape a=u64[4*degree*n] k=u64[4*n] out=u64[4*degree*n] */
int32_t sylow_hip_fext_scale_batch(const uint64_t* a, const uint64_t* k, uint64_t* out, int32_t degree, size_t n, void* stream);
/* @shape a=u64[8*n] b=u64[8*n] out=u64[8*n] */
int32_t sylow_hip_fp2_mul_batch(const uint64_t* a, const uint64_t* b, uint64_t* out, size_t n, void* stream);
/* @shape a=u64[8*n] out=u64[8*n] */
int32_t sylow_hip_fp2_sqr_batch(const uint64_t* a, uint64_t* out, size_t n, void* stream);
/* @shape a=u64[8*n] out=u64[8*n] */
int32_t sylow_hip_fp2_inv_batch(const uint64_t* a, uint64_t* out, size_t n, void* stream);
/* @shape a=u64[24*n] b=u64[24*n] out=u64[24*n] */
int32_t sylow_hip_fp6_mul_batch(const uint64_t* a, const uint64_t* b, uint64_t* out, size_t n, void* stream);
/* @shape a=u64[24*n] out=u64[24*n] */
int32_t sylow_hip_fp6_inv_batch(const uint64_t* a, uint64_t* out, size_t n, void* stream);
/* The small tower items the pairing composes, as entry points of their own: Fp2::residue_mul (x (9 + u), fp2.rs:99-107),
 * Fp2::frobenius(exponent) (fp2.rs:119-133: conjugation for odd exponents), Fp6::square (fp6.rs:213-236), Fp6::residue_mul
 * (x v, fp6.rs:189-192), Fp6::frobenius(exponent) (fp6.rs:205-211: any exponent, tables indexed mod 6) */
/* @shape a=u64[8*n] out=u64[8*n] */
int32_t sylow_hip_fp2_residue_mul_batch(const uint64_t* a, uint64_t* out, size_t n, void* stream);
/* @shape a=u64[8*n] out=u64[8*n] */
int32_t sylow_hip_fp2_frobenius_batch(const uint64_t* a, uint64_t exponent, uint64_t* out, size_t n, void* stream);
/* @shape a=u64[24*n] out=u64[24*n] */
int32_t sylow_hip_fp6_sqr_batch(const uint64_t* a, uint64_t* out, size_t n, void* stream);
/* @shape a=u64[24*n] out=u64[24*n] */
int32_t sylow_hip_fp6_residue_mul_batch(const uint64_t* a, uint64_t* out, size_t n, void* stream);
/* @shape a=u64[24*n] out=u64[24*n] */
int32_t sylow_hip_fp6_frobenius_batch(const uint64_t* a, uint64_t exponent, uint64_t* out, size_t n, void* stream);
/* @shape a=u64[48*n] b=u64[48*n] out=u64[48*n] */
int32_t sylow_hip_fp12_mul_batch(const uint64_t* a, const uint64_t* b, uint64_t* out, size_t n, void* stream);
/* @shape a=u64[48*n] out=u64[48*n] */
int32_t sylow_hip_fp12_sqr_batch(const uint64_t* a, uint64_t* out, size_t n, void* stream);
/* @shape a=u64[48*n] out=u64[48*n] */
int32_t sylow_hip_fp12_inv_batch(const uint64_t* a, uint64_t* out, size_t n, void* stream);
/* Fp12::frobenius(exponent), exponent in {1,2,3} (the ones the pairing uses) */
/* @shape a=u64[48*n] out=u64[48*n] */
int32_t sylow_hip_fp12_frobenius_batch(const uint64_t* a, int32_t exponent, uint64_t* out, size_t n, void* stream);
/* Fp12::sparse_mul(ell_0, ell_vw, ell_vv): ell is [24][n] = (ell_0, ell_vw, ell_vv) as Fp2 each */
/* @shape f=u64[48*n] ell=u64[24*n] out=u64[48*n] */
int32_t sylow_hip_fp12_sparse_mul_batch(const uint64_t* f, const uint64_t* ell, uint64_t* out, size_t n, void* stream);

/* test hook for the carry-free 9 x 29-bit core used inside the final exponentiation (csrc/bn254_f29.hpp):
 * op 0: round trip; 1: a*b; 2: 2ab via the fused two-product pass; 3: 2a(b-a) through lazy add/sub + normalise */
/* @shape a=u64[*] b=u64[*]? out=u64[*] */
int32_t sylow_hip_f29_hook_batch(int32_t op, const uint64_t* a, const uint64_t* b, uint64_t* out, size_t n, void* stream);

/* ---- groups: src/groups/group.rs, g1.rs, g2.rs ----------------------------------------------- */
/* Mul<&Fp> for &G1Projective / &G2Projective (group.rs:639-667): out_i = k_i * P_i.
 * Points affine in (+ optional infinity flags), affine out + infinity flags (comparison is by
 * affine normalisation).  Scalars are Fp VALUES (k < p, not reduced mod r, N4). */
/* @shape p_xy=u64[8*n] p_inf=u8[n]? k=u64[4*n] out_xy=u64[8*n] out_inf=u8[n] */
int32_t sylow_hip_g1_scalar_mul_batch(const uint64_t* p_xy, const uint8_t* p_inf, const uint64_t* k,
                                      uint64_t* out_xy, uint8_t* out_inf, size_t n, void* stream);
/* @shape p_xy=u64[16*n] p_inf=u8[n]? k=u64[4*n] out_xy=u64[16*n] out_inf=u8[n] */
int32_t sylow_hip_g2_scalar_mul_batch(const uint64_t* p_xy, const uint8_t* p_inf, const uint64_t* k,
                                      uint64_t* out_xy, uint8_t* out_inf, size_t n, void* stream);
/* The same product for points of G2 proper (the r-torsion) -- every G2Projective the reference lets a caller build
 * (G2Projective::new checks membership, g2.rs:460-525; from_be_bytes, generator multiples, sums): the scalar is split
 * four ways along the endomorphism psi (g2.rs:140-152), ~1.8x faster, same affine result.  PRECONDITION: p_i in the r-torsion
 * (sylow_hip_g2_subgroup_check_batch / g2_from_be_bytes_batch establish it); for other points of the twist use
 * sylow_hip_g2_scalar_mul_batch, which is exact on the whole curve. */
/* @shape p_xy=u64[16*n] p_inf=u8[n]? k=u64[4*n] out_xy=u64[16*n] out_inf=u8[n] */
int32_t sylow_hip_g2_scalar_mul_subgroup_batch(const uint64_t* p_xy, const uint8_t* p_inf, const uint64_t* k,
                                               uint64_t* out_xy, uint8_t* out_inf, size_t n, void* stream);
/* G2Projective::generator() * k_i for a batch of scalars -- the public half of KeyPair::generate (lib.rs:131-137): a fixed-base
 * table of the generator (built once per device, 590 KB) turns the product into 32 additions, no doublings; same affine result
 * as sylow_hip_g2_scalar_mul_batch on the generator. */
/* @shape k=u64[4*n] out_xy=u64[16*n] out_inf=u8[n] */
int32_t sylow_hip_g2_generator_mul_batch(const uint64_t* k, uint64_t* out_xy, uint8_t* out_inf, size_t n, void* stream);
/* The same for G1Projective::generator() * k_i (GroupTrait::rand, test data): 295 KB table, 32 additions. */
/* @shape k=u64[4*n] out_xy=u64[8*n] out_inf=u8[n] */
int32_t sylow_hip_g1_generator_mul_batch(const uint64_t* k, uint64_t* out_xy, uint8_t* out_inf, size_t n, void* stream);
/* Add for &G1Projective (group.rs:528-599) on affine inputs, affine output */
/* @shape a_xy=u64[8*n] a_inf=u8[n]? b_xy=u64[8*n] b_inf=u8[n]? out_xy=u64[8*n] out_inf=u8[n] */
int32_t sylow_hip_g1_add_batch(const uint64_t* a_xy, const uint8_t* a_inf, const uint64_t* b_xy, const uint8_t* b_inf,
                               uint64_t* out_xy, uint8_t* out_inf, size_t n, void* stream);
/* Add for &G2Projective, GroupProjective::double for G1 / G2 (group.rs:528-599, 339-386), affine in / out */
/* @shape a_xy=u64[16*n] a_inf=u8[n]? b_xy=u64[16*n] b_inf=u8[n]? out_xy=u64[16*n] out_inf=u8[n] */
int32_t sylow_hip_g2_add_batch(const uint64_t* a_xy, const uint8_t* a_inf, const uint64_t* b_xy, const uint8_t* b_inf,
                               uint64_t* out_xy, uint8_t* out_inf, size_t n, void* stream);
/* Sub for &G1Projective / &G2Projective (group.rs:614-624: self + (-other)), affine in / out */
/* @shape a_xy=u64[8*n] a_inf=u8[n]? b_xy=u64[8*n] b_inf=u8[n]? out_xy=u64[8*n] out_inf=u8[n] */
int32_t sylow_hip_g1_sub_batch(const uint64_t* a_xy, const uint8_t* a_inf, const uint64_t* b_xy, const uint8_t* b_inf,
                               uint64_t* out_xy, uint8_t* out_inf, size_t n, void* stream);
/* @shape a_xy=u64[16*n] a_inf=u8[n]? b_xy=u64[16*n] b_inf=u8[n]? out_xy=u64[16*n] out_inf=u8[n] */
int32_t sylow_hip_g2_sub_batch(const uint64_t* a_xy, const uint8_t* a_inf, const uint64_t* b_xy, const uint8_t* b_inf,
                               uint64_t* out_xy, uint8_t* out_inf, size_t n, void* stream);
/* G1Projective::new([x, y, z]) (g1.rs:383-402) and G2Projective::new([x, y, z]) (g2.rs:460-525) on projective SoA input
 * ([12][n] / [24][n]): status OK / NOT_ON_CURVE / NOT_IN_SUBGROUP (G2 only); Z = 0 is accepted as the reference does.  Where the
 * reference panics (an off-curve G2 input reaches endomorphism(), g2.rs:151) the status is NOT_ON_CURVE. */
/* @shape p_xyz=u64[12*n] status=u8[n] */
int32_t sylow_hip_g1_projective_new_batch(const uint64_t* p_xyz, uint8_t* status, size_t n, void* stream);
/* @shape p_xyz=u64[24*n] status=u8[n] */
int32_t sylow_hip_g2_projective_new_batch(const uint64_t* p_xyz, uint8_t* status, size_t n, void* stream);
/* ConstantTimeEq / PartialEq for projective points (group.rs:426-447): eq[i] = 1 iff both are the identity, or neither is and
 * the cross-multiplied coordinates agree.  a, b projective SoA [12][n] / [24][n]. */
/* @shape a_xyz=u64[12*n] b_xyz=u64[12*n] eq=u8[n] */
int32_t sylow_hip_g1_ct_eq_batch(const uint64_t* a_xyz, const uint64_t* b_xyz, uint8_t* eq, size_t n, void* stream);
/* @shape a_xyz=u64[24*n] b_xyz=u64[24*n] eq=u8[n] */
int32_t sylow_hip_g2_ct_eq_batch(const uint64_t* a_xyz, const uint64_t* b_xyz, uint8_t* eq, size_t n, void* stream);
/* @shape a_xy=u64[8*n] a_inf=u8[n]? out_xy=u64[8*n] out_inf=u8[n] */
int32_t sylow_hip_g1_double_batch(const uint64_t* a_xy, const uint8_t* a_inf, uint64_t* out_xy, uint8_t* out_inf, size_t n, void* stream);
/* @shape a_xy=u64[16*n] a_inf=u8[n]? out_xy=u64[16*n] out_inf=u8[n] */
int32_t sylow_hip_g2_double_batch(const uint64_t* a_xy, const uint8_t* a_inf, uint64_t* out_xy, uint8_t* out_inf, size_t n, void* stream);
/* Weighted aggregation sum_i k_{j,i} * P_{j,i} (examples/threshold_signing.rs:124-143: partial signatures times
 * Lagrange coefficients), n_jobs independent sums of n_terms terms each.  p_xy [8][n_jobs*n_terms], k [4][n_jobs*n_terms]
 * (Fr / Fp values), term-major: element (job j, term i) is at index i*n_jobs + j.  out [8][n_jobs] affine + flags.
 * n_terms = 0 yields the identity, like G1Projective::default(). */
/* @shape p_xy=u64[8*n_jobs*n_terms] p_inf=u8[n_jobs*n_terms]? k=u64[4*n_jobs*n_terms] out_xy=u64[8*n_jobs] out_inf=u8[n_jobs] */
int32_t sylow_hip_g1_lincomb_batch(const uint64_t* p_xy, const uint8_t* p_inf, const uint64_t* k, uint64_t* out_xy, uint8_t* out_inf,
                                   size_t n_jobs, size_t n_terms, void* stream);
/* Mul<&Fr> for &Gt (groups/gt.rs:161-187): out_i = gt_i "times" k_i, i.e. gt_i^k_i in Fp12, by the reference's own
 * 256-step signed-digit square-and-multiply (negative digits multiply by the conjugate).  k: Fr values, [4][n]. */
/* @shape gt=u64[48*n] k=u64[4*n] out=u64[48*n] */
int32_t sylow_hip_gt_pow_batch(const uint64_t* gt, const uint64_t* k, uint64_t* out, size_t n, void* stream);
/* GroupAffine::from(&GroupProjective) (group.rs:475-495) */
/* @shape p_xyz=u64[12*n] out_xy=u64[8*n] out_inf=u8[n] */
int32_t sylow_hip_g1_normalize_batch(const uint64_t* p_xyz, uint64_t* out_xy, uint8_t* out_inf, size_t n, void* stream);
/* @shape p_xyz=u64[24*n] out_xy=u64[16*n] out_inf=u8[n] */
int32_t sylow_hip_g2_normalize_batch(const uint64_t* p_xyz, uint64_t* out_xy, uint8_t* out_inf, size_t n, void* stream);
/* sum_i P_i of a batch of G1 points as ONE point (the `+` fold over signatures / hashes of
 * examples/verify_multiple_messages_same_signer.rs:41-60, Add for G1Projective, group.rs:528-599): p_xy [8][n] affine + flags in,
 * out_xy [8][1] + out_inf [1] out; n = 0 gives the identity.  Serial per-lane accumulation in stages (g1.hip), complete formulas. */
/* @shape p_xy=u64[8*n] p_inf=u8[n]? out_xy=u64[8] out_inf=u8[1] */
int32_t sylow_hip_g1_sum_batch(const uint64_t* p_xy, const uint8_t* p_inf, size_t n, uint64_t* out_xy, uint8_t* out_inf, void* stream);
/* G1Affine::new (g1.rs:111-132): status[i] = OK when y^2 == x^3 + 3 (or the identity flag is set), NOT_ON_CURVE otherwise */
/* @shape p_xy=u64[8*n] p_inf=u8[n]? status=u8[n] */
int32_t sylow_hip_g1_on_curve_batch(const uint64_t* p_xy, const uint8_t* p_inf, uint8_t* status, size_t n, void* stream);
/* G2Affine::endomorphism (g2.rs:140-152): psi(x, y) = (xi^((p-1)/3) conj x, xi^((p-1)/2) conj y), identity -> identity.
 * status (may be NULL): NOT_ON_CURVE where the reference's on-curve re-check of the image would panic. */
/* @shape q_xy=u64[16*n] q_inf=u8[n]? out_xy=u64[16*n] out_inf=u8[n] status=u8[n] */
int32_t sylow_hip_g2_psi_batch(const uint64_t* q_xy, const uint8_t* q_inf, uint64_t* out_xy, uint8_t* out_inf, uint8_t* status, size_t n, void* stream);
/* G2Projective::new on affine input (g2.rs:460-525): status = OK / NOT_ON_CURVE / NOT_IN_SUBGROUP.
 * (The reference panics for off-curve input, g2.rs:151; this returns NOT_ON_CURVE instead.) */
/* @shape q_xy=u64[16*n] q_inf=u8[n]? status=u8[n] */
int32_t sylow_hip_g2_subgroup_check_batch(const uint64_t* q_xy, const uint8_t* q_inf, uint8_t* status, size_t n, void* stream);

/* ---- pairing: src/pairing.rs ------------------------------------------------------------------- */
/* G2Affine::precompute().miller_loop(&G1Affine) (pairing.rs:590-619, 676-708): raw Miller value,
 * strict replay of the reference's line formulas and digit schedule.  No infinity handling. */
/* @shape p_xy=u64[8*n] q_xy=u64[16*n] f_out=u64[48*n] */
int32_t sylow_hip_miller_loop_batch(const uint64_t* p_xy, const uint64_t* q_xy, uint64_t* f_out, size_t n, void* stream);
/* MillerLoopResult::final_exponentiation (pairing.rs:245-492) */
/* @shape f=u64[48*n] gt_out=u64[48*n] */
int32_t sylow_hip_final_exp_batch(const uint64_t* f, uint64_t* gt_out, size_t n, void* stream);
/* pairing(&G1, &G2) (pairing.rs:870-893): n independent Gt values; either input at infinity ->
 * Gt identity.  p_inf / q_inf may be NULL. */
/* @shape p_xy=u64[8*n] p_inf=u8[n]? q_xy=u64[16*n] q_inf=u8[n]? gt_out=u64[48*n] */
int32_t sylow_hip_pairing_batch(const uint64_t* p_xy, const uint8_t* p_inf, const uint64_t* q_xy, const uint8_t* q_inf,
                                uint64_t* gt_out, size_t n, void* stream);
/* glued_pairing (pairing.rs:970-1037), one product per job: job j multiplies the pairs
 * [pair_offsets[j], pair_offsets[j+1]) (uint64 device array of n_jobs+1 entries; n_pairs =
 * pair_offsets[n_jobs] is the SoA stride of p_xy / q_xy) with shared
 * squarings and ONE final exponentiation.  skip_infinity = 0 replays the reference (infinity flags
 * ignored: a G2 identity zeroes the product); skip_infinity = 1 drops pairs with an
 * identity on either side (EIP-197 semantics).  gt_out [48][n_jobs] may be NULL;
 * is_one [n_jobs] (may be NULL) receives product == Gt::identity(). */
/* @shape p_xy=u64[8*n_pairs]? p_inf=u8[n_pairs]? q_xy=u64[16*n_pairs]? q_inf=u8[n_pairs]? pair_offsets=u64[n_jobs+1] gt_out=u64[48*n_jobs]? is_one=u8[n_jobs]? */
int32_t sylow_hip_multi_pairing_batch(const uint64_t* p_xy, const uint8_t* p_inf, const uint64_t* q_xy, const uint8_t* q_inf,
                                      const uint64_t* pair_offsets, size_t n_jobs, size_t n_pairs, int32_t skip_infinity,
                                      uint64_t* gt_out, uint8_t* is_one, void* stream);
/* glued_miller_loop(&[G2PreComputed], &[G1Affine]) -> MillerLoopResult (pairing.rs:970-1022), one raw value per job (same job
 * layout as multi_pairing_batch, no final exponentiation, no identity handling -- exactly like the reference's loop).  The value
 * is the product of the per-pair Miller values, which is what the shared-squaring loop computes. */
/* @shape p_xy=u64[8*n_pairs]? q_xy=u64[16*n_pairs]? pair_offsets=u64[n_jobs+1] f_out=u64[48*n_jobs] */
int32_t sylow_hip_glued_miller_loop_batch(const uint64_t* p_xy, const uint64_t* q_xy, const uint64_t* pair_offsets, size_t n_jobs, size_t n_pairs,
                                          uint64_t* f_out, void* stream);
/* glued_pairing over the WHOLE batch as one product (pairing.rs:1029-1037 applied to n_pairs pairs; the batch-verification
 * shape of examples/verify_multiple_messages_same_signer.rs:41-60 and threshold_signing.rs:92-121): gt_out [48][1] =
 * final_exponentiation(prod_i miller(P_i, Q_i)), is_one[0] = (that == Gt::identity()).  The pairs are spread over the whole
 * GPU (chunks with shared squarings, a product tree, one final exponentiation); the value is the one the reference's
 * sequential glued loop yields.  skip_infinity as for multi_pairing_batch.  n_pairs = 0 gives the identity. */
/* @shape p_xy=u64[8*n_pairs]? p_inf=u8[n_pairs]? q_xy=u64[16*n_pairs]? q_inf=u8[n_pairs]? gt_out=u64[48]? is_one=u8[1]? */
int32_t sylow_hip_pairing_product_batch(const uint64_t* p_xy, const uint8_t* p_inf, const uint64_t* q_xy, const uint8_t* q_inf,
                                        size_t n_pairs, int32_t skip_infinity, uint64_t* gt_out, uint8_t* is_one, void* stream);

/* The same loops against line tables a host CACHED from sylow_hip_g2_precompute_batch (`G2PreComputed`, pairing.rs:556):
 * G2PreComputed::miller_loop(&G1Affine) (pairing.rs:590-619) and glued_miller_loop(&[G2PreComputed], &[G1Affine])
 * (pairing.rs:970-1022).  coeffs is the canonical SoA array [87*24][n_tables] exactly as g2_precompute_batch wrote it; pair i
 * uses table table_idx[i] (uint64 device array, every entry < n_tables -- NOT checked on the device: an out-of-range index
 * reads outside coeffs), or table i when table_idx is NULL (then n_tables must equal the number of
 * pairs) -- so one cached key serves any number of G1 points.  Raw MillerLoopResult out, no identity handling (as the
 * reference).  The glued form takes the job layout of multi_pairing_batch; an empty job yields 1. */
/* @shape coeffs=u64[87*24*n_tables] table_idx=u64[n]? p_xy=u64[8*n] f_out=u64[48*n] */
int32_t sylow_hip_miller_loop_precomputed_batch(const uint64_t* coeffs, size_t n_tables, const uint64_t* table_idx, const uint64_t* p_xy,
                                                uint64_t* f_out, size_t n, void* stream);
/* @shape coeffs=u64[87*24*n_tables]? table_idx=u64[n_pairs]? p_xy=u64[8*n_pairs]? pair_offsets=u64[n_jobs+1] f_out=u64[48*n_jobs] */
int32_t sylow_hip_glued_miller_loop_precomputed_batch(const uint64_t* coeffs, size_t n_tables, const uint64_t* table_idx, const uint64_t* p_xy,
                                                      const uint64_t* pair_offsets, size_t n_jobs, size_t n_pairs, uint64_t* f_out, void* stream);
/* The two halves of pairing_product_batch, for hosts that split one product over several GPUs:
 * partial: f_out [48][1] = c * prod_i miller(P_i, Q_i) of this shard (no final exponentiation; n_pairs = 0 gives 1) with some c in Fp* that
 *          depends on the route taken -- an intermediate for _final_ only, where c disappears (c^(p^6 - 1) = 1): the Miller loops of a value
 *          that ends in a final exponentiation may run on isomorphic curves.  The reference's raw MillerLoopResult
 *          itself comes from sylow_hip_miller_loop_batch / sylow_hip_glued_miller_loop_batch;
 * final:   gt_out [48][1] = final_exponentiation(prod_{j<k} parts_j), parts SoA [48][k]; is_one[0] = (== Gt::identity()).
 * glued_pairing over all shards == fp12_product_final_exp over the shards' partials (Fp12 products commute). */
/* @shape p_xy=u64[8*n_pairs]? p_inf=u8[n_pairs]? q_xy=u64[16*n_pairs]? q_inf=u8[n_pairs]? f_out=u64[48] */
int32_t sylow_hip_pairing_product_partial_batch(const uint64_t* p_xy, const uint8_t* p_inf, const uint64_t* q_xy, const uint8_t* q_inf,
                                                size_t n_pairs, int32_t skip_infinity, uint64_t* f_out, void* stream);
/* @shape parts=u64[48*k] gt_out=u64[48]? is_one=u8[1]? */
int32_t sylow_hip_fp12_product_final_exp(const uint64_t* parts, size_t k, uint64_t* gt_out, uint8_t* is_one, void* stream);

/* ---- hash-to-curve and BLS: src/hasher.rs, src/svdw.rs, src/groups/g1.rs:307-331, src/lib.rs --- */
/* Expander::hash_to_field(msg, 2, 48) with XMDExpander<Keccak256>(dst, 128) (hasher.rs:84-128, 157-250): out_u [8][n] = (u0, u1),
 * each the 48-byte big-endian slice of expand_message_xmd(msg, DST', 96) reduced mod p.  dst_host NULL = the library DST. */
/* @shape msgs=u8[*] msg_offsets=u64[n+1] dst_host=u8[dst_len]? out_u=u64[8*n] */
int32_t sylow_hip_hash_to_field_batch(const uint8_t* msgs, const uint64_t* msg_offsets, const uint8_t* dst_host, size_t dst_len,
                                      uint64_t* out_u, size_t n, void* stream);
/* G1Projective::hash_to_curve with XMDExpander<Keccak256>(dst, 128), COUNT=2, L=48.
 * msgs: concatenated message bytes; msg_offsets: n+1 uint64 byte offsets.  dst/dst_len: HOST
 * pointer to the domain separation tag (NULL -> sylow's DST, lib.rs:90). */
/* @shape msgs=u8[*] msg_offsets=u64[n+1] dst_host=u8[dst_len]? out_xy=u64[8*n] out_inf=u8[n] */
int32_t sylow_hip_hash_to_g1_batch(const uint8_t* msgs, const uint64_t* msg_offsets, const uint8_t* dst_host, size_t dst_len,
                                   uint64_t* out_xy, uint8_t* out_inf, size_t n, void* stream);
/* SvdW::unchecked_map_to_point (svdw.rs:180-262, RFC 9380 6.6.1 straight line, Z = 1) by itself: u [4][n] -> (x, y) [8][n] on the
 * curve; status (may be NULL) = CANNOT_HASH where the reference returns MapError. */
/* @shape u=u64[4*n] out_xy=u64[8*n] status=u8[n] */
int32_t sylow_hip_svdw_map_batch(const uint64_t* u, uint64_t* out_xy, uint8_t* status, size_t n, void* stream);
/* Fp::compute_naf (fp.rs:653-662) on the raw 256-bit words k [4][n]: out_np / out_nm [4][n] = the masks of the +1 / -1 digits
 * (digit_i = np_i - nm_i; x + (x >> 1) is taken modulo 2^256 exactly as the reference's 256-bit arithmetic does). */
/* @shape k=u64[4*n] out_np=u64[4*n] out_nm=u64[4*n] */
int32_t sylow_hip_fp_compute_naf_batch(const uint64_t* k, uint64_t* out_np, uint64_t* out_nm, size_t n, void* stream);
/* sign(&Fp, &[u8]) (lib.rs:179-187): sig_i = sk_i * H(msg_i), affine out */
/* @shape sk=u64[4*n] msgs=u8[*] msg_offsets=u64[n+1] sig_xy=u64[8*n] sig_inf=u8[n] */
int32_t sylow_hip_bls_sign_batch(const uint64_t* sk, const uint8_t* msgs, const uint64_t* msg_offsets,
                                 uint64_t* sig_xy, uint8_t* sig_inf, size_t n, void* stream);
/* verify(&G2Projective, &[u8], &G1Projective) (lib.rs:223-236): ok_i = [ e(sig_i, G2gen) == e(H(msg_i), pk_i) ]; identity inputs
 * as pairing() treats them (that pairing is Gt::identity()).  Evaluated as e(sig_i, G2gen) * e(-H(msg_i), pk_i) == 1 with one
 * shared-squaring 2-pair Miller loop and ONE final exponentiation per element (the shape of
 * examples/verify_multiple_messages_same_signer.rs:41-60, threshold_signing.rs:92-121): FE(a) == FE(b) <=> FE(a conj(b)) == 1 and
 * conj(miller(H, pk)) = miller(-H, pk) exactly, so the boolean is the reference's for EVERY input.  _fused_ is the same
 * kernel under its round-1 name. */
/* @shape pk_xy=u64[16*n] pk_inf=u8[n]? msgs=u8[*] msg_offsets=u64[n+1] sig_xy=u64[8*n] sig_inf=u8[n]? ok=u8[n] */
int32_t sylow_hip_bls_verify_batch(const uint64_t* pk_xy, const uint8_t* pk_inf, const uint8_t* msgs, const uint64_t* msg_offsets,
                                   const uint64_t* sig_xy, const uint8_t* sig_inf, uint8_t* ok, size_t n, void* stream);
/* @shape pk_xy=u64[16*n] pk_inf=u8[n]? msgs=u8[*] msg_offsets=u64[n+1] sig_xy=u64[8*n] sig_inf=u8[n]? ok=u8[n] */
int32_t sylow_hip_bls_verify_fused_batch(const uint64_t* pk_xy, const uint8_t* pk_inf, const uint8_t* msgs, const uint64_t* msg_offsets,
                                         const uint64_t* sig_xy, const uint8_t* sig_inf, uint8_t* ok, size_t n, void* stream);
/* The same boolean evaluated literally as lib.rs:223-236 writes it: two Miller loops, two final exponentiations, compare
 * (~1.5x the time; kept as the second implementation the first is tested against, and to price the reference's shape). */
/* @shape pk_xy=u64[16*n] pk_inf=u8[n]? msgs=u8[*] msg_offsets=u64[n+1] sig_xy=u64[8*n] sig_inf=u8[n]? ok=u8[n] */
int32_t sylow_hip_bls_verify_two_pairings_batch(const uint64_t* pk_xy, const uint8_t* pk_inf, const uint8_t* msgs, const uint64_t* msg_offsets,
                                                const uint64_t* sig_xy, const uint8_t* sig_inf, uint8_t* ok, size_t n, void* stream);
/* ---- wire formats: G1Affine/G2Affine::{to,from}_be_bytes (g1.rs:151-280, g2.rs:319-433) -------------- */
/* G1: 64 bytes x | y big-endian; G2: 128 bytes x.c1 | x.c0 | y.c1 | y.c0; bit 7 of byte 0 is the infinity flag and
 * the identity is written as (0, 1) + flag.  from_be_bytes masks the flag, then: a coordinate >= p, or a set flag
 * with coordinates other than (0, 1) -> DECODE_ERROR (the reference's CtOption is none); off the curve ->
 * NOT_ON_CURVE; G2 outside the r-torsion -> NOT_IN_SUBGROUP.  Failed elements decode to the identity. */
/* @shape p_xy=u64[8*n] p_inf=u8[n]? out=u8[64*n] */
int32_t sylow_hip_g1_to_be_bytes_batch(const uint64_t* p_xy, const uint8_t* p_inf, uint8_t* out /*[n][64]*/, size_t n, void* stream);
/* @shape in=u8[64*n] out_xy=u64[8*n] out_inf=u8[n] status=u8[n] */
int32_t sylow_hip_g1_from_be_bytes_batch(const uint8_t* in /*[n][64]*/, uint64_t* out_xy, uint8_t* out_inf, uint8_t* status, size_t n, void* stream);
/* @shape p_xy=u64[16*n] p_inf=u8[n]? out=u8[128*n] */
int32_t sylow_hip_g2_to_be_bytes_batch(const uint64_t* p_xy, const uint8_t* p_inf, uint8_t* out /*[n][128]*/, size_t n, void* stream);
/* @shape in=u8[128*n] out_xy=u64[16*n] out_inf=u8[n] status=u8[n] */
int32_t sylow_hip_g2_from_be_bytes_batch(const uint8_t* in /*[n][128]*/, uint64_t* out_xy, uint8_t* out_inf, uint8_t* status, size_t n, void* stream);

/* ---- EVM alt_bn128 precompile shapes: examples/reth_bn128.rs:99-217 (EIP-196 / EIP-197) --------- */
/* Byte-level batches.  Field elements are 32-byte big-endian and must be < p (else DECODE_ERROR =
 * Bn128FieldPointNotAMember); (0,0) is the identity; G1 points must be on the curve, G2 points on the twist
 * and in the r-torsion (else NOT_ON_CURVE / NOT_IN_SUBGROUP = Bn128AffineGFailedToCreate).
 * ecadd: in [n][128] = two points, out [n][64] (identity -> 64 zero bytes, to_be_bytes_scrubbed g1.rs:182-192).
 * ecmul: in [n][96] = point | 32-byte scalar (any 256-bit value, reduced mod r), out [n][64].
 * ecpairing: `in` = n_pairs 192-byte elements (G1 x|y, G2 x.c1|x.c0|y.c1|y.c0), job j owns elements
 * [pair_offsets[j], pair_offsets[j+1]); result[j] = 1 iff the product of pairings is one (empty job -> 1).
 * Identity pairs are skipped as EIP-197 requires (the reference adapter inherits glued_pairing's Q = identity
 * defect and answers false there).  Padding, length % 192 and gas rules are host-side (sylow_amd/evm.py). */
/* @shape in=u8[128*n] out=u8[64*n] status=u8[n] */
int32_t sylow_hip_evm_ecadd_batch(const uint8_t* in, uint8_t* out, uint8_t* status, size_t n, void* stream);
/* @shape in=u8[96*n] out=u8[64*n] status=u8[n] */
int32_t sylow_hip_evm_ecmul_batch(const uint8_t* in, uint8_t* out, uint8_t* status, size_t n, void* stream);
/* @shape in=u8[192*n_pairs]? pair_offsets=u64[n_jobs+1] result=u8[n_jobs] status=u8[n_jobs] */
int32_t sylow_hip_evm_ecpairing_batch(const uint8_t* in, const uint64_t* pair_offsets, size_t n_jobs, size_t n_pairs,
                                      uint8_t* result, uint8_t* status, void* stream);
/* Same-signer batch (examples/verify_multiple_messages_same_signer.rs:41-60): ONE public key (pk_xy is a
 * 1-element SoA array, pk_inf one byte or NULL) against n (message, signature) pairs.  The key's G2PreComputed
 * line table is built once per call and both pairs of every element read wave-uniform tables, so the Miller
 * loops contain no G2 arithmetic. */
/* @shape pk_xy=u64[16] pk_inf=u8[1]? msgs=u8[*] msg_offsets=u64[n+1] sig_xy=u64[8*n] sig_inf=u8[n]? ok=u8[n] */
int32_t sylow_hip_bls_verify_same_signer_batch(const uint64_t* pk_xy, const uint8_t* pk_inf, const uint8_t* msgs, const uint64_t* msg_offsets,
                                               const uint64_t* sig_xy, const uint8_t* sig_inf, uint8_t* ok, size_t n, void* stream);
/* The same check against a key table the host keeps across calls ("G2PreComputed cached per pk"): sylow_hip_g2_line_table
 * writes the line table of element idx of an SoA G2 array (n = its stride) into `table`, a device buffer of
 * sylow_hip_g2_line_table_words() int32 words (opaque, device-internal digit layout); bls_verify_line_table_batch then runs
 * the same-signer check with no G2 arithmetic and nothing rebuilt per call.  pk_inf: one device byte or NULL. */
int32_t sylow_hip_g2_line_table_words(void);
/* @shape q_xy=u64[16*n]? table=i32[*] */
int32_t sylow_hip_g2_line_table(const uint64_t* q_xy, size_t n, size_t idx, int32_t* table, void* stream);
/* @shape pk_table=i32[*] pk_inf=u8[1]? msgs=u8[*] msg_offsets=u64[n+1] sig_xy=u64[8*n] sig_inf=u8[n]? ok=u8[n] */
int32_t sylow_hip_bls_verify_line_table_batch(const int32_t* pk_table, const uint8_t* pk_inf, const uint8_t* msgs, const uint64_t* msg_offsets,
                                              const uint64_t* sig_xy, const uint8_t* sig_inf, uint8_t* ok, size_t n, void* stream);
/* G2Affine::precompute (pairing.rs:676-708): the 87 line-coefficient triples [Ell; 87] of each point, canonical
 * words, SoA [87*24][n] (triple t = words 24t..24t+23 = ell.0, ell.1, ell.2 as Fp2).  Strict replay. */
/* @shape q_xy=u64[16*n] coeffs=u64[87*24*n] */
int32_t sylow_hip_g2_precompute_batch(const uint64_t* q_xy, uint64_t* coeffs, size_t n, void* stream);
/* AND of a flag array -> one int32 on the device (1 = all set); the multi-GPU aggregate then
 * MIN-reduces that word over ranks (RCCL has no bit-AND; min over {0,1} is AND). */
/* @shape flags=u8[n] out_dev=i32[1] */
int32_t sylow_hip_flags_all(const uint8_t* flags, size_t n, int32_t* out_dev, void* stream);


/* ---- multi-GPU aggregates (one process per GPU): the only exchange steps of the path --------------------------------------
 * `comm` is the caller's RCCL communicator (ncclComm_t) passed as void*; NULL means a single rank.  RCCL is bound at run time
 * (dlopen of librccl.so.1), so single-GPU hosts never load it.  Both calls are asynchronous on `stream` and must be issued by
 * every rank of the communicator, like any collective.
 * all_valid: out_dev[0] (device int32) = 1 iff every flag on EVERY rank is set -- flags_all + a 4-byte MIN all-reduce
 *            (verify(...) over a sharded batch -> one boolean, lib.rs:223-236).
 * pairing_product_all: glued_pairing (pairing.rs:1029-1037) over the union of all ranks' pairs: every rank computes its
 *            partial Miller product, the 384-byte partials are all-gathered and each rank finishes product + final
 *            exponentiation, so every rank ends with the same gt_out [48][1] / is_one[0]. */
/* @shape flags=u8[n] comm=void[*]? out_dev=i32[1] */
int32_t sylow_hip_all_valid(const uint8_t* flags, size_t n, void* comm, int32_t* out_dev, void* stream);
/* @shape p_xy=u64[8*n_pairs]? p_inf=u8[n_pairs]? q_xy=u64[16*n_pairs]? q_inf=u8[n_pairs]? comm=void[*]? gt_out=u64[48]? is_one=u8[1]? */
int32_t sylow_hip_pairing_product_all(const uint64_t* p_xy, const uint8_t* p_inf, const uint64_t* q_xy, const uint8_t* q_inf, size_t n_pairs,
                                      int32_t skip_infinity, void* comm, uint64_t* gt_out, uint8_t* is_one, void* stream);

/* Aggregate verification -- "are ALL n signatures valid?" as ONE Gt comparison, the batch shape of
 * examples/verify_multiple_messages_same_signer.rs:41-60 and threshold_signing.rs:92-121, where the reference glues the 2n pairs
 * (sig_i, G2gen), (-H(msg_i), pk_i) into one product:  gt_out [48][1] = that product after the final exponentiation,
 * is_one[0] = (gt_out == Gt::identity()).  Evaluated through bilinearity: prod_i e(sig_i, G2gen) = e(sum_i sig_i, G2gen), so the
 * G2gen half costs n G1 additions and ONE Miller loop; with one key for the whole batch (n_pk = 1) the other half collapses the
 * same way, prod_i e(-H(msg_i), pk) = e(-sum_i H(msg_i), pk), and the check is n hashes + two sums + a two-pair product.  The Gt
 * value is the same group element as the reference's 2n-pair product, hence the same words.  n_pk = n: pk_xy [16][n] one key per
 * message; n_pk = 1: pk_xy [16][1].  Identity signatures / keys contribute 1 (pairing() semantics).  n = 0 gives the identity.
 * (As in the reference's example there are no random weights: it answers "is the PRODUCT the identity" -- the product sees the
 * signatures only through their sum, so signatures permuted among the messages still pass; per-element flags: bls_verify_batch.)
 *   _partial_: f_out [48][1] = this shard's Miller product up to a factor in Fp* (see sylow_hip_pairing_product_partial_batch; for hosts
 *              that combine shards themselves, with sylow_hip_fp12_product_final_exp);
 *   _verify_:  the whole check; comm = the host's ncclComm_t for a batch sharded over the GPUs of a node (every rank passes its
 *              shard and receives the same answer; 384 bytes per rank are all-gathered), NULL = this process alone. */
/* @shape pk_xy=u64[16*n_pk] pk_inf=u8[n_pk]? msgs=u8[*] msg_offsets=u64[n+1] sig_xy=u64[8*n] sig_inf=u8[n]? f_out=u64[48] */
int32_t sylow_hip_bls_aggregate_partial_batch(const uint64_t* pk_xy, const uint8_t* pk_inf, size_t n_pk, const uint8_t* msgs, const uint64_t* msg_offsets,
                                              const uint64_t* sig_xy, const uint8_t* sig_inf, size_t n, uint64_t* f_out, void* stream);
/* @shape pk_xy=u64[16*n_pk] pk_inf=u8[n_pk]? msgs=u8[*] msg_offsets=u64[n+1] sig_xy=u64[8*n] sig_inf=u8[n]? comm=void[*]? gt_out=u64[48]? is_one=u8[1]? */
int32_t sylow_hip_bls_aggregate_verify_batch(const uint64_t* pk_xy, const uint8_t* pk_inf, size_t n_pk, const uint8_t* msgs, const uint64_t* msg_offsets,
                                             const uint64_t* sig_xy, const uint8_t* sig_inf, size_t n, void* comm, uint64_t* gt_out, uint8_t* is_one, void* stream);
/* The SOUND one-boolean form ("random-linear-combination multi-pairing"): the small-exponent batch test
 *   prod_i [ e(sig_i, G2gen) e(-H(msg_i), pk_i) ]^(w_i) == identity,   evaluated as e(sum_i w_i sig_i, G2gen) prod_i e(-w_i H(msg_i), pk_i).
 * weights [4][n]: w_i as Fp values (the caller draws them AFTER the signatures are fixed, e.g. 64 or 128 random bits each; 0 removes an
 * element from the test).  If every signature is valid the result is the identity; if any is not, the test passes with probability at most
 * 2^-(bits of the weights) over the caller's randomness (keys in G2 proper, as G2Projective::new guarantees).  No counterpart exists upstream --
 * the reference's examples multiply unweighted (the two entry points above); the Gt value equals the reference's glued_pairing over the 2n
 * pairs (w_i sig_i, G2gen), (-w_i H(msg_i), pk_i), which is how it is tested.  Shapes, n_pk, comm and the _partial_ form as above. */
/* @shape pk_xy=u64[16*n_pk] pk_inf=u8[n_pk]? msgs=u8[*] msg_offsets=u64[n+1] sig_xy=u64[8*n] sig_inf=u8[n]? weights=u64[4*n] f_out=u64[48] */
int32_t sylow_hip_bls_weighted_partial_batch(const uint64_t* pk_xy, const uint8_t* pk_inf, size_t n_pk, const uint8_t* msgs, const uint64_t* msg_offsets,
                                             const uint64_t* sig_xy, const uint8_t* sig_inf, const uint64_t* weights, size_t n, uint64_t* f_out, void* stream);
/* @shape pk_xy=u64[16*n_pk] pk_inf=u8[n_pk]? msgs=u8[*] msg_offsets=u64[n+1] sig_xy=u64[8*n] sig_inf=u8[n]? weights=u64[4*n] comm=void[*]? gt_out=u64[48]? is_one=u8[1]? */
int32_t sylow_hip_bls_batch_verify_weighted(const uint64_t* pk_xy, const uint8_t* pk_inf, size_t n_pk, const uint8_t* msgs, const uint64_t* msg_offsets,
                                            const uint64_t* sig_xy, const uint8_t* sig_inf, const uint64_t* weights, size_t n, void* comm,
                                            uint64_t* gt_out, uint8_t* is_one, void* stream);

/* ---- test hooks (stable enough for the repo's own tests; not part of the drop-in surface) ------------------------------------
 * Granger-Scott cyclotomic square (pairing.rs:309-350) and the raw Fp12 selector: 0..7 one-element-per-lane tower ops (tower.hip), 8 / 9 product /
 * cyclotomic square on the carry-free core, 10 / 11 exp_by_neg_z (carry-free / saturated), 16..29 the lane-pair Fp12 layer. */
/* @shape a=u64[48*n] out=u64[48*n] */
int32_t sylow_hip_fp12_cyclotomic_sqr_batch(const uint64_t* a, uint64_t* out, size_t n, void* stream);
/* @shape a=u64[48*n] b=u64[*]? out=u64[48*n] */
int32_t sylow_hip_fp12_hook_batch(int32_t op, const uint64_t* a, const uint64_t* b, uint64_t* out, size_t n, void* stream);

/* ---- value-typed forms: HOST arrays in, HOST results out (pipeline.hip) ------------------------------------------------------
 * The reference's API takes and returns values (pairing(&G1Projective, &G2Projective) -> Gt, pairing.rs:870-893;
 * verify(&G2Projective, &[u8], &G1Projective) -> bool, lib.rs:223-236): a host that switches holds arrays of structs in HOST memory.
 * These two calls are the batched forms on such arrays -- EVERY pointer is a HOST pointer, element-major ("array of structs", what a
 * Rust Vec<[u64; W]> is): p_aos [n][8] (x, y), q_aos / pk_aos [n][16], sig_aos [n][8], gt_aos [n][48], flags [n] (NULL = none), msgs +
 * msg_offsets [n + 1] as in sylow_hip_bls_verify_batch, ok [n].  The batch is cut into chunks -- `chunk` elements (0 = 2^16: one resident
 * set of lane pairs) first and last, up to four times that in between -- that alternate between two internal streams, each chunk H2D -> AoS->SoA -> the same kernels as the device-pointer
 * entry points -> D2H, issued so that the copy engines move chunk k - 1 out and chunk k + 1 in while chunk k computes.  Synchronous:
 * the results are in host memory when the call returns.  Bit-identical to upload + sylow_hip_pairing_batch /
 * sylow_hip_bls_verify_batch + download.  Pageable memory works; pinned memory (sylow_hip_host_malloc) makes every copy asynchronous. */
/* @shape p_aos=u64[8*n] p_inf=u8[n]? q_aos=u64[16*n] q_inf=u8[n]? gt_aos=u64[48*n] */
int32_t sylow_hip_pairing_host(const uint64_t* p_aos, const uint8_t* p_inf, const uint64_t* q_aos, const uint8_t* q_inf,
                               uint64_t* gt_aos, size_t n, size_t chunk);
/* @shape pk_aos=u64[16*n] pk_inf=u8[n]? msgs=u8[*]? msg_offsets=u64[n+1] sig_aos=u64[8*n] sig_inf=u8[n]? ok=u8[n] */
int32_t sylow_hip_bls_verify_host(const uint64_t* pk_aos, const uint8_t* pk_inf, const uint8_t* msgs, const uint64_t* msg_offsets,
                                  const uint64_t* sig_aos, const uint8_t* sig_inf, uint8_t* ok, size_t n, size_t chunk);
/* The same two pipelines fed with the reference's WIRE format, which is how a Rust host gets points across without relying on sylow's
 * (non-repr(C)) memory layout: p_be / sig_be [n][64] = G1Affine::to_be_bytes (g1.rs:151-180), q_be / pk_be [n][128] =
 * G2Affine::to_be_bytes (g2.rs:319-359).  Decoding and validation run on the device inside the pipeline (from_be_bytes + curve check;
 * G2 also the r-torsion check of G2Projective::new, g2.rs:460-525); status_* [n] (HOST) receive SYLOW_HIP_ST_* per element, and an
 * element that failed enters the computation as the identity: its Gt is one, and its `ok` is forced to ZERO -- the reference returns Err
 * from from_be_bytes / G2Projective::new and never reaches verify, and identity inputs can satisfy the pairing equation (a rejected key
 * with an all-zero signature), so a caller that reads only `ok` must not see 1 for a rejected blob.  msg_offsets must be non-decreasing
 * over the whole batch (checked on the host: SYLOW_HIP_E_ARG otherwise). */
/* @shape p_be=u8[64*n] q_be=u8[128*n] gt_aos=u64[48*n] status_p=u8[n] status_q=u8[n] */
int32_t sylow_hip_pairing_host_bytes(const uint8_t* p_be, const uint8_t* q_be, uint64_t* gt_aos, uint8_t* status_p, uint8_t* status_q,
                                     size_t n, size_t chunk);
/* @shape pk_be=u8[128*n] msgs=u8[*]? msg_offsets=u64[n+1] sig_be=u8[64*n] ok=u8[n] status_pk=u8[n] status_sig=u8[n] */
int32_t sylow_hip_bls_verify_host_bytes(const uint8_t* pk_be, const uint8_t* msgs, const uint64_t* msg_offsets, const uint8_t* sig_be,
                                        uint8_t* ok, uint8_t* status_pk, uint8_t* status_sig, size_t n, size_t chunk);
/* page-locked host memory for the staging side of the calls above (hipHostMalloc / hipHostFree) */
int32_t sylow_hip_host_malloc(void** hptr, size_t bytes);
int32_t sylow_hip_host_free(void* hptr);

#ifdef __cplusplus
}
#endif
#endif /* SYLOW_HIP_H */
