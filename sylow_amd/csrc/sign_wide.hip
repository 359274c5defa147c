// sign_wide.hip -- BLS signing (lib.rs:179-187: sk * H(m)) for SINGLE calls and small batches: EIGHT lanes per signature.
//
// One signature on one lane (sign.hip) is a chain of ~2 900 dependent field multiplications: 2.1 ms whatever the batch size, because a lone
// lane issues a multiply-add every ~8 cycles.  What is independent inside one signature is spread over the 8 lanes of a group here:
//   * the two SvdW maps of hash_to_curve (g1.rs:307-331) run side by side, one per QUAD (4 lanes); inside a map the three square-root
//     candidates run on three lanes at once and replace the Jacobi symbols (is_square(gx1), is_square(gx2): svdw.rs:218-232 -- read off
//     as "candidate^2 == gx"); the maps' inversions stay shared (Montgomery's trick);
//   * the scalar multiplication is GLV-split (k = k1 + k2 lambda, bn254_pairing.hpp) with k1 * (+-P) on quad 0 and k2 * (+-phi P) on quad 1,
//     each with its own accumulator and window table (LDS), added once at the end;
//   * inside a quad every point is REPLICATED on the four lanes and each lane computes one of the independent products of a formula level:
//     a doubling (RCB'15 alg. 9, group.rs:339-386) is two levels of four products, a complete addition (alg. 7, group.rs:528-599) four
//     levels (6 + 6 products on 4 lanes); results travel by DPP quad_perm broadcasts.  128 doublings + 33 additions per half become
//     ~390 product latencies instead of ~2 050.
// Same formulas, same operand classes as proj_double_lazy / proj_add_lazy on the carry-free core; the affine result is the same field
// elements (tests/test_gpu_hash_bls.py, test_gpu_hash_chain.py run through this route at their batch sizes; tests/test_gpu_routes.py forces
// the one-lane kernel over the same files).  A unit of its own: sign.hip's kernels are compiled for three wavefronts per SIMD, this one is
// latency-bound at one and takes the default budget.
#include "host.hpp"

namespace wsign {
constexpr int GROUP = 8;                     // lanes per signature
constexpr int WBLOCK = 64;                   // one wavefront per block: 8 signatures
constexpr int EPB = WBLOCK / GROUP;

template <int S> BN_DEV int qbi(int v) { return __builtin_amdgcn_mov_dpp(v, S * 0x55, 0xF, 0xF, true); }   // lane S of this lane's quad
template <int S> BN_DEV F29 qb(const F29& a) {
  F29 r;
#pragma unroll
  for (int i = 0; i < 9; ++i) r.v[i] = qbi<S>(a.v[i]);
  return r;
}
// Operand of lane j: three conditional moves per limb.  Written as a CHAIN of two-way selects whose intermediate values the compiler
// cannot see through (inline v_cndmask_b32 on ballot masks) -- left alone it lowers the nested form j == 0 ? a0 : j == 1 ? a1 : ... as a switch with divergent control
// flow (53 exec-mask regions and ~570 instructions per doubling where 108 conditional moves do), and the plain chain as a table in the
// stack frame read back at a lane-dependent address (127 scratch accesses per addition).
BN_DEV i32 qcsel(i32 a, i32 b, u64 m) {          // lanes of mask m take b, the others a: one v_cndmask_b32, whatever the optimizer thinks of it
  i32 r;
  asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "s"(m));
  return r;
}
BN_DEV F29 qcsel9(const F29& a, const F29& b, u64 m) {
  F29 r;
#pragma unroll
  for (int i = 0; i < 9; ++i) r.v[i] = qcsel(a.v[i], b.v[i], m);
  return r;
}
// Call sites pass the same object for equal candidates (the address comparisons fold at compile time): one or two selects then
BN_DEV F29 qsel(int j, const F29& a0, const F29& a1, const F29& a2, const F29& a3) {
  const bool e01 = &a0 == &a1, e23 = &a2 == &a3, e02 = &a0 == &a2, e13 = &a1 == &a3, e12 = &a1 == &a2, e03 = &a0 == &a3;
  if (e02 && e13) return qcsel9(a0, a1, __builtin_amdgcn_ballot_w64((j & 1) != 0));                     // a b a b
  if (e01 && e23) return qcsel9(a2, a0, __builtin_amdgcn_ballot_w64(j < 2));                            // a a b b
  if (e12 && e03) return qcsel9(a0, a1, __builtin_amdgcn_ballot_w64(j == 1 || j == 2));                 // a b b a
  if (e01) return qcsel9(qcsel9(a3, a2, __builtin_amdgcn_ballot_w64(j == 2)), a0, __builtin_amdgcn_ballot_w64(j < 2));   // a a b c
  if (e23) return qcsel9(qcsel9(a2, a1, __builtin_amdgcn_ballot_w64(j == 1)), a0, __builtin_amdgcn_ballot_w64(j == 0));  // a b c c
  const u64 m0 = __builtin_amdgcn_ballot_w64(j == 0), m1 = __builtin_amdgcn_ballot_w64(j == 1), m2 = __builtin_amdgcn_ballot_w64(j == 2);
  return qcsel9(qcsel9(qcsel9(a3, a2, m2), a1, m1), a0, m0);
}
BN_DEV Fp xq_fp(const Fp& a) {               // the other quad's value
  Fp r;
#pragma unroll
  for (int i = 0; i < 8; ++i) r.v[i] = (u32)__shfl_xor((int)a.v[i], 4);
  return r;
}
BN_DEV F29 xq_f29(const F29& a) {
  F29 r;
#pragma unroll
  for (int i = 0; i < 9; ++i) r.v[i] = __shfl_xor(a.v[i], 4);
  return r;
}

// proj_double_lazy<OpsF29> (bn254_pairing.hpp) with its eight products on the four lanes of a quad, two levels.  p replicated, N-class.
// The products are INLINED (OpsF29I): a lone wavefront pays for every instruction, the 27 argument / result moves and the call included.
// No identity select: the accumulators and table entries here only ever hold the CANONICAL identity (0 : 1 : 0), which the formulas map to itself.
BN_DEV G1W qdouble(const G1W& p, int j) {
  const F29 m1 = OpsF29I::mul(qsel(j, p.y, p.y, p.z, p.x), qsel(j, p.y, p.z, p.z, p.y));
  const F29 t0 = qb<0>(m1), t1 = qb<1>(m1), zz = qb<2>(m1), xy = qb<3>(m1);      // y^2, y z, z^2, x y
  const F29 z8 = f29_norm_x8(t0);
  const F29 t2 = OpsF29::mul_b3(zz);
  const F29 s = f29_norm(f29_add(t0, t2));
  const F29 d = f29_norm_sub3(t0, t2);
  const F29 m2 = OpsF29I::mul(qsel(j, t2, t1, d, d), qsel(j, z8, z8, s, xy));
  const F29 x3a = qb<0>(m2), z3 = qb<1>(m2), y3a = qb<2>(m2), x3b = qb<3>(m2);
  return G1W{f29_norm(f29_add(x3b, x3b)), f29_norm(f29_add(x3a, y3a)), z3};
}
// proj_add_lazy<OpsF29> with its twelve products on the four lanes of a quad, four levels (4 + 2 + 4 + 2).  p, q replicated, N-class.
BN_DEV G1W qadd(const G1W& p, const G1W& q, int j) {
  const F29 pxy = f29_sub(p.x, p.y), qxy = f29_sub(q.x, q.y), pyz = f29_sub(p.y, p.z), qyz = f29_sub(q.y, q.z),
            pxz = f29_sub(p.x, p.z), qxz = f29_sub(q.x, q.z);
  const F29 m1 = OpsF29I::mul(qsel(j, p.x, p.y, p.z, pxy), qsel(j, q.x, q.y, q.z, qxy));
  F29 t0 = qb<0>(m1), t1 = qb<1>(m1), t2 = qb<2>(m1);
  const F29 c3 = qb<3>(m1);
  const F29 m2 = OpsF29I::mul(qsel(j, pyz, pxz, pyz, pxz), qsel(j, qyz, qxz, qyz, qxz));
  const F29 c4 = qb<0>(m2), c5 = qb<1>(m2);
  const F29 t3 = f29_norm(f29_sub(f29_add(t0, t1), c3));            // x1 y2 + x2 y1
  const F29 t4 = f29_norm(f29_sub(f29_add(t1, t2), c4));            // y1 z2 + y2 z1
  F29 y3 = f29_sub(f29_add(t0, t2), c5);                            // x1 z2 + x2 z1 (lazy)
  t0 = f29_norm(f29_add(f29_add(t0, t0), t0));
  t2 = OpsF29::mul_b3(t2);
  const F29 z3 = f29_norm(f29_add(t1, t2));
  t1 = f29_sub(t1, t2);                                             // product operand only
  y3 = OpsF29::mul_b3_lazy(y3);
  const F29 m3 = OpsF29I::mul(qsel(j, t3, t4, t1, y3), qsel(j, t1, y3, z3, t0));
  const F29 a = qb<0>(m3), b = qb<1>(m3), c = qb<2>(m3), d = qb<3>(m3);
  const F29 m4 = OpsF29I::mul(qsel(j, z3, t0, z3, t0), qsel(j, t4, t3, t4, t3));
  const F29 e = qb<0>(m4), f = qb<1>(m4);
  return G1W{f29_norm(f29_sub(a, b)), f29_norm(f29_add(c, d)), f29_norm(f29_add(e, f))};
}

template <int S> BN_DEV Fp qb_fp(const Fp& a) {          // lane S of this lane's quad
  Fp r;
#pragma unroll
  for (int i = 0; i < 8; ++i) r.v[i] = (u32)qbi<S>((int)a.v[i]);
  return r;
}
// svdw_back (bn254_hash.hpp; svdw.rs:180-262) without Jacobi symbols: the quad has a lane for each of the THREE candidates, so lane 0 takes
// the square-root candidate gx1^((p+1)/4), lane 1 that of gx2, lanes 2 and 3 that of gx3 -- one power chain, the same instruction stream
// on every lane -- and is_square(gx_i) (fp.rs:625-631: gx_i^((p-1)/2) in {0, 1}) is read off as "candidate^2 == gx_i" (true for 0 as
// well, like the reference).  The selected x and the root are the reference's: e1 ? x1 : e2 ? x2 : x3 and its gx^((p+1)/4).
BN_DEV bool svdw_back_quad(Fp& xo, Fp& yo, const Fp& u, const Fp& tv1, const Fp& tv2, const Fp& tv3, int j) {
  const Fp c2 = fp_const(C_SVDW[1]), c3 = fp_const(C_SVDW[2]), c4 = fp_const(C_SVDW[3]), z = fp_const(C_SVDW[4]);
  const Fp b = fp_small(3);
  const Fp tv4 = fp_mul(fp_mul(fp_mul(u, tv1), tv3), c3);
  const Fp x1 = fp_sub(c2, tv4);
  const Fp x2 = fp_add(c2, tv4);
  Fp x3 = fp_mul(fp_mul(tv2, tv2), tv3);
  x3 = fp_mul(fp_mul(x3, x3), c4);
  x3 = fp_add(x3, z);
  const Fp xc = fp_select(fp_select(x3, x2, j == 1), x1, j == 0);              // this lane's candidate
  const Fp gx = fp_add(fp_mul(fp_mul(xc, xc), xc), b);
  const Fp yc = fp_mul(gx, fp_pow_pm3_quarter(gx));
  const int sq = fp_eq(fp_mul(yc, yc), gx) ? 1 : 0;
  const bool e1 = qbi<0>(sq) != 0;
  const bool e2 = qbi<1>(sq) != 0 && !e1;
  const bool ok = e1 || e2 || qbi<2>(sq) != 0;
  Fp x = fp_select(x3, x1, e1);
  x = fp_select(x, x2, e2);
  Fp y = fp_select(fp_select(qb_fp<2>(yc), qb_fp<1>(yc), e2), qb_fp<0>(yc), e1);
  const bool e3 = fp_sgn0(u) == fp_sgn0(y);
  y = fp_select(fp_neg(y), y, e3);
  xo = x;
  yo = y;
  return ok;
}

// hash_to_curve (g1.rs:307-331) on the eight lanes of a group: every lane expands the message, quad q maps field element u_q (the three
// square-root candidates of a map on three lanes at once, the maps' inversions shared), the two points meet through the quad exchange and every lane adds them.
// Returns H(m) projective, the same on all eight lanes.
template <bool STAMPS>
BN_DEV G1P hash_to_g1_group(const uint8_t* msg, size_t len, const DstPrime& dp, int q, int j, u64* stamps) {
  auto stamp = [&](int k) { if (STAMPS && blockIdx.x == 0 && threadIdx.x == 0) stamps[k] = (u64)clock64(); };
  u64 em[12];
  expand_message_xmd96_words(em, msg, len, dp);
  stamp(1);
  u64 half[6];
#pragma unroll
  for (int k = 0; k < 6; ++k) half[k] = q ? em[6 + k] : em[k];
  const Fp u = fp_from_be48_words(half);
  const SvdwHalf me = svdw_front(u);
  const bool zme = fp_is_zero(me.d);
  const Fp one = fp_one(), zero = fp_zero();
  const Fp dme = fp_select(me.d, one, zme), dot = xq_fp(dme);
  const Fp t = fp_inv(fp_mul(dme, dot));                  // one inversion for both maps (svdw_map2)
  const Fp inv = fp_select(fp_mul(t, dot), zero, zme);
  stamp(2);
  Fp x, y;
  (void)svdw_back_quad(x, y, me.u, me.tv1, me.tv2, inv, j);
  stamp(3);
  const Fp xo = xq_fp(x), yo = xq_fp(y);
  const G1P a{q ? xo : x, q ? yo : y, one}, b{q ? x : xo, q ? y : yo, one};       // map(u0), map(u1) on every lane
  return g1_add(a, b);
}

// k * P on the eight lanes of a group (g1_scalar_mul's GLV split, bn254_pairing.hpp): k1 * (+-P) on quad 0, k2 * (+-phi P) on quad 1, each with
// its own accumulator and window table `tab[slot]` (LDS, slot = quad of the block), added once at the end.  p is the same on all eight lanes
// (any representative; Z = 0 is the identity); the affine result comes back on every lane.
typedef i32 (*QuadTab)[9][28];                              // per quad: 0P .. 8P, 27 words each
template <bool STAMPS>
BN_DEV void group_scalar_mul(Fp& x, Fp& y, bool& inf, const G1P& p, const u32 (&k)[8], int q, int j, int slot, QuadTab tab, u64* stamps) {
  auto stamp = [&](int s) { if (STAMPS && blockIdx.x == 0 && threadIdx.x == 0) stamps[s] = (u64)clock64(); };
  u32 m1[4], m2[4];
  bool n1, n2;
  glv_decompose(m1, n1, m2, n2, k);
  signed char dig[33];
  {
    u32 mq[4];
#pragma unroll
    for (int w = 0; w < 4; ++w) mq[w] = q ? m2[w] : m1[w];
    glv_digits(dig, mq);
  }
  auto put = [&](int m, const G1W& v) {
    if (j == 0) {
#pragma unroll
      for (int w = 0; w < 9; ++w) { tab[slot][m][w] = v.x.v[w]; tab[slot][m][9 + w] = v.y.v[w]; tab[slot][m][18 + w] = v.z.v[w]; }
    }
  };
  auto get = [&](int m) {
    G1W r;
#pragma unroll
    for (int w = 0; w < 9; ++w) { r.x.v[w] = tab[slot][m][w]; r.y.v[w] = tab[slot][m][9 + w]; r.z.v[w] = tab[slot][m][18 + w]; }
    return r;
  };
  {
    const F29 beta{{0x18ccb791, 0x175b1c3a, 0x0b83d6e2, 0x0e8ed071, 0x1282bee2, 0x04220e84, 0x1fe4017f, 0x15084d4a, 0x00169119}};   // beta 2^261 mod p
    G1W t1{f29_from_fp_reduced(p.x), f29_from_fp_reduced(p.y), f29_from_fp_reduced(p.z)};
    const bool pinf = OpsF29::is_zero(t1.z);                // canonical identity, as g1_scalar_mul_t
    t1.x = OpsF29::select(t1.x, OpsF29::zero(), pinf);
    t1.y = OpsF29::select(t1.y, OpsF29::one(), pinf);
    t1.z = OpsF29::select(t1.z, OpsF29::zero(), pinf);
    if (q) t1.x = OpsF29::mul(t1.x, beta);                  // quad 1 works on phi(P) = (beta x, y)
    if (q ? n2 : n1) t1.y = OpsF29::neg(t1.y);              // the table holds multiples of sign(k_q) times the base
    put(0, proj_zero<OpsF29>());
    put(1, t1);
    const G1W t2 = qdouble(t1, j);
    put(2, t2);
    const G1W t3 = qadd(t2, t1, j);
    put(3, t3);
    const G1W t4 = qdouble(t2, j);
    put(4, t4);
    put(5, qadd(t4, t1, j));
    const G1W t6 = qdouble(t3, j);
    put(6, t6);
    put(7, qadd(t6, t1, j));
    put(8, qdouble(t4, j));
  }
  __syncthreads();
  stamp(5);
  G1W res = proj_zero<OpsF29>();
#pragma unroll 1
  for (int w = 32; w >= 0; --w) {
    if (w != 32) {
#pragma unroll 1
      for (int r = 0; r < 4; ++r) res = qdouble(res, j);
    }
    const int d = dig[w], m = d < 0 ? -d : d;
    G1W e = get(m);
    e.y = OpsF29::select(e.y, OpsF29::neg(e.y), d < 0);
    res = qadd(res, e, j);
  }
  stamp(6);
  // ---- k1 P + k2 phi(P), to affine
  {
    const G1W o{xq_f29(res.x), xq_f29(res.y), xq_f29(res.z)};
    const G1W a{q ? o.x : res.x, q ? o.y : res.y, q ? o.z : res.z}, b{q ? res.x : o.x, q ? res.y : o.y, q ? res.z : o.z};
    res = qadd(a, b, j);
  }
  const G1P s{f29_to_fp(res.x), f29_to_fp(res.y), f29_to_fp(res.z)};
  g1_to_affine(x, y, inf, s);
}

// STAMPS (tools/ubench/sign_wide_phases.hip only): clock64() at the phase boundaries of the block's first group into stamps[0..7]
template <bool STAMPS>
__global__ void __launch_bounds__(WBLOCK)
k_bls_sign_wide(const u64* sk, const uint8_t* msgs, const u64* off, DstPrime dp, u64* oxy, uint8_t* oinf, size_t n, u64* stamps) {
  auto stamp = [&](int k) { if (STAMPS && blockIdx.x == 0 && threadIdx.x == 0) stamps[k] = (u64)clock64(); };
  stamp(0);
  __shared__ i32 tab[2 * EPB][9][28];
  const int lane = threadIdx.x & (GROUP - 1), q = lane >> 2, j = lane & 3, slot = (int)(threadIdx.x >> 2);
  size_t i = (size_t)blockIdx.x * EPB + (threadIdx.x >> 3);
  const bool live = i < n;
  if (!live) i = n - 1;                                     // tail groups repeat the last element and store nothing (no lane leaves early)
  // ---- H(m): both quads expand the message; quad 0 maps u0, quad 1 maps u1
  const G1P h = hash_to_g1_group<STAMPS>(msgs + off[i], (size_t)(off[i + 1] - off[i]), dp, q, j, stamps);
  stamp(4);
  // ---- sk * H: GLV halves on the two quads
  u32 k[8];
  load_scalar(k, sk, n, i);
  Fp x, y; bool inf;
  group_scalar_mul<STAMPS>(x, y, inf, h, k, q, j, slot, tab, stamps);
  if (live && lane == 0) {
    store_fp(oxy, n, i, 0, x); store_fp(oxy, n, i, 4, y);
    oinf[i] = inf ? 1 : 0;
  }
  stamp(7);
}
// k_i * P_i for small batches (sylow_hip_g1_scalar_mul_batch: group.rs:619-650 through the GLV split), eight lanes per product: one product
// ~0.35 ms against ~1.0 ms on one lane
__global__ void __launch_bounds__(WBLOCK)
k_g1_scalar_mul_wide(const u64* pxy, const uint8_t* pinf, const u64* ks, u64* oxy, uint8_t* oinf, size_t n) {
  __shared__ i32 tab[2 * EPB][9][28];
  const int lane = threadIdx.x & (GROUP - 1), q = lane >> 2, j = lane & 3, slot = (int)(threadIdx.x >> 2);
  size_t i = (size_t)blockIdx.x * EPB + (threadIdx.x >> 3);
  const bool live = i < n;
  if (!live) i = n - 1;
  const bool pi = pinf && pinf[i];
  const G1P p{load_fp(pxy, n, i, 0), load_fp(pxy, n, i, 4), pi ? fp_zero() : fp_one()};
  u32 k[8];
  load_scalar(k, ks, n, i);
  Fp x, y; bool inf;
  group_scalar_mul<false>(x, y, inf, p, k, q, j, slot, tab, nullptr);
  if (live && lane == 0) {
    store_fp(oxy, n, i, 0, x); store_fp(oxy, n, i, 4, y);
    oinf[i] = inf ? 1 : 0;
  }
}
// ecMul (EIP-196, examples/reth_bn128.rs:137-157) for single calls and small batches: every lane of the group decodes the 96 bytes, the
// product runs on the group's eight lanes, its first lane writes the 64 bytes (zeros with an error status, like k_evm_ecmul)
__global__ void __launch_bounds__(WBLOCK)
k_evm_ecmul_wide(const uint8_t* in, uint8_t* out, uint8_t* status, size_t n) {
  __shared__ i32 tab[2 * EPB][9][28];
  const int lane = threadIdx.x & (GROUP - 1), q = lane >> 2, j = lane & 3, slot = (int)(threadIdx.x >> 2);
  size_t i = (size_t)blockIdx.x * EPB + (threadIdx.x >> 3);
  const bool live = i < n;
  if (!live) i = n - 1;
  G1P a;
  const uint8_t st = evm_read_g1(a, in + 96 * i);
  u32 k[8];
  evm_read_scalar(k, in + 96 * i + 64);
  Fp x, y; bool inf;
  group_scalar_mul<false>(x, y, inf, a, k, q, j, slot, tab, nullptr);      // a is the identity when the point did not decode
  if (live && lane == 0) {
    status[i] = st;
    if (st) {
      __builtin_memset(out + 64 * i, 0, 64);
    } else {
      const Fp zero = fp_zero();
      write_be_fp(out + 64 * i, inf ? zero : fp_from_mont(x));
      write_be_fp(out + 64 * i + 32, inf ? zero : fp_from_mont(y));
    }
  }
}
// H(m_i) (or -H(m_i)) affine for small batches: the hash part alone, eight lanes per message (one message: ~0.4 ms against ~1.0 ms on one lane)
__global__ void __launch_bounds__(WBLOCK)
k_hash_to_g1_wide(const uint8_t* msgs, const u64* off, DstPrime dp, u64* oxy, uint8_t* oinf, size_t n, int negate) {
  const int lane = threadIdx.x & (GROUP - 1), q = lane >> 2, j = lane & 3;
  size_t i = (size_t)blockIdx.x * EPB + (threadIdx.x >> 3);
  const bool live = i < n;
  if (!live) i = n - 1;
  const G1P h = hash_to_g1_group<false>(msgs + off[i], (size_t)(off[i + 1] - off[i]), dp, q, j, nullptr);
  Fp x, y; bool inf;
  g1_to_affine(x, y, inf, h);
  if (negate && !inf) y = fp_neg(y);
  if (live && lane == 0) {
    store_fp(oxy, n, i, 0, x); store_fp(oxy, n, i, 4, y);
    oinf[i] = inf ? 1 : 0;
  }
}
}  // namespace wsign

namespace g1h {
int32_t hash_to_g1_wide(const uint8_t* msgs, const uint64_t* msg_offsets, const DstPrime& dp, uint64_t* out_xy, uint8_t* out_inf, size_t n, int negate, void* stream) {
  if (!n) return SYLOW_HIP_OK;
  wsign::k_hash_to_g1_wide<<<dim3((unsigned)((n + wsign::EPB - 1) / wsign::EPB)), dim3(wsign::WBLOCK), 0, (hipStream_t)stream>>>(msgs, msg_offsets, dp, out_xy, out_inf, n, negate);
  LAUNCHED();
}
// k_i * P_i, i < n <= sign_wide_max(), on eight lanes each
int32_t g1_scalar_mul_wide(const uint64_t* p_xy, const uint8_t* p_inf, const uint64_t* k, uint64_t* out_xy, uint8_t* out_inf, size_t n, void* stream) {
  if (!n) return SYLOW_HIP_OK;
  wsign::k_g1_scalar_mul_wide<<<dim3((unsigned)((n + wsign::EPB - 1) / wsign::EPB)), dim3(wsign::WBLOCK), 0, (hipStream_t)stream>>>(p_xy, p_inf, k, out_xy, out_inf, n);
  LAUNCHED();
}
int32_t evm_ecmul_wide(const uint8_t* in, uint8_t* out, uint8_t* status, size_t n, void* stream) {
  if (!n) return SYLOW_HIP_OK;
  wsign::k_evm_ecmul_wide<<<dim3((unsigned)((n + wsign::EPB - 1) / wsign::EPB)), dim3(wsign::WBLOCK), 0, (hipStream_t)stream>>>(in, out, status, n);
  LAUNCHED();
}
// signatures of n <= sign_wide_max() messages on eight lanes each
int32_t sign_wide(const uint64_t* sk, const uint8_t* msgs, const uint64_t* msg_offsets, uint64_t* sig_xy, uint8_t* sig_inf, size_t n, void* stream) {
  if (!n) return SYLOW_HIP_OK;
  DstPrime dp; host::dst_arg(dp, nullptr, 0);
  wsign::k_bls_sign_wide<false><<<dim3((unsigned)((n + wsign::EPB - 1) / wsign::EPB)), dim3(wsign::WBLOCK), 0, (hipStream_t)stream>>>(sk, msgs, msg_offsets, dp, sig_xy, sig_inf, n, nullptr);
  LAUNCHED();
}
}  // namespace g1h
