// Development harness for the lane-pair tower (sylow_amd/csrc/bn254_pair.hpp): op hooks with an iteration count
// (parity at iters = 1, throughput at iters >> 1) for the lane-pair and the single-lane implementation side by side.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include "../../sylow_amd/csrc/bn254_pairing.hpp"
#include "../../sylow_amd/csrc/bn254_pair29.hpp"
using namespace bn254;
#define TID ((size_t)blockIdx.x * blockDim.x + threadIdx.x)
BN_DEV Fp load_plain(const u64* __restrict__ base, size_t n, size_t i, int w0) {
  Fp r;
#pragma unroll
  for (int k = 0; k < 4; ++k) { u64 w = base[(size_t)(w0 + k) * n + i]; r.v[2 * k] = (u32)w; r.v[2 * k + 1] = (u32)(w >> 32); }
  return r;
}
BN_DEV void store_plain(u64* __restrict__ base, size_t n, size_t i, int w0, const Fp& a) {
#pragma unroll
  for (int k = 0; k < 4; ++k) base[(size_t)(w0 + k) * n + i] = (u64)a.v[2 * k] | ((u64)a.v[2 * k + 1] << 32);
}
BN_DEV Fp load_fp(const u64* base, size_t n, size_t i, int w0) { return fp_to_mont(load_plain(base, n, i, w0)); }
BN_DEV void store_fp(u64* base, size_t n, size_t i, int w0, const Fp& a) { store_plain(base, n, i, w0, fp_from_mont(a)); }
BN_DEV pl::S2 load_s2(const u64* base, size_t n, size_t i, int w0, int odd) { return pl::S2{load_fp(base, n, i, w0 + 4 * odd)}; }
BN_DEV void load_s12(pl::S12& r, const u64* base, size_t n, size_t i, int odd) {
  r.c0.c0 = load_s2(base, n, i, 0, odd); r.c0.c1 = load_s2(base, n, i, 8, odd); r.c0.c2 = load_s2(base, n, i, 16, odd);
  r.c1.c0 = load_s2(base, n, i, 24, odd); r.c1.c1 = load_s2(base, n, i, 32, odd); r.c1.c2 = load_s2(base, n, i, 40, odd);
}
BN_DEV void store_s12(u64* base, size_t n, size_t i, int odd, const pl::S12& a) {
  store_fp(base, n, i, 0 + 4 * odd, a.c0.c0.c); store_fp(base, n, i, 8 + 4 * odd, a.c0.c1.c); store_fp(base, n, i, 16 + 4 * odd, a.c0.c2.c);
  store_fp(base, n, i, 24 + 4 * odd, a.c1.c0.c); store_fp(base, n, i, 32 + 4 * odd, a.c1.c1.c); store_fp(base, n, i, 40 + 4 * odd, a.c1.c2.c);
}
enum { OP_MUL = 0, OP_SQR = 1, OP_INV = 2, OP_FROB1 = 3, OP_FROB2 = 4, OP_FROB3 = 5, OP_SPARSE = 6, OP_CYCSQR = 7, OP_EXPZ = 8, OP_FINAL = 9 };

__global__ void __launch_bounds__(256, 2) k_pl_op(int op, const u64* a, const u64* b, u64* out, size_t n, int iters) {
  size_t t = TID, i = t >> 1;
  const int odd = (int)(t & 1);
  if (i >= n) return;
  pl::S12 x, y, r;
  load_s12(x, a, n, i, odd);
  if (b) load_s12(y, b, n, i, odd);
#pragma unroll 1
  for (int it = 0; it < iters; ++it) {
    switch (op) {
      case OP_MUL: r = pl::s12_mul(x, y); break;
      case OP_SQR: r = pl::s12_sqr(x); break;
      case OP_INV: r = pl::s12_inv(x); break;
      case OP_FROB1: r = pl::s12_frobenius<1>(x); break;
      case OP_FROB2: r = pl::s12_frobenius<2>(x); break;
      case OP_FROB3: r = pl::s12_frobenius<3>(x); break;
      case OP_SPARSE: r = pl::s12_sparse_mul(x, y.c0.c0, y.c0.c1, y.c0.c2); break;   // ell = first 3 Fp2 of b: (l0, lvw, lvv)
      case OP_CYCSQR: r = pl::cyclotomic_sqr(x); break;
      case OP_EXPZ: pl::exp_by_neg_z(r, x); break;
      default: pl::final_exponentiation(r, x); break;
    }
    x = r;
  }
  store_s12(out, n, i, odd, r);
}
__global__ void __launch_bounds__(256, 2) k_pw_op(int op, const u64* a, const u64* b, u64* out, size_t n, int iters) {
  size_t t = TID, i = t >> 1;
  const int odd = (int)(t & 1);
  if (i >= n) return;
  pl::S12 sx, sy, sr;
  load_s12(sx, a, n, i, odd);
  if (b) load_s12(sy, b, n, i, odd);
  if (op == OP_FINAL) {
#pragma unroll 1
    for (int it = 0; it < iters; ++it) { pl::final_exponentiation29(sr, sx); sx = sr; }
  } else {
    pl::W12 x, y, r;
    pl::w12_from_s12(x, sx);
    if (b) pl::w12_from_s12(y, sy);
#pragma unroll 1
    for (int it = 0; it < iters; ++it) {
      switch (op) {
        case OP_MUL: r = pl::w12_mul(x, y); break;
        case OP_SQR: r = pl::w12_sqr(x); break;
        case OP_FROB1: r = pl::w12_frobenius<1>(x); break;
        case OP_FROB2: r = pl::w12_frobenius<2>(x); break;
        case OP_FROB3: r = pl::w12_frobenius<3>(x); break;
        case OP_SPARSE: r = pl::w12_sparse_mul(x, y.c0.c0, y.c0.c1, y.c0.c2); break;
        case OP_CYCSQR: r = pl::w12_cyclotomic_sqr(x); break;
        default: pl::exp_by_neg_z29(r, x); break;
      }
      x = r;
    }
    pl::w12_to_s12(sr, r);
  }
  store_s12(out, n, i, odd, sr);
}
__global__ void __launch_bounds__(256, 2) k_pw_pairing(const u64* pxy, const u64* qxy, u64* gout, u64* fout, size_t n, int stage) {
  size_t t = TID, i = t >> 1;
  const int odd = (int)(t & 1);
  if (i >= n) return;
  Fp px = load_fp(pxy, n, i, 0), py = load_fp(pxy, n, i, 4);
  pl::S2 qx = load_s2(qxy, n, i, 0, odd), qy = load_s2(qxy, n, i, 8, odd);
  pl::S12 f, g;
  if (stage != 2) pl::miller_loop29(f, px, py, qx, qy); else load_s12(f, fout, n, i, odd);
  if (fout && stage != 2) store_s12(fout, n, i, odd, f);
  if (stage != 1) { pl::final_exponentiation29(g, f); store_s12(gout, n, i, odd, g); }
}
BN_DEV void load_fp12(Fp12& r, const u64* base, size_t n, size_t i) {
  Fp* f = reinterpret_cast<Fp*>(&r);
#pragma unroll
  for (int k = 0; k < 12; ++k) f[k] = load_fp(base, n, i, 4 * k);
}
BN_DEV void store_fp12(u64* base, size_t n, size_t i, const Fp12& r) {
  const Fp* f = reinterpret_cast<const Fp*>(&r);
#pragma unroll
  for (int k = 0; k < 12; ++k) store_fp(base, n, i, 4 * k, f[k]);
}
__global__ void __launch_bounds__(256, 2) k_sl_op(int op, const u64* a, const u64* b, u64* out, size_t n, int iters) {
  size_t i = TID;
  if (i >= n) return;
  Fp12 x, y, r;
  load_fp12(x, a, n, i);
  if (b) load_fp12(y, b, n, i);
#pragma unroll 1
  for (int it = 0; it < iters; ++it) {
    switch (op) {
      case OP_MUL: fp12_mul(r, x, y); break;
      case OP_SQR: fp12_sqr(r, x); break;
      case OP_INV: fp12_inv(r, x); break;
      case OP_SPARSE: fp12_sparse_mul(r, x, y.c0.c0, y.c0.c1, y.c0.c2); break;
      case OP_CYCSQR: cyclotomic_sqr(r, x); break;
      case OP_EXPZ: exp_by_neg_z_sat(r, x); break;
      default: final_exponentiation(r, x); break;
    }
    x = r;
  }
  store_fp12(out, n, i, r);
}
// pairing: P [8][n], Q [16][n] -> Gt [48][n]
__global__ void __launch_bounds__(256, 2) k_pl_pairing(const u64* pxy, const u64* qxy, u64* gout, u64* fout, size_t n, int stage) {
  size_t t = TID, i = t >> 1;
  const int odd = (int)(t & 1);
  if (i >= n) return;
  Fp px = load_fp(pxy, n, i, 0), py = load_fp(pxy, n, i, 4);
  pl::S2 qx = load_s2(qxy, n, i, 0, odd), qy = load_s2(qxy, n, i, 8, odd);
  pl::S12 f, g;
  if (stage != 2) pl::miller_loop(f, px, py, qx, qy); else load_s12(f, fout, n, i, odd);
  if (fout && stage != 2) store_s12(fout, n, i, odd, f);
  if (stage != 1) { pl::final_exponentiation(g, f); store_s12(gout, n, i, odd, g); }
}
__global__ void __launch_bounds__(256, 2) k_pg_pairing(const u64* pxy, const u64* qxy, u64* gout, u64* fout, size_t n, int stage) {
  size_t t = TID, i = t >> 1;
  const int odd = (int)(t & 1);
  if (i >= n) return;
  Fp px = load_fp(pxy, n, i, 0), py = load_fp(pxy, n, i, 4);
  pl::S2 qx = load_s2(qxy, n, i, 0, odd), qy = load_s2(qxy, n, i, 8, odd);
  pl::S12 f, g;
  if (stage != 2) pl::miller_loop29g(f, px, py, qx, qy); else load_s12(f, fout, n, i, odd);
  if (fout && stage != 2) store_s12(fout, n, i, odd, f);
  if (stage != 1) { pl::final_exponentiation29(g, f); store_s12(gout, n, i, odd, g); }
}
__global__ void __launch_bounds__(256, 2) k_sl_pairing(const u64* pxy, const u64* qxy, u64* gout, u64* fout, size_t n, int stage) {
  size_t i = TID;
  if (i >= n) return;
  Fp px = load_fp(pxy, n, i, 0), py = load_fp(pxy, n, i, 4);
  Fp2 qx{load_fp(qxy, n, i, 0), load_fp(qxy, n, i, 4)}, qy{load_fp(qxy, n, i, 8), load_fp(qxy, n, i, 12)};
  Fp12 f, g;
  if (stage != 2) miller_loop(f, px, py, qx, qy); else load_fp12(f, fout, n, i);
  if (fout && stage != 2) store_fp12(fout, n, i, f);
  if (stage != 1) { final_exponentiation(g, f); store_fp12(gout, n, i, g); }
}

static float timed(void (*launch)(void*), void* ctx, int reps) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  launch(ctx);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int r = 0; r < reps; ++r) launch(ctx);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  hipError_t err = hipGetLastError();
  if (err != hipSuccess) { fprintf(stderr, "HIP error: %s\n", hipGetErrorString(err)); return -1.f; }
  return ms / reps;
}
struct OpCtx { int pair, op; const u64 *a, *b; u64* out; size_t n; int iters; };
static void launch_op(void* p) {
  OpCtx* c = (OpCtx*)p;
  if (c->pair == 2) k_pw_op<<<dim3((unsigned)((2 * c->n + 255) / 256)), dim3(256)>>>(c->op, c->a, c->b, c->out, c->n, c->iters);
  else if (c->pair) k_pl_op<<<dim3((unsigned)((2 * c->n + 255) / 256)), dim3(256)>>>(c->op, c->a, c->b, c->out, c->n, c->iters);
  else k_sl_op<<<dim3((unsigned)((c->n + 255) / 256)), dim3(256)>>>(c->op, c->a, c->b, c->out, c->n, c->iters);
}
struct PairCtx { int pair; const u64 *p, *q; u64 *g, *f; size_t n; int stage; };
static void launch_pairing(void* p) {
  PairCtx* c = (PairCtx*)p;
  if (c->pair == 3) k_pg_pairing<<<dim3((unsigned)((2 * c->n + 255) / 256)), dim3(256)>>>(c->p, c->q, c->g, c->f, c->n, c->stage);
  else if (c->pair == 2) k_pw_pairing<<<dim3((unsigned)((2 * c->n + 255) / 256)), dim3(256)>>>(c->p, c->q, c->g, c->f, c->n, c->stage);
  else if (c->pair) k_pl_pairing<<<dim3((unsigned)((2 * c->n + 255) / 256)), dim3(256)>>>(c->p, c->q, c->g, c->f, c->n, c->stage);
  else k_sl_pairing<<<dim3((unsigned)((c->n + 255) / 256)), dim3(256)>>>(c->p, c->q, c->g, c->f, c->n, c->stage);
}
extern "C" {
// returns average ms per launch (reps timed launches after one warm-up), < 0 on error
float pl_op(int pair, int op, const u64* a, const u64* b, u64* out, size_t n, int iters, int reps) {
  OpCtx c{pair, op, a, b, out, n, iters};
  return timed(launch_op, &c, reps);
}
float pl_pairing(int pair, const u64* p, const u64* q, u64* g, u64* f, size_t n, int reps, int stage) {
  PairCtx c{pair, p, q, g, f, n, stage};
  return timed(launch_pairing, &c, reps);
}
}
