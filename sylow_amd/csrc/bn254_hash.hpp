// Hash-to-G1 on the device: Keccak-256, RFC 9380 expand_message_xmd, hash_to_field, and the
// Shallue-van de Woestijne map for y^2 = x^3 + 3.  Batched replacement for sylow's
// src/hasher.rs:84-128,157-250 (XMDExpander<Keccak256>), src/svdw.rs:180-262 and
// src/groups/g1.rs:307-331.  One message per lane.
//
// Keccak comes from the un-vendored crate sha3 0.11.0-pre.4 in the reference (Cargo.toml:41);
// Keccak-f[1600] is restated here from FIPS 202 with the original Keccak padding (0x01 .. 0x80).
#pragma once
#include "bn254_pairing.hpp"

namespace bn254 {

struct DstPrime {        // DST || I2OSP(len(DST), 1), already shortened if the tag was > 255 bytes
  uint8_t bytes[256];
  uint32_t len;          // length of DST' (<= 256)
  // The one-block message of b_1 .. b_3 (hasher.rs:223-245: 32 bytes, the block counter, DST') with everything but the first 32 bytes
  // filled in: byte 32 = 0 (the counter is OR-ed in), DST', Keccak's 0x01 .. 0x80 padding -- as the 17 little-endian rate words.
  // Valid when 33 + len <= 135 (tail_ok); longer tags take the byte-wise absorber.
  uint64_t tail[17];
  uint32_t tail_ok;
};

__host__ __device__ inline u64 rotl64(u64 x, int n) { return n ? (x << n) | (x >> (64 - n)) : x; }

// Device form of the permutation on 32-bit halves: the five-way column parities as two three-input XORs and chi as one
// a ^ (~b & c) per half (v_bitop3_b32), every rotation as two v_alignbit_b32: ~190 VALU instructions per round where the generic 64-bit
// source compiles to 288.
#if defined(__HIP_DEVICE_COMPILE__)
__device__ inline void k_rotl(u32& olo, u32& ohi, u32 lo, u32 hi, int n) {     // n a compile-time constant in 1 .. 63
  if (n == 32) { olo = hi; ohi = lo; return; }
  if (n > 32) { const u32 t = lo; lo = hi; hi = t; n -= 32; }
  olo = __builtin_amdgcn_alignbit(lo, hi, 32 - n);
  ohi = __builtin_amdgcn_alignbit(hi, lo, 32 - n);
}
static __device__ const u32 KECCAK_RC32[24][2] = {
      {0x00000001u, 0x00000000u}, {0x00008082u, 0x00000000u}, {0x0000808Au, 0x80000000u}, {0x80008000u, 0x80000000u}, {0x0000808Bu, 0x00000000u},
      {0x80000001u, 0x00000000u}, {0x80008081u, 0x80000000u}, {0x00008009u, 0x80000000u}, {0x0000008Au, 0x00000000u}, {0x00000088u, 0x00000000u},
      {0x80008009u, 0x00000000u}, {0x8000000Au, 0x00000000u}, {0x8000808Bu, 0x00000000u}, {0x0000008Bu, 0x80000000u}, {0x00008089u, 0x80000000u},
      {0x00008003u, 0x80000000u}, {0x00008002u, 0x80000000u}, {0x00000080u, 0x80000000u}, {0x0000800Au, 0x00000000u}, {0x8000000Au, 0x80000000u},
      {0x80008081u, 0x80000000u}, {0x00008080u, 0x80000000u}, {0x80000001u, 0x00000000u}, {0x80008008u, 0x80000000u}};
__device__ inline void keccak_f1600_halves(u64 (&st)[25]) {
  u32 l[25], h[25];
#pragma unroll
  for (int i = 0; i < 25; ++i) { l[i] = (u32)st[i]; h[i] = (u32)(st[i] >> 32); }
#pragma unroll 1
  for (int rnd = 0; rnd < 24; ++rnd) {
    u32 cl[5], ch[5], dl[5], dh[5];
#pragma unroll
    for (int x = 0; x < 5; ++x) {
      cl[x] = __builtin_amdgcn_bitop3_b32(__builtin_amdgcn_bitop3_b32(l[x], l[x + 5], l[x + 10], 0x96), l[x + 15], l[x + 20], 0x96);
      ch[x] = __builtin_amdgcn_bitop3_b32(__builtin_amdgcn_bitop3_b32(h[x], h[x + 5], h[x + 10], 0x96), h[x + 15], h[x + 20], 0x96);
    }
#pragma unroll
    for (int x = 0; x < 5; ++x) {                     // d[x] = c[x - 1] ^ rotl(c[x + 1], 1)
      u32 rl, rh;
      k_rotl(rl, rh, cl[(x + 1) % 5], ch[(x + 1) % 5], 1);
      dl[x] = cl[(x + 4) % 5] ^ rl;
      dh[x] = ch[(x + 4) % 5] ^ rh;
    }
#pragma unroll
    for (int i = 0; i < 25; ++i) { l[i] ^= dl[i % 5]; h[i] ^= dh[i % 5]; }
    // rho + pi: b[y][2x + 3y] = rotl(s[x][y], r[x][y]) -- destination index and offset of every source lane
    constexpr int DST[25] = {0, 10, 20, 5, 15, 16, 1, 11, 21, 6, 7, 17, 2, 12, 22, 23, 8, 18, 3, 13, 14, 24, 9, 19, 4};
    constexpr int ROT[25] = {0, 1, 62, 28, 27, 36, 44, 6, 55, 20, 3, 10, 43, 25, 39, 41, 45, 15, 21, 8, 18, 2, 61, 56, 14};
    u32 bl[25], bh[25];
    bl[0] = l[0]; bh[0] = h[0];
#pragma unroll
    for (int i = 1; i < 25; ++i) k_rotl(bl[DST[i]], bh[DST[i]], l[i], h[i], ROT[i]);
#pragma unroll
    for (int y = 0; y < 25; y += 5) {
#pragma unroll
      for (int x = 0; x < 5; ++x) {                   // chi: b[x] ^ (~b[x + 1] & b[x + 2])
        l[y + x] = __builtin_amdgcn_bitop3_b32(bl[y + x], bl[y + (x + 1) % 5], bl[y + (x + 2) % 5], 0xD2);
        h[y + x] = __builtin_amdgcn_bitop3_b32(bh[y + x], bh[y + (x + 1) % 5], bh[y + (x + 2) % 5], 0xD2);
      }
    }
    l[0] ^= KECCAK_RC32[rnd][0];
    h[0] ^= KECCAK_RC32[rnd][1];
  }
#pragma unroll
  for (int i = 0; i < 25; ++i) st[i] = ((u64)h[i] << 32) | l[i];
}
#endif

__host__ __device__ inline void keccak_f1600(u64 (&s)[25]) {
#if defined(__HIP_DEVICE_COMPILE__)
  keccak_f1600_halves(s);
  return;
#endif
  const u64 RC[24] = {
      0x0000000000000001ull, 0x0000000000008082ull, 0x800000000000808Aull, 0x8000000080008000ull, 0x000000000000808Bull,
      0x0000000080000001ull, 0x8000000080008081ull, 0x8000000000008009ull, 0x000000000000008Aull, 0x0000000000000088ull,
      0x0000000080008009ull, 0x000000008000000Aull, 0x000000008000808Bull, 0x800000000000008Bull, 0x8000000000008089ull,
      0x8000000000008003ull, 0x8000000000008002ull, 0x8000000000000080ull, 0x000000000000800Aull, 0x800000008000000Aull,
      0x8000000080008081ull, 0x8000000000008080ull, 0x0000000080000001ull, 0x8000000080008008ull};
#pragma unroll 1
  for (int rnd = 0; rnd < 24; ++rnd) {
    u64 c0 = s[0] ^ s[5] ^ s[10] ^ s[15] ^ s[20];
    u64 c1 = s[1] ^ s[6] ^ s[11] ^ s[16] ^ s[21];
    u64 c2 = s[2] ^ s[7] ^ s[12] ^ s[17] ^ s[22];
    u64 c3 = s[3] ^ s[8] ^ s[13] ^ s[18] ^ s[23];
    u64 c4 = s[4] ^ s[9] ^ s[14] ^ s[19] ^ s[24];
    u64 d0 = c4 ^ rotl64(c1, 1), d1 = c0 ^ rotl64(c2, 1), d2 = c1 ^ rotl64(c3, 1), d3 = c2 ^ rotl64(c4, 1), d4 = c3 ^ rotl64(c0, 1);
#pragma unroll
    for (int y = 0; y < 25; y += 5) { s[y] ^= d0; s[y + 1] ^= d1; s[y + 2] ^= d2; s[y + 3] ^= d3; s[y + 4] ^= d4; }
    // rho + pi
    u64 b[25];
    b[0] = s[0];
    b[10] = rotl64(s[1], 1);   b[20] = rotl64(s[2], 62);  b[5] = rotl64(s[3], 28);   b[15] = rotl64(s[4], 27);
    b[16] = rotl64(s[5], 36);  b[1] = rotl64(s[6], 44);   b[11] = rotl64(s[7], 6);   b[21] = rotl64(s[8], 55);
    b[6] = rotl64(s[9], 20);   b[7] = rotl64(s[10], 3);   b[17] = rotl64(s[11], 10); b[2] = rotl64(s[12], 43);
    b[12] = rotl64(s[13], 25); b[22] = rotl64(s[14], 39); b[23] = rotl64(s[15], 41); b[8] = rotl64(s[16], 45);
    b[18] = rotl64(s[17], 15); b[3] = rotl64(s[18], 21);  b[13] = rotl64(s[19], 8);  b[14] = rotl64(s[20], 18);
    b[24] = rotl64(s[21], 2);  b[9] = rotl64(s[22], 61);  b[19] = rotl64(s[23], 56); b[4] = rotl64(s[24], 14);
#pragma unroll
    for (int y = 0; y < 25; y += 5) {
      s[y] = b[y] ^ (~b[y + 1] & b[y + 2]);
      s[y + 1] = b[y + 1] ^ (~b[y + 2] & b[y + 3]);
      s[y + 2] = b[y + 2] ^ (~b[y + 3] & b[y + 4]);
      s[y + 3] = b[y + 3] ^ (~b[y + 4] & b[y]);
      s[y + 4] = b[y + 4] ^ (~b[y] & b[y + 1]);
    }
    s[0] ^= RC[rnd];
  }
}

// Streaming Keccak-256 absorber: rate 136 bytes = 17 words.  Bytes are collected in a 64-bit staging register and XORed into the
// state one WORD at a time through a select over the 17 rate words, so the state is only ever indexed with constants: it lives in
// registers across the permutations (a dynamically indexed state sits in the stack frame -- every round of every permutation then
// went through scratch memory: 5.7 k scratch instructions per hash, 38 % of the hashing kernel's wave cycles waiting).
struct Keccak256 {
  u64 s[25];
  u64 cur;
  uint32_t fill;
  __host__ __device__ inline void init() {
#pragma unroll
    for (int i = 0; i < 25; ++i) s[i] = 0;
    cur = 0;
    fill = 0;
  }
  __host__ __device__ inline void flush_word(uint32_t idx) {          // s[idx] ^= cur, idx in 0..16
#pragma unroll
    for (uint32_t w = 0; w < 17; ++w) s[w] ^= (w == idx) ? cur : 0ull;
    cur = 0;
  }
  __host__ __device__ inline void put(uint8_t byte) {
    cur |= (u64)byte << (8 * (fill & 7));
    ++fill;
    if ((fill & 7) == 0) {
      flush_word((fill >> 3) - 1);
      if (fill == 136) { keccak_f1600(s); fill = 0; }
    }
  }
  __host__ __device__ inline void update(const uint8_t* d, size_t n) {
    for (size_t i = 0; i < n; ++i) put(d[i]);
  }
  __host__ __device__ inline void finish(uint8_t out[32]) {
    cur |= (u64)0x01 << (8 * (fill & 7));           // original Keccak padding 0x01 .. 0x80
    flush_word(fill >> 3);
    s[16] ^= 0x8000000000000000ull;                   // last byte of the 136-byte rate block
    keccak_f1600(s);
#pragma unroll
    for (int i = 0; i < 32; ++i) out[i] = (uint8_t)(s[i >> 3] >> (8 * (i & 7)));
  }
};

// hasher.rs:157-173: DST' from DST (host side, once per call)
inline void make_dst_prime(DstPrime& dp, const uint8_t* dst, size_t len) {
  uint8_t h[32];
  if (len > 255) {
    Keccak256 k;
    k.init();
    k.update((const uint8_t*)"H2C-OVERSIZE-DST-", 17);
    k.update(dst, len);
    k.finish(h);
    dst = h;
    len = 32;
  }
  for (size_t i = 0; i < len; ++i) dp.bytes[i] = dst[i];
  dp.bytes[len] = (uint8_t)len;
  dp.len = (uint32_t)len + 1;
  uint8_t blk[136];
  for (int i = 0; i < 136; ++i) blk[i] = 0;
  dp.tail_ok = (33 + dp.len <= 135) ? 1u : 0u;
  if (dp.tail_ok) {
    for (uint32_t i = 0; i < dp.len; ++i) blk[33 + i] = dp.bytes[i];
    blk[33 + dp.len] ^= 0x01;
    blk[135] ^= 0x80;
  }
  for (int w = 0; w < 17; ++w) {
    uint64_t v = 0;
    for (int j = 0; j < 8; ++j) v |= (uint64_t)blk[8 * w + j] << (8 * j);
    dp.tail[w] = v;
  }
}

// hasher.rs:201-250 with len_in_bytes = 96 (ell = 3): out = b1 || b2 || b3
__device__ inline void expand_message_xmd96(uint8_t out[96], const uint8_t* msg, size_t msg_len, const DstPrime& dp) {
  Keccak256 k;
  uint8_t b0[32];
  k.init();
  // Z_pad: one whole rate block of zeros = one permutation of the zero state -- a constant (Keccak-f[1600](0), first lane the
  // well-known 0xF1258F7940E1DDE7), not worth 24 rounds per message
  {
    const u64 z[25] = {0xf1258f7940e1dde7ull, 0x84d5ccf933c0478aull, 0xd598261ea65aa9eeull, 0xbd1547306f80494dull, 0x8b284e056253d057ull,
                       0xff97a42d7f8e6fd4ull, 0x90fee5a0a44647c4ull, 0x8c5bda0cd6192e76ull, 0xad30a6f71b19059cull, 0x30935ab7d08ffc64ull,
                       0xeb5aa93f2317d635ull, 0xa9a6e6260d712103ull, 0x81a57c16dbcf555full, 0x43b831cd0347c826ull, 0x01f22f1a11a5569full,
                       0x05e5635a21d9ae61ull, 0x64befef28cc970f2ull, 0x613670957bc46611ull, 0xb87c5a554fd00ecbull, 0x8c3ee88a1ccf32c8ull,
                       0x940c7922ae3a2614ull, 0x1841f924a2c509e4ull, 0x16f53526e70465c2ull, 0x75f644e97f30a13bull, 0xeaf1ff7b5ceca249ull};
#pragma unroll
    for (int i = 0; i < 25; ++i) k.s[i] = z[i];
  }
  k.update(msg, msg_len);
  k.put(0); k.put(96);          // l_i_b_str = I2OSP(96, 2)
  k.put(0);                     // I2OSP(0, 1)
  k.update(dp.bytes, dp.len);
  k.finish(b0);
  uint8_t prev[32];
  for (int i = 0; i < 32; ++i) prev[i] = 0;
  if (dp.tail_ok) {
    // b_i = H((b_0 xor b_(i-1)) || i || DST'): one block whose words 4 .. 16 are the same for every message -- no byte loop
    u64 x0[4], xp[4] = {0, 0, 0, 0};
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      u64 v = 0;
#pragma unroll
      for (int j = 0; j < 8; ++j) v |= (u64)b0[8 * w + j] << (8 * j);
      x0[w] = v;
    }
#pragma unroll 1
    for (int blk = 1; blk <= 3; ++blk) {
#pragma unroll
      for (int w = 0; w < 4; ++w) k.s[w] = x0[w] ^ xp[w];
#pragma unroll
      for (int w = 4; w < 17; ++w) k.s[w] = dp.tail[w];
      k.s[4] |= (u64)blk;
#pragma unroll
      for (int w = 17; w < 25; ++w) k.s[w] = 0;
      keccak_f1600(k.s);
#pragma unroll
      for (int w = 0; w < 4; ++w) xp[w] = k.s[w];
#pragma unroll
      for (int i = 0; i < 32; ++i) out[32 * (blk - 1) + i] = (uint8_t)(xp[i >> 3] >> (8 * (i & 7)));
    }
    return;
  }
#pragma unroll 1
  for (int blk = 1; blk <= 3; ++blk) {
    k.init();
    for (int i = 0; i < 32; ++i) k.put(b0[i] ^ prev[i]);     // b_0 for blk = 1 (prev = 0), else b_0 xor b_(i-1)
    k.put((uint8_t)blk);
    k.update(dp.bytes, dp.len);
    k.finish(prev);
    for (int i = 0; i < 32; ++i) out[32 * (blk - 1) + i] = prev[i];
  }
}

// The same 96 bytes as twelve little-endian words (w[4 (i - 1) + k] = word k of b_i), every index a constant: for callers that keep the
// bytes in registers and pick a half per LANE (sign_wide.hip) -- a byte array written under a loop counter and read at a lane-dependent
// offset costs ~100 compare-and-select instructions per byte there.
__device__ inline void expand_message_xmd96_words(u64 (&w)[12], const uint8_t* msg, size_t msg_len, const DstPrime& dp) {
  if (!dp.tail_ok) {                                  // long tags: the byte-wise route, packed afterwards
    uint8_t em[96];
    expand_message_xmd96(em, msg, msg_len, dp);
#pragma unroll
    for (int k = 0; k < 12; ++k) {
      u64 v = 0;
#pragma unroll
      for (int j = 0; j < 8; ++j) v |= (u64)em[8 * k + j] << (8 * j);
      w[k] = v;
    }
    return;
  }
  Keccak256 k;
  k.init();
  {
    const u64 z[25] = {0xf1258f7940e1dde7ull, 0x84d5ccf933c0478aull, 0xd598261ea65aa9eeull, 0xbd1547306f80494dull, 0x8b284e056253d057ull,
                       0xff97a42d7f8e6fd4ull, 0x90fee5a0a44647c4ull, 0x8c5bda0cd6192e76ull, 0xad30a6f71b19059cull, 0x30935ab7d08ffc64ull,
                       0xeb5aa93f2317d635ull, 0xa9a6e6260d712103ull, 0x81a57c16dbcf555full, 0x43b831cd0347c826ull, 0x01f22f1a11a5569full,
                       0x05e5635a21d9ae61ull, 0x64befef28cc970f2ull, 0x613670957bc46611ull, 0xb87c5a554fd00ecbull, 0x8c3ee88a1ccf32c8ull,
                       0x940c7922ae3a2614ull, 0x1841f924a2c509e4ull, 0x16f53526e70465c2ull, 0x75f644e97f30a13bull, 0xeaf1ff7b5ceca249ull};
#pragma unroll
    for (int i = 0; i < 25; ++i) k.s[i] = z[i];           // Z_pad absorbed (expand_message_xmd96)
  }
  k.update(msg, msg_len);
  k.put(0); k.put(96);
  k.put(0);
  k.update(dp.bytes, dp.len);
  // finish() without the byte output: b_0 stays in the state's first four words
  k.cur |= (u64)0x01 << (8 * (k.fill & 7));
  k.flush_word(k.fill >> 3);
  k.s[16] ^= 0x8000000000000000ull;
  keccak_f1600(k.s);
  const u64 x0[4] = {k.s[0], k.s[1], k.s[2], k.s[3]};
  u64 xp[4] = {0, 0, 0, 0};
#pragma unroll
  for (int blk = 1; blk <= 3; ++blk) {
#pragma unroll
    for (int i = 0; i < 4; ++i) k.s[i] = x0[i] ^ xp[i];
#pragma unroll
    for (int i = 4; i < 17; ++i) k.s[i] = dp.tail[i];
    k.s[4] |= (u64)blk;
#pragma unroll
    for (int i = 17; i < 25; ++i) k.s[i] = 0;
    keccak_f1600(k.s);
#pragma unroll
    for (int i = 0; i < 4; ++i) { xp[i] = k.s[i]; w[4 * (blk - 1) + i] = k.s[i]; }
  }
}

// value = hi * 2^256 + lo (plain limbs, hi < 2^128) reduced mod p, in Montgomery form
__device__ inline Fp fp_from_wide_limbs(const Fp& lo, const Fp& hi) {
  // mont(hi) * R^2 / R = mont(hi * R)
  Fp h = fp_mul(fp_to_mont(hi), fp_r2());
  return fp_add(h, fp_to_mont(lo));
}
// fp_from_be48 on six little-endian words holding the 48 bytes
__device__ inline Fp fp_from_be48_words(const u64 (&w)[6]) {
  Fp lo, hi;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int o = 44 - 4 * i;                         // byte offset of the big-endian 32-bit group
    lo.v[i] = __builtin_bswap32((u32)(w[o >> 3] >> (8 * (o & 4))));
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int o = 12 - 4 * i;
    hi.v[i] = __builtin_bswap32((u32)(w[o >> 3] >> (8 * (o & 4))));
  }
  hi.v[4] = hi.v[5] = hi.v[6] = hi.v[7] = 0;
  return fp_from_wide_limbs(lo, hi);
}

// hasher.rs:84-128: a 48-byte big-endian integer reduced mod p, in Montgomery form
__device__ inline Fp fp_from_be48(const uint8_t* b) {
  Fp lo, hi;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const uint8_t* q = b + 16 + 4 * (7 - i);
    lo.v[i] = ((u32)q[0] << 24) | ((u32)q[1] << 16) | ((u32)q[2] << 8) | (u32)q[3];
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const uint8_t* q = b + 4 * (3 - i);
    hi.v[i] = ((u32)q[0] << 24) | ((u32)q[1] << 16) | ((u32)q[2] << 8) | (u32)q[3];
  }
  hi.v[4] = hi.v[5] = hi.v[6] = hi.v[7] = 0;
  return fp_from_wide_limbs(lo, hi);
}

// is_square (fp.rs:625-631: a^((p-1)/2) in {0, 1}) as a Jacobi symbol: the binary algorithm with a FIXED trip count and no
// lane-dependent branches (every lane of the wavefront runs the same steps: at most 508, in blocks of 4 until all lanes are done), ~7 cheap instructions per live limb and step against
// the ~320-product power.  Invariant: n odd, 0 <= a; each step either halves an even a (sign flips when n = 3, 5 mod 8) or,
// for odd a, orders the pair (quadratic reciprocity: flip when both are 3 mod 4), subtracts and halves.  bits(a) + bits(n)
// drops every step, so 2 * 254 steps always reach a = 0 with n = gcd.  The Montgomery factor R = (2^128)^2 is a square, so
// the symbol of the Montgomery representative is the symbol of the value.  (0 / p) = 0 counts as a square, like the reference.
// d = x - y on L limbs, returns the borrow as a mask (all ones when x < y): one v_sub_co / v_subb_co chain (written from 64-bit
// arithmetic the compiler emits four instructions per limb, one of them a quarter-rate 64-bit add)
template <int L> BN_DEV u32 sub_borrow(u32 (&d)[L], const u32 (&x)[8], const u32 (&y)[8]);
template <> BN_DEV u32 sub_borrow<1>(u32 (&d)[1], const u32 (&x)[8], const u32 (&y)[8]) {
  u32 b;
  asm("v_sub_co_u32 %0, vcc, %2, %3\n\t"
      "v_cndmask_b32 %1, 0, -1, vcc"
      : "=&v"(d[0]), "=&v"(b)
      : "v"(x[0]), "v"(y[0])
      : "vcc");
  return b;
}
template <> BN_DEV u32 sub_borrow<2>(u32 (&d)[2], const u32 (&x)[8], const u32 (&y)[8]) {
  u32 b;
  asm("v_sub_co_u32 %0, vcc, %3, %5\n\t"
      "v_subb_co_u32 %1, vcc, %4, %6, vcc\n\t"
      "v_cndmask_b32 %2, 0, -1, vcc"
      : "=&v"(d[0]), "=&v"(d[1]), "=&v"(b)
      : "v"(x[0]), "v"(x[1]), "v"(y[0]), "v"(y[1])
      : "vcc");
  return b;
}
template <> BN_DEV u32 sub_borrow<3>(u32 (&d)[3], const u32 (&x)[8], const u32 (&y)[8]) {
  u32 b;
  asm("v_sub_co_u32 %0, vcc, %4, %7\n\t"
      "v_subb_co_u32 %1, vcc, %5, %8, vcc\n\t"
      "v_subb_co_u32 %2, vcc, %6, %9, vcc\n\t"
      "v_cndmask_b32 %3, 0, -1, vcc"
      : "=&v"(d[0]), "=&v"(d[1]), "=&v"(d[2]), "=&v"(b)
      : "v"(x[0]), "v"(x[1]), "v"(x[2]), "v"(y[0]), "v"(y[1]), "v"(y[2])
      : "vcc");
  return b;
}
template <> BN_DEV u32 sub_borrow<4>(u32 (&d)[4], const u32 (&x)[8], const u32 (&y)[8]) {
  u32 b;
  asm("v_sub_co_u32 %0, vcc, %5, %9\n\t"
      "v_subb_co_u32 %1, vcc, %6, %10, vcc\n\t"
      "v_subb_co_u32 %2, vcc, %7, %11, vcc\n\t"
      "v_subb_co_u32 %3, vcc, %8, %12, vcc\n\t"
      "v_cndmask_b32 %4, 0, -1, vcc"
      : "=&v"(d[0]), "=&v"(d[1]), "=&v"(d[2]), "=&v"(d[3]), "=&v"(b)
      : "v"(x[0]), "v"(x[1]), "v"(x[2]), "v"(x[3]), "v"(y[0]), "v"(y[1]), "v"(y[2]), "v"(y[3])
      : "vcc");
  return b;
}
template <> BN_DEV u32 sub_borrow<5>(u32 (&d)[5], const u32 (&x)[8], const u32 (&y)[8]) {
  u32 b;
  asm("v_sub_co_u32 %0, vcc, %6, %11\n\t"
      "v_subb_co_u32 %1, vcc, %7, %12, vcc\n\t"
      "v_subb_co_u32 %2, vcc, %8, %13, vcc\n\t"
      "v_subb_co_u32 %3, vcc, %9, %14, vcc\n\t"
      "v_subb_co_u32 %4, vcc, %10, %15, vcc\n\t"
      "v_cndmask_b32 %5, 0, -1, vcc"
      : "=&v"(d[0]), "=&v"(d[1]), "=&v"(d[2]), "=&v"(d[3]), "=&v"(d[4]), "=&v"(b)
      : "v"(x[0]), "v"(x[1]), "v"(x[2]), "v"(x[3]), "v"(x[4]), "v"(y[0]), "v"(y[1]), "v"(y[2]), "v"(y[3]), "v"(y[4])
      : "vcc");
  return b;
}
template <> BN_DEV u32 sub_borrow<6>(u32 (&d)[6], const u32 (&x)[8], const u32 (&y)[8]) {
  u32 b;
  asm("v_sub_co_u32 %0, vcc, %7, %13\n\t"
      "v_subb_co_u32 %1, vcc, %8, %14, vcc\n\t"
      "v_subb_co_u32 %2, vcc, %9, %15, vcc\n\t"
      "v_subb_co_u32 %3, vcc, %10, %16, vcc\n\t"
      "v_subb_co_u32 %4, vcc, %11, %17, vcc\n\t"
      "v_subb_co_u32 %5, vcc, %12, %18, vcc\n\t"
      "v_cndmask_b32 %6, 0, -1, vcc"
      : "=&v"(d[0]), "=&v"(d[1]), "=&v"(d[2]), "=&v"(d[3]), "=&v"(d[4]), "=&v"(d[5]), "=&v"(b)
      : "v"(x[0]), "v"(x[1]), "v"(x[2]), "v"(x[3]), "v"(x[4]), "v"(x[5]), "v"(y[0]), "v"(y[1]), "v"(y[2]), "v"(y[3]), "v"(y[4]), "v"(y[5])
      : "vcc");
  return b;
}
template <> BN_DEV u32 sub_borrow<7>(u32 (&d)[7], const u32 (&x)[8], const u32 (&y)[8]) {
  u32 b;
  asm("v_sub_co_u32 %0, vcc, %8, %15\n\t"
      "v_subb_co_u32 %1, vcc, %9, %16, vcc\n\t"
      "v_subb_co_u32 %2, vcc, %10, %17, vcc\n\t"
      "v_subb_co_u32 %3, vcc, %11, %18, vcc\n\t"
      "v_subb_co_u32 %4, vcc, %12, %19, vcc\n\t"
      "v_subb_co_u32 %5, vcc, %13, %20, vcc\n\t"
      "v_subb_co_u32 %6, vcc, %14, %21, vcc\n\t"
      "v_cndmask_b32 %7, 0, -1, vcc"
      : "=&v"(d[0]), "=&v"(d[1]), "=&v"(d[2]), "=&v"(d[3]), "=&v"(d[4]), "=&v"(d[5]), "=&v"(d[6]), "=&v"(b)
      : "v"(x[0]), "v"(x[1]), "v"(x[2]), "v"(x[3]), "v"(x[4]), "v"(x[5]), "v"(x[6]), "v"(y[0]), "v"(y[1]), "v"(y[2]), "v"(y[3]), "v"(y[4]), "v"(y[5]), "v"(y[6])
      : "vcc");
  return b;
}
template <> BN_DEV u32 sub_borrow<8>(u32 (&d)[8], const u32 (&x)[8], const u32 (&y)[8]) {
  u32 b;
  asm("v_sub_co_u32 %0, vcc, %9, %17\n\t"
      "v_subb_co_u32 %1, vcc, %10, %18, vcc\n\t"
      "v_subb_co_u32 %2, vcc, %11, %19, vcc\n\t"
      "v_subb_co_u32 %3, vcc, %12, %20, vcc\n\t"
      "v_subb_co_u32 %4, vcc, %13, %21, vcc\n\t"
      "v_subb_co_u32 %5, vcc, %14, %22, vcc\n\t"
      "v_subb_co_u32 %6, vcc, %15, %23, vcc\n\t"
      "v_subb_co_u32 %7, vcc, %16, %24, vcc\n\t"
      "v_cndmask_b32 %8, 0, -1, vcc"
      : "=&v"(d[0]), "=&v"(d[1]), "=&v"(d[2]), "=&v"(d[3]), "=&v"(d[4]), "=&v"(d[5]), "=&v"(d[6]), "=&v"(d[7]), "=&v"(b)
      : "v"(x[0]), "v"(x[1]), "v"(x[2]), "v"(x[3]), "v"(x[4]), "v"(x[5]), "v"(x[6]), "v"(x[7]), "v"(y[0]), "v"(y[1]), "v"(y[2]), "v"(y[3]), "v"(y[4]), "v"(y[5]), "v"(y[6]), "v"(y[7])
      : "vcc");
  return b;
}
// Four steps on the low L limbs (all lanes of the wavefront have a, n < 2^(32 L) where a is still non-zero).  Both differences are
// formed (two independent borrow chains) and the non-negative one selected: ~7 single-rate instructions per limb and step.
template <int L>
BN_DEV void jacobi_steps4(u32 (&a)[8], u32 (&n)[8], u32& t) {
#pragma unroll 2
  for (int it = 0; it < 4; ++it) {
    const u32 odd = 0u - (a[0] & 1u);
    u32 d1[L], d2[L];
    const u32 lt = sub_borrow<L>(d1, a, n);        // a - n, lt = (a < n)
    (void)sub_borrow<L>(d2, n, a);                 // n - a
    const u32 sw = odd & lt;                       // odd a below n: continue with (n - a, a)
    t ^= sw & (a[0] >> 1) & (n[0] >> 1);          // both 3 mod 4
#pragma unroll
    for (int i = 0; i < L; ++i) {
      const u32 nd = lt ? d2[i] : d1[i];           // |a - n|
      n[i] = sw ? a[i] : n[i];
      a[i] = odd ? nd : a[i];
    }
    t ^= (n[0] >> 1) ^ (n[0] >> 2);               // halving an even value: (2 / n); irrelevant once a = 0 (then n = 1 or the symbol is 0)
#pragma unroll
    for (int i = 0; i + 1 < L; ++i) a[i] = (a[i] >> 1) | (a[i + 1] << 31);
    a[L - 1] >>= 1;
  }
}
// blocks of 4 steps on L limbs while some lane with a != 0 still has a or n at least 2^(32 (L - 1)); neither value ever grows, so
// once the top limb is clear for the whole wavefront the remaining steps run on L - 1 limbs (the operands lose ~1.4 bits per step:
// the average step works on half the limbs)
template <int L>
BN_DEV void jacobi_level(u32 (&a)[8], u32 (&n)[8], u32& t, int& blk) {
#pragma unroll 1
  while (blk < 508 / 4) {
    u32 nz = 0;
#pragma unroll
    for (int i = 0; i < L; ++i) nz |= a[i];
    const bool busy = nz != 0u && (L == 1 || (a[L - 1] | n[L - 1]) != 0u);
    if (!__any(busy)) break;
    jacobi_steps4<L>(a, n, t);
    ++blk;
  }
}
BN_NOINLINE bool fp_is_square(Fp x) {
  u32 a[8], n[8] = {BN_P0, BN_P1, BN_P2, BN_P3, BN_P4, BN_P5, BN_P6, BN_P7};
#pragma unroll
  for (int i = 0; i < 8; ++i) a[i] = x.v[i];
  u32 t = 0;
  int blk = 0;
  // every lane whose a has reached 0 only idles (t and n are final); random inputs need ~360 steps, the slowest of 64 lanes ~380; the
  // 508 bound is for the worst case
  jacobi_level<8>(a, n, t, blk); jacobi_level<7>(a, n, t, blk); jacobi_level<6>(a, n, t, blk); jacobi_level<5>(a, n, t, blk);
  jacobi_level<4>(a, n, t, blk); jacobi_level<3>(a, n, t, blk); jacobi_level<2>(a, n, t, blk); jacobi_level<1>(a, n, t, blk);
  const bool n_is_one = (n[0] == 1u) && ((n[1] | n[2] | n[3] | n[4] | n[5] | n[6] | n[7]) == 0u);
  return !(n_is_one && (t & 1u));
}
BN_DEV u32 fp_sgn0(const Fp& a) { return fp_from_mont(a).v[0] & 1; }   // fp.rs:636-644

// svdw.rs:180-262 (RFC 9380 6.6.1, straight line) with A = 0, B = 3, Z = 1, in two parts around the inversion tv3 = inv0(tv1 tv2):
// svdw_front (tv1, tv2 and the value to invert) and svdw_back (everything after it).  Returns false where the reference would return
// MapError (cannot happen for this curve).
struct SvdwHalf { Fp u, tv1, tv2, d; };
BN_DEV SvdwHalf svdw_front(const Fp& u) {
  const Fp one = fp_one();
  const Fp t = fp_mul(fp_mul(u, u), fp_const(C_SVDW[0]));
  SvdwHalf h;
  h.u = u;
  h.tv2 = fp_add(one, t);
  h.tv1 = fp_sub(one, t);
  h.d = fp_mul(h.tv1, h.tv2);
  return h;
}
BN_NOINLINE bool svdw_back(Fp& xo, Fp& yo, Fp u, Fp tv1, Fp tv2, Fp tv3) {
  const Fp c2 = fp_const(C_SVDW[1]), c3 = fp_const(C_SVDW[2]), c4 = fp_const(C_SVDW[3]), z = fp_const(C_SVDW[4]);
  const Fp b = fp_small(3);
  Fp tv4 = fp_mul(fp_mul(fp_mul(u, tv1), tv3), c3);
  Fp x1 = fp_sub(c2, tv4);
  Fp gx1 = fp_add(fp_mul(fp_mul(x1, x1), x1), b);
  const bool e1 = fp_is_square(gx1);
  Fp x2 = fp_add(c2, tv4);
  Fp gx2 = fp_add(fp_mul(fp_mul(x2, x2), x2), b);
  const bool e2 = fp_is_square(gx2) && !e1;
  Fp x3 = fp_mul(fp_mul(tv2, tv2), tv3);
  x3 = fp_mul(fp_mul(x3, x3), c4);
  x3 = fp_add(x3, z);
  Fp x = fp_select(x3, x1, e1);
  x = fp_select(x, x2, e2);
  Fp gx = fp_add(fp_mul(fp_mul(x, x), x), b);
  Fp y = fp_mul(gx, fp_pow_pm3_quarter(gx));     // gx^((p+1)/4), the reference's square-root candidate (fp.rs:611-616)
  bool ok = fp_eq(fp_mul(y, y), gx);
  bool e3 = fp_sgn0(u) == fp_sgn0(y);
  y = fp_select(fp_neg(y), y, e3);
  xo = x;
  yo = y;
  return ok;
}
BN_NOINLINE bool svdw_map(Fp& xo, Fp& yo, Fp u) {
  const SvdwHalf h = svdw_front(u);
  return svdw_back(xo, yo, h.u, h.tv1, h.tv2, fp_inv(h.d));
}

// The two maps of one hash side by side up to their inversion, which they SHARE (Montgomery's trick: one safegcd + three products
// instead of two safegcd; inv0 semantics kept: a zero operand is replaced by one going in and yields zero coming out), then the rest of
// each map.  Same values as svdw_map(u0), svdw_map(u1) -- the inverse of a field element is unique.
BN_DEV bool svdw_map2(Fp& x0, Fp& y0, Fp& x1, Fp& y1, const Fp& u0, const Fp& u1) {
  const SvdwHalf a = svdw_front(u0), b = svdw_front(u1);
  const bool za = fp_is_zero(a.d), zb = fp_is_zero(b.d);
  const Fp one = fp_one(), zero = fp_zero();
  const Fp da = fp_select(a.d, one, za), db = fp_select(b.d, one, zb);
  const Fp t = fp_inv(fp_mul(da, db));
  const Fp ia = fp_select(fp_mul(t, db), zero, za), ib = fp_select(fp_mul(t, da), zero, zb);
  bool ok = svdw_back(x0, y0, a.u, a.tv1, a.tv2, ia);
  ok = svdw_back(x1, y1, b.u, b.tv1, b.tv2, ib) && ok;
  return ok;
}

// g1.rs:307-331: map(u0) + map(u1) with the complete projective addition; projective result
__device__ inline bool hash_to_g1(G1P& out, const uint8_t* msg, size_t msg_len, const DstPrime& dp) {
  uint8_t em[96];
  expand_message_xmd96(em, msg, msg_len, dp);
  Fp u0 = fp_from_be48(em), u1 = fp_from_be48(em + 48);
  Fp x0, y0, x1, y1;
  bool ok = svdw_map2(x0, y0, x1, y1, u0, u1);
  G1P a{x0, y0, fp_one()}, b{x1, y1, fp_one()};
  out = g1_add(a, b);
  return ok;
}

}  // namespace bn254
