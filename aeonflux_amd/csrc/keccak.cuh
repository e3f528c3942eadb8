// Keccak-f[1600] for gfx950: 25 x 64-bit lanes in registers, one sponge per GPU lane.
// Underlies the STROBE-128 / merlin transcript hashing the reference reaches through zkp::Transcript
// [3P] (/root/reference/src/nizk/presentation.rs:355, encryption.rs:160, issuance.rs:142).
//
// Written on 32-bit halves: a 64-bit rotation is two v_alignbit_b32 (none when the amount is 32), chi and the three-input
// xors of theta are one v_bitop3_b32 per half.  Left to the compiler as uint64_t shifts, the same round came out 3.5 times longer
// (64-bit shift/add and multiply forms of the rotations plus ~260 register moves per round).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

__device__ __constant__ const uint64_t KECCAK_RC[24] = {
  0x0000000000000001ULL, 0x0000000000008082ULL, 0x800000000000808aULL, 0x8000000080008000ULL, 0x000000000000808bULL,
  0x0000000080000001ULL, 0x8000000080008081ULL, 0x8000000000008009ULL, 0x000000000000008aULL, 0x0000000000000088ULL,
  0x0000000080008009ULL, 0x000000008000000aULL, 0x000000008000808bULL, 0x800000000000008bULL, 0x8000000000008089ULL,
  0x8000000000008003ULL, 0x8000000000008002ULL, 0x8000000000000080ULL, 0x000000000000800aULL, 0x800000008000000aULL,
  0x8000000080008081ULL, 0x8000000000008080ULL, 0x0000000080000001ULL, 0x8000000080008008ULL };

// low 32 bits of ((hi:lo) >> s), 0 < s < 32
__device__ __forceinline__ uint32_t kk_align(uint32_t hi, uint32_t lo, int s) {
#if defined(__HIPCC__)
  return __builtin_amdgcn_alignbit(hi, lo, (uint32_t)s);
#else
  return (uint32_t)((((uint64_t)hi << 32) | lo) >> s);
#endif
}
struct kk_lane { uint32_t lo, hi; };
template <int N>
__device__ __forceinline__ kk_lane kk_rotl(const kk_lane& x) {
  kk_lane r;
  if constexpr (N == 0) { r = x; }
  else if constexpr (N == 32) { r.lo = x.hi; r.hi = x.lo; }
  else if constexpr (N < 32) { r.lo = kk_align(x.lo, x.hi, 32 - N); r.hi = kk_align(x.hi, x.lo, 32 - N); }
  else { r.lo = kk_align(x.hi, x.lo, 64 - N); r.hi = kk_align(x.lo, x.hi, 64 - N); }
  return r;
}
__device__ __forceinline__ kk_lane kk_xor(const kk_lane& a, const kk_lane& b) { return { a.lo ^ b.lo, a.hi ^ b.hi }; }
// a ^ b ^ c: one v_bitop3_b32 per half (gfx950; truth table 0x96)
__device__ __forceinline__ uint32_t kk_xor3w(uint32_t a, uint32_t b, uint32_t c) {
#if defined(__HIPCC__)
  return __builtin_amdgcn_bitop3_b32(a, b, c, 0x96);
#else
  return a ^ b ^ c;
#endif
}
__device__ __forceinline__ kk_lane kk_xor3(const kk_lane& a, const kk_lane& b, const kk_lane& c) { return { kk_xor3w(a.lo, b.lo, c.lo), kk_xor3w(a.hi, b.hi, c.hi) }; }
// a ^ (~b & c)
__device__ __forceinline__ kk_lane kk_chi(const kk_lane& a, const kk_lane& b, const kk_lane& c) {
  return { a.lo ^ (~b.lo & c.lo), a.hi ^ (~b.hi & c.hi) };
}

// state index = x + 5*y
__device__ __forceinline__ void keccak_f1600(uint64_t st[25]) {
  kk_lane a[25];
#pragma unroll
  for (int i = 0; i < 25; i++) { a[i].lo = (uint32_t)st[i]; a[i].hi = (uint32_t)(st[i] >> 32); }
#pragma unroll 1
  for (int round = 0; round < 24; round++) {
    // theta on three-input xors: the column parities (2 per half), then each lane ^= parity(x-1) ^ rotl(parity(x+1), 1) in one
    kk_lane c[5], c1[5];
#pragma unroll
    for (int x = 0; x < 5; x++) c[x] = kk_xor3(kk_xor3(a[x], a[x + 5], a[x + 10]), a[x + 15], a[x + 20]);
#pragma unroll
    for (int x = 0; x < 5; x++) c1[x] = kk_rotl<1>(c[x]);
#pragma unroll
    for (int i = 0; i < 25; i++) a[i] = kk_xor3(a[i], c[(i % 5 + 4) % 5], c1[(i % 5 + 1) % 5]);
    // rho + pi: b[y + 5*((2x+3y)%5)] = rotl(a[x+5y], r[x][y])
    kk_lane b[25];
    b[0] = a[0];
    b[10] = kk_rotl<1>(a[1]);   b[20] = kk_rotl<62>(a[2]);  b[5] = kk_rotl<28>(a[3]);   b[15] = kk_rotl<27>(a[4]);
    b[16] = kk_rotl<36>(a[5]);  b[1] = kk_rotl<44>(a[6]);   b[11] = kk_rotl<6>(a[7]);   b[21] = kk_rotl<55>(a[8]);  b[6] = kk_rotl<20>(a[9]);
    b[7] = kk_rotl<3>(a[10]);   b[17] = kk_rotl<10>(a[11]); b[2] = kk_rotl<43>(a[12]);  b[12] = kk_rotl<25>(a[13]); b[22] = kk_rotl<39>(a[14]);
    b[23] = kk_rotl<41>(a[15]); b[8] = kk_rotl<45>(a[16]);  b[18] = kk_rotl<15>(a[17]); b[3] = kk_rotl<21>(a[18]);  b[13] = kk_rotl<8>(a[19]);
    b[14] = kk_rotl<18>(a[20]); b[24] = kk_rotl<2>(a[21]);  b[9] = kk_rotl<61>(a[22]);  b[19] = kk_rotl<56>(a[23]); b[4] = kk_rotl<14>(a[24]);
#pragma unroll
    for (int y = 0; y < 25; y += 5) {
#pragma unroll
      for (int x = 0; x < 5; x++) a[y + x] = kk_chi(b[y + x], b[y + (x + 1) % 5], b[y + (x + 2) % 5]);
    }
    const uint64_t rc = KECCAK_RC[round];
    a[0].lo ^= (uint32_t)rc;
    a[0].hi ^= (uint32_t)(rc >> 32);
  }
#pragma unroll
  for (int i = 0; i < 25; i++) st[i] = (uint64_t)a[i].lo | ((uint64_t)a[i].hi << 32);
}
