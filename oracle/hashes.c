/* ORACLE — test infrastructure only (see afx_oracle_internal.h header).
 *
 * SHA-512 (FIPS 180-4; the reference's `sha2::Sha512` [3P] at src/symmetric.rs:138-139,202-204),
 * Keccak-f[1600] (FIPS 202), STROBE-128 and merlin `Transcript` [3P, via zkp::Transcript at
 * src/nizk/presentation.rs:355, encryption.rs:160, issuance.rs:142], restated per SURVEY.md App. A.1.
 */
#include "afx_oracle_internal.h"

/* ---------------- SHA-512 ---------------- */
static const uint64_t K512[80] = {
  0x428a2f98d728ae22ULL, 0x7137449123ef65cdULL, 0xb5c0fbcfec4d3b2fULL, 0xe9b5dba58189dbbcULL, 0x3956c25bf348b538ULL,
  0x59f111f1b605d019ULL, 0x923f82a4af194f9bULL, 0xab1c5ed5da6d8118ULL, 0xd807aa98a3030242ULL, 0x12835b0145706fbeULL,
  0x243185be4ee4b28cULL, 0x550c7dc3d5ffb4e2ULL, 0x72be5d74f27b896fULL, 0x80deb1fe3b1696b1ULL, 0x9bdc06a725c71235ULL,
  0xc19bf174cf692694ULL, 0xe49b69c19ef14ad2ULL, 0xefbe4786384f25e3ULL, 0x0fc19dc68b8cd5b5ULL, 0x240ca1cc77ac9c65ULL,
  0x2de92c6f592b0275ULL, 0x4a7484aa6ea6e483ULL, 0x5cb0a9dcbd41fbd4ULL, 0x76f988da831153b5ULL, 0x983e5152ee66dfabULL,
  0xa831c66d2db43210ULL, 0xb00327c898fb213fULL, 0xbf597fc7beef0ee4ULL, 0xc6e00bf33da88fc2ULL, 0xd5a79147930aa725ULL,
  0x06ca6351e003826fULL, 0x142929670a0e6e70ULL, 0x27b70a8546d22ffcULL, 0x2e1b21385c26c926ULL, 0x4d2c6dfc5ac42aedULL,
  0x53380d139d95b3dfULL, 0x650a73548baf63deULL, 0x766a0abb3c77b2a8ULL, 0x81c2c92e47edaee6ULL, 0x92722c851482353bULL,
  0xa2bfe8a14cf10364ULL, 0xa81a664bbc423001ULL, 0xc24b8b70d0f89791ULL, 0xc76c51a30654be30ULL, 0xd192e819d6ef5218ULL,
  0xd69906245565a910ULL, 0xf40e35855771202aULL, 0x106aa07032bbd1b8ULL, 0x19a4c116b8d2d0c8ULL, 0x1e376c085141ab53ULL,
  0x2748774cdf8eeb99ULL, 0x34b0bcb5e19b48a8ULL, 0x391c0cb3c5c95a63ULL, 0x4ed8aa4ae3418acbULL, 0x5b9cca4f7763e373ULL,
  0x682e6ff3d6b2b8a3ULL, 0x748f82ee5defb2fcULL, 0x78a5636f43172f60ULL, 0x84c87814a1f0ab72ULL, 0x8cc702081a6439ecULL,
  0x90befffa23631e28ULL, 0xa4506cebde82bde9ULL, 0xbef9a3f7b2c67915ULL, 0xc67178f2e372532bULL, 0xca273eceea26619cULL,
  0xd186b8c721c0c207ULL, 0xeada7dd6cde0eb1eULL, 0xf57d4f7fee6ed178ULL, 0x06f067aa72176fbaULL, 0x0a637dc5a2c898a6ULL,
  0x113f9804bef90daeULL, 0x1b710b35131c471bULL, 0x28db77f523047d84ULL, 0x32caab7b40c72493ULL, 0x3c9ebe0a15c9bebcULL,
  0x431d67c49c100d4cULL, 0x4cc5d4becb3e42b6ULL, 0x597f299cfc657e2aULL, 0x5fcb6fab3ad6faecULL, 0x6c44198c4a475817ULL };

#define ROR64(x, n) (((x) >> (n)) | ((x) << (64 - (n))))

static void sha512_block(uint64_t h[8], const uint8_t* p) {
  uint64_t w[80];
  for (int i = 0; i < 16; i++) {
    w[i] = 0;
    for (int j = 0; j < 8; j++) w[i] = (w[i] << 8) | p[8 * i + j];
  }
  for (int i = 16; i < 80; i++) {
    uint64_t s0 = ROR64(w[i - 15], 1) ^ ROR64(w[i - 15], 8) ^ (w[i - 15] >> 7);
    uint64_t s1 = ROR64(w[i - 2], 19) ^ ROR64(w[i - 2], 61) ^ (w[i - 2] >> 6);
    w[i] = w[i - 16] + s0 + w[i - 7] + s1;
  }
  uint64_t a = h[0], b = h[1], c = h[2], d = h[3], e = h[4], f = h[5], g = h[6], hh = h[7];
  for (int i = 0; i < 80; i++) {
    uint64_t S1 = ROR64(e, 14) ^ ROR64(e, 18) ^ ROR64(e, 41);
    uint64_t ch = (e & f) ^ (~e & g);
    uint64_t t1 = hh + S1 + ch + K512[i] + w[i];
    uint64_t S0 = ROR64(a, 28) ^ ROR64(a, 34) ^ ROR64(a, 39);
    uint64_t mj = (a & b) ^ (a & c) ^ (b & c);
    uint64_t t2 = S0 + mj;
    hh = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + t2;
  }
  h[0] += a; h[1] += b; h[2] += c; h[3] += d; h[4] += e; h[5] += f; h[6] += g; h[7] += hh;
}

void afxo_sha512(uint8_t out[64], const uint8_t* msg, size_t len) {
  uint64_t h[8] = { 0x6a09e667f3bcc908ULL, 0xbb67ae8584caa73bULL, 0x3c6ef372fe94f82bULL, 0xa54ff53a5f1d36f1ULL,
                    0x510e527fade682d1ULL, 0x9b05688c2b3e6c1fULL, 0x1f83d9abfb41bd6bULL, 0x5be0cd19137e2179ULL };
  size_t off = 0;
  for (; off + 128 <= len; off += 128) sha512_block(h, msg + off);
  uint8_t last[256];
  size_t rem = len - off;
  memset(last, 0, sizeof last);
  memcpy(last, msg + off, rem);
  last[rem] = 0x80;
  size_t total = (rem + 1 + 16 <= 128) ? 128 : 256;
  uint64_t bits = (uint64_t)len * 8;
  for (int i = 0; i < 8; i++) last[total - 1 - i] = (uint8_t)(bits >> (8 * i));
  sha512_block(h, last);
  if (total == 256) sha512_block(h, last + 128);
  for (int i = 0; i < 8; i++)
    for (int j = 0; j < 8; j++) out[8 * i + j] = (uint8_t)(h[i] >> (56 - 8 * j));
}

/* ---------------- Keccak-f[1600] ---------------- */
static const uint64_t KRC[24] = {
  0x0000000000000001ULL, 0x0000000000008082ULL, 0x800000000000808aULL, 0x8000000080008000ULL, 0x000000000000808bULL,
  0x0000000080000001ULL, 0x8000000080008081ULL, 0x8000000000008009ULL, 0x000000000000008aULL, 0x0000000000000088ULL,
  0x0000000080008009ULL, 0x000000008000000aULL, 0x000000008000808bULL, 0x800000000000008bULL, 0x8000000000008089ULL,
  0x8000000000008003ULL, 0x8000000000008002ULL, 0x8000000000000080ULL, 0x000000000000800aULL, 0x800000008000000aULL,
  0x8000000080008081ULL, 0x8000000000008080ULL, 0x0000000080000001ULL, 0x8000000080008008ULL };
static const int KROT[24] = { 1, 3, 6, 10, 15, 21, 28, 36, 45, 55, 2, 14, 27, 41, 56, 8, 25, 43, 62, 18, 39, 61, 20, 44 };
static const int KPIL[24] = { 10, 7, 11, 17, 18, 3, 5, 16, 8, 21, 24, 4, 15, 23, 19, 13, 12, 2, 20, 14, 22, 9, 6, 1 };

#define ROL64(x, n) (((x) << (n)) | ((x) >> (64 - (n))))

void keccak_f1600(uint8_t st8[200]) {
  uint64_t st[25], bc[5];
  for (int i = 0; i < 25; i++) {
    st[i] = 0;
    for (int j = 0; j < 8; j++) st[i] |= (uint64_t)st8[8 * i + j] << (8 * j);
  }
  for (int round = 0; round < 24; round++) {
    for (int i = 0; i < 5; i++) bc[i] = st[i] ^ st[i + 5] ^ st[i + 10] ^ st[i + 15] ^ st[i + 20];
    for (int i = 0; i < 5; i++) {
      uint64_t t = bc[(i + 4) % 5] ^ ROL64(bc[(i + 1) % 5], 1);
      for (int j = 0; j < 25; j += 5) st[j + i] ^= t;
    }
    uint64_t t = st[1];
    for (int i = 0; i < 24; i++) {
      int j = KPIL[i];
      uint64_t b = st[j];
      st[j] = ROL64(t, KROT[i]);
      t = b;
    }
    for (int j = 0; j < 25; j += 5) {
      for (int i = 0; i < 5; i++) bc[i] = st[j + i];
      for (int i = 0; i < 5; i++) st[j + i] ^= (~bc[(i + 1) % 5]) & bc[(i + 2) % 5];
    }
    st[0] ^= KRC[round];
  }
  for (int i = 0; i < 25; i++)
    for (int j = 0; j < 8; j++) st8[8 * i + j] = (uint8_t)(st[i] >> (8 * j));
}

/* ---------------- STROBE-128 (merlin's subset) ---------------- */
#define STROBE_R 166
#define FLAG_I 1
#define FLAG_A 2
#define FLAG_C 4
#define FLAG_T 8
#define FLAG_M 16
#define FLAG_K 32

static void strobe_run_f(strobe128* s) {
  s->st[s->pos] ^= s->pos_begin;
  s->st[s->pos + 1] ^= 0x04;
  s->st[STROBE_R + 1] ^= 0x80;
  keccak_f1600(s->st);
  s->pos = 0;
  s->pos_begin = 0;
}
static void strobe_absorb(strobe128* s, const uint8_t* d, size_t n) {
  for (size_t i = 0; i < n; i++) {
    s->st[s->pos] ^= d[i];
    s->pos++;
    if (s->pos == STROBE_R) strobe_run_f(s);
  }
}
static void strobe_overwrite(strobe128* s, const uint8_t* d, size_t n) {
  for (size_t i = 0; i < n; i++) {
    s->st[s->pos] = d[i];
    s->pos++;
    if (s->pos == STROBE_R) strobe_run_f(s);
  }
}
static void strobe_squeeze(strobe128* s, uint8_t* d, size_t n) {
  for (size_t i = 0; i < n; i++) {
    d[i] = s->st[s->pos];
    s->st[s->pos] = 0;
    s->pos++;
    if (s->pos == STROBE_R) strobe_run_f(s);
  }
}
static void strobe_begin_op(strobe128* s, uint8_t flags, int more) {
  if (more) return; /* caller guarantees same flags */
  uint8_t old_begin = s->pos_begin;
  s->pos_begin = s->pos + 1;
  s->cur_flags = flags;
  uint8_t hdr[2] = { old_begin, flags };
  strobe_absorb(s, hdr, 2);
  int force_f = (flags & (FLAG_C | FLAG_K)) != 0;
  if (force_f && s->pos != 0) strobe_run_f(s);
}
void strobe_new(strobe128* s, const uint8_t* label, size_t len) {
  memset(s, 0, sizeof *s);
  static const uint8_t init[6] = { 1, STROBE_R + 2, 1, 0, 1, 96 };
  memcpy(s->st, init, 6);
  memcpy(s->st + 6, "STROBEv1.0.2", 12);
  keccak_f1600(s->st);
  strobe_meta_ad(s, label, len, 0);
}
void strobe_meta_ad(strobe128* s, const uint8_t* d, size_t n, int more) {
  strobe_begin_op(s, FLAG_M | FLAG_A, more);
  strobe_absorb(s, d, n);
}
void strobe_ad(strobe128* s, const uint8_t* d, size_t n, int more) {
  strobe_begin_op(s, FLAG_A, more);
  strobe_absorb(s, d, n);
}
void strobe_prf(strobe128* s, uint8_t* d, size_t n, int more) {
  strobe_begin_op(s, FLAG_I | FLAG_A | FLAG_C, more);
  strobe_squeeze(s, d, n);
}
void strobe_key(strobe128* s, const uint8_t* d, size_t n, int more) {
  strobe_begin_op(s, FLAG_A | FLAG_C, more);
  strobe_overwrite(s, d, n);
}

/* ---------------- merlin ---------------- */
static void le32(uint8_t b[4], uint32_t x) { b[0] = (uint8_t)x; b[1] = (uint8_t)(x >> 8); b[2] = (uint8_t)(x >> 16); b[3] = (uint8_t)(x >> 24); }

void merlin_new(merlin_transcript* t, const uint8_t* label, size_t len) {
  strobe_new(&t->s, (const uint8_t*)"Merlin v1.0", 11);
  merlin_append_message(t, (const uint8_t*)"dom-sep", 7, label, len);
}
void merlin_append_message(merlin_transcript* t, const uint8_t* label, size_t llen, const uint8_t* msg, size_t mlen) {
  uint8_t l4[4];
  le32(l4, (uint32_t)mlen);
  strobe_meta_ad(&t->s, label, llen, 0);
  strobe_meta_ad(&t->s, l4, 4, 1);
  strobe_ad(&t->s, msg, mlen, 0);
}
void merlin_challenge_bytes(merlin_transcript* t, const uint8_t* label, size_t llen, uint8_t* dest, size_t dlen) {
  uint8_t l4[4];
  le32(l4, (uint32_t)dlen);
  strobe_meta_ad(&t->s, label, llen, 0);
  strobe_meta_ad(&t->s, l4, 4, 1);
  strobe_prf(&t->s, dest, dlen, 0);
}
