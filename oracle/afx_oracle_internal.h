/*
 * ORACLE — TEST INFRASTRUCTURE ONLY.  Not part of the shipped product.
 *
 * CPU restatement (plain C, gcc, unsigned __int128) of the arithmetic that aeonflux's hot path
 * bottoms out in.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library, and only as the checker / timed CPU baseline — never as a product path.
 *
 * PARITY UNPINNED (by the reference): /root/reference holds no golden vector or known-answer
 * test for this path (all 21 tests are thread_rng() round-trips) and the crate
 * cannot be built here (Rust nightly, un-vendored deps).  The arithmetic lives in third-party
 * crates absent from /root/reference: curve25519-dalek ^2 (Cargo.toml:34), zkp ^0.7
 * (Cargo.toml:40), merlin ^2 (transitive), sha2 ^0.8 (Cargo.toml:37); no Cargo.lock.  Their
 * published algorithms are restated here (RFC 9496 ristretto255, RFC 8032 field/scalar,
 * STROBE-128/merlin, FIPS 180-4/202) and pinned by third-party KATs + a libsodium 1.0.18
 * cross-check (tests/golden/, tests/gen_golden.py).
 */
#ifndef AFX_ORACLE_INTERNAL_H
#define AFX_ORACLE_INTERNAL_H

#include <stddef.h>
#include <stdint.h>
#include <string.h>

typedef unsigned __int128 u128;

/* ---- GF(2^255-19), 5 x 51-bit limbs (same class as dalek's u64_backend) ---- */
typedef struct { uint64_t v[5]; } fe;

void fe_0(fe* h);
void fe_1(fe* h);
void fe_copy(fe* h, const fe* f);
void fe_add(fe* h, const fe* f, const fe* g);
void fe_sub(fe* h, const fe* f, const fe* g);
void fe_neg(fe* h, const fe* f);
void fe_mul(fe* h, const fe* f, const fe* g);
void fe_sq(fe* h, const fe* f);
void fe_sqn(fe* h, const fe* f, int n);
void fe_frombytes(fe* h, const uint8_t s[32]);   /* ignores bit 255 (dalek FieldElement::from_bytes) */
void fe_tobytes(uint8_t s[32], const fe* h);     /* canonical */
void fe_invert(fe* out, const fe* z);
void fe_pow22523(fe* out, const fe* z);
int  fe_is_negative(const fe* f);                /* LSB of canonical encoding */
int  fe_is_zero(const fe* f);
int  fe_eq(const fe* f, const fe* g);
void fe_cmov(fe* f, const fe* g, int b);         /* f = b ? g : f */
void fe_cneg(fe* f, int b);                      /* f = b ? -f : f */
void fe_abs(fe* h, const fe* f);
/* dalek FieldElement::sqrt_ratio_i  (RFC 9496 §4.2 SQRT_RATIO_M1); returns was_square */
int  fe_sqrt_ratio_i(fe* r, const fe* u, const fe* v);

extern fe FE_D, FE_D2, FE_SQRT_M1, FE_SQRT_AD_MINUS_ONE, FE_INVSQRT_A_MINUS_D, FE_ONE_MINUS_D_SQ, FE_D_MINUS_ONE_SQ;
void afxo_init_constants(void);

/* ---- scalars mod l = 2^252 + 27742317777372353535851937790883648493, 32-byte LE canonical ---- */
typedef struct { uint8_t b[32]; } sc;
void sc_reduce_wide(sc* r, const uint8_t in[64]);       /* Scalar::from_bytes_mod_order_wide */
void sc_from_bytes_mod_order(sc* r, const uint8_t in[32]);
int  sc_is_canonical(const uint8_t in[32]);
void sc_add(sc* r, const sc* a, const sc* b);
void sc_sub(sc* r, const sc* a, const sc* b);
void sc_neg(sc* r, const sc* a);
void sc_mul(sc* r, const sc* a, const sc* b);
void sc_muladd(sc* r, const sc* a, const sc* b, const sc* c); /* a*b + c */
void sc_one(sc* r);
void sc_zero(sc* r);
int  sc_eq(const sc* a, const sc* b);

/* ---- edwards25519 extended coordinates / ristretto255 ---- */
typedef struct { fe X, Y, Z, T; } ge;
void ge_identity(ge* h);
void ge_add(ge* r, const ge* p, const ge* q);
void ge_sub(ge* r, const ge* p, const ge* q);
void ge_neg(ge* r, const ge* p);
void ge_double(ge* r, const ge* p);
void ge_scalarmult(ge* r, const sc* s, const ge* p);                 /* fixed 4-bit window */
void ge_multiscalar_vartime(ge* r, const sc* s, const ge* p, int n); /* Straus, width-5 NAF */
void ge_multiscalar(ge* r, const sc* s, const ge* p, int n);         /* Straus, radix-16 (all digits) */
int  ristretto_decode(ge* r, const uint8_t s[32]);   /* 1 = ok (CompressedRistretto::decompress) */
void ristretto_encode(uint8_t s[32], const ge* p);   /* RistrettoPoint::compress */
void ristretto_from_uniform_bytes(ge* r, const uint8_t b[64]);
int  ristretto_eq(const ge* p, const ge* q);
void ristretto_basepoint(ge* r);

/* ---- SHA-512 ---- */
void afxo_sha512(uint8_t out[64], const uint8_t* msg, size_t len);

/* ---- Keccak-f[1600], STROBE-128, merlin ---- */
typedef struct {
  uint8_t st[200];
  uint8_t pos, pos_begin, cur_flags;
} strobe128;
void keccak_f1600(uint8_t st[200]);
void strobe_new(strobe128* s, const uint8_t* label, size_t len);
void strobe_meta_ad(strobe128* s, const uint8_t* data, size_t len, int more);
void strobe_ad(strobe128* s, const uint8_t* data, size_t len, int more);
void strobe_prf(strobe128* s, uint8_t* data, size_t len, int more);
void strobe_key(strobe128* s, const uint8_t* data, size_t len, int more);

typedef struct { strobe128 s; } merlin_transcript;
void merlin_new(merlin_transcript* t, const uint8_t* label, size_t len);
void merlin_append_message(merlin_transcript* t, const uint8_t* label, size_t llen, const uint8_t* msg, size_t mlen);
void merlin_challenge_bytes(merlin_transcript* t, const uint8_t* label, size_t llen, uint8_t* dest, size_t dlen);

/* ---- zkp 0.7 toolbox (Schnorr constraint systems, CompactProof) ---- */
#define ZKP_MAX_SCALARS 48
#define ZKP_MAX_POINTS  128
#define ZKP_MAX_CONSTRAINTS 48
#define ZKP_MAX_TERMS 72
typedef struct {
  int lhs;
  int n;
  int sc[ZKP_MAX_TERMS];
  int pt[ZKP_MAX_TERMS];
} zkp_constraint;

typedef struct {
  merlin_transcript t;
  int is_prover;
  int failed;                      /* verifier: an allocated point was the identity encoding */
  int n_scalars, n_points, n_constraints;
  sc scalars[ZKP_MAX_SCALARS];     /* prover: witness values */
  ge points[ZKP_MAX_POINTS];       /* prover: values; verifier: filled at verify time */
  uint8_t enc[ZKP_MAX_POINTS][32]; /* compressed encodings as appended to the transcript */
  const char* point_labels[ZKP_MAX_POINTS];
  zkp_constraint cs[ZKP_MAX_CONSTRAINTS];
} zkp_cs;

void zkp_init(zkp_cs* z, int is_prover, const char* transcript_label, const char* proof_label);
int  zkp_alloc_scalar(zkp_cs* z, const char* label, const sc* value /* NULL for verifier */);
int  zkp_alloc_point_prover(zkp_cs* z, const char* label, const ge* p);
int  zkp_alloc_point_verifier(zkp_cs* z, const char* label, const uint8_t enc[32]);
void zkp_constrain(zkp_cs* z, int lhs, int n, const int* scs, const int* pts);
/* rng_seed stands in for the 32 bytes zkp draws from thread_rng() inside prove_compact */
void zkp_prove_compact(zkp_cs* z, const uint8_t rng_seed[32], sc* challenge, sc* responses);
/* returns 1 = verified.  n_responses is CompactProof.responses.len() */
int  zkp_verify_compact(zkp_cs* z, const uint8_t challenge[32], const uint8_t* responses, int n_responses);
/* debugging/parity hooks: last commitments computed by prove/verify */
extern __thread int zkp_debug_ncommit;
extern __thread uint8_t zkp_debug_commit[ZKP_MAX_CONSTRAINTS][32];
extern __thread uint8_t zkp_debug_challenge[32];

#endif
