/*
 * ORACLE — TEST INFRASTRUCTURE ONLY (see afx_oracle_internal.h for the full statement).
 * Public C interface of the CPU restatement, loaded through ctypes by tests/, smoke() and
 * bench.py's cpu_baseline leg.  PARITY UNPINNED by the reference; pinned by third-party KATs.
 */
#ifndef AFX_ORACLE_H
#define AFX_ORACLE_H
#include <stddef.h>
#include <stdint.h>
#include "../include/aeonflux_gpu.h" /* shared constants and SoA batch structs */

typedef struct afxo_ctx afxo_ctx;

/* one ProofOfEncryption (src/nizk/encryption.rs:32-41) */
typedef struct {
  uint8_t challenge[32];
  uint8_t responses[6][32];
  uint8_t pk[32], E1[32], E2[32], C_y_1[32], C_y_2[32], C_y_3[32], C_y_2p[32];
  uint16_t index;
} afxo_encproof;

/* one ProofOfValidCredential (src/nizk/presentation.rs:118-127) */
typedef struct {
  uint32_t n_attributes;
  uint32_t n_responses;
  uint8_t challenge[32];
  uint8_t responses[3 + AFX_MAX_ATTRIBUTES][32];
  uint8_t C_x_0[32], C_x_1[32], C_V[32];
  uint8_t C_y[AFX_MAX_ATTRIBUTES][32];
  uint8_t kinds[AFX_MAX_ATTRIBUTES];
  uint8_t attr_values[AFX_MAX_ATTRIBUTES][32];
  uint32_t n_hidden_scalars;
  uint16_t hidden_scalar_indices[AFX_MAX_ATTRIBUTES];
  uint32_t n_enc_proofs;
  afxo_encproof enc[AFX_MAX_ATTRIBUTES];
} afxo_presentation;

size_t afxo_sizeof_system_parameters(uint32_t n);
size_t afxo_sizeof_secret_key(uint32_t n);
afxo_ctx* afxo_ctx_new(const uint8_t* params, size_t plen, const uint8_t* key, size_t klen, const uint8_t* issuer_params);
void afxo_ctx_free(afxo_ctx* c);
uint32_t afxo_ctx_n(const afxo_ctx* c);
long afxo_system_parameters_generate(uint32_t n, const uint8_t* stream, size_t stream_len, uint8_t* out);
int afxo_issuer_new(const uint8_t* params, size_t plen, const uint8_t* draws, uint8_t* key_out, uint8_t issuer_params_out[64]);
int afxo_encode_to_group(const uint8_t* data, size_t len, uint8_t out[32]);
int afxo_decode_from_group(const uint8_t pt[32], uint8_t data[30]);
int afxo_plaintext_from_bytes(const uint8_t msg[30], uint8_t out[96]);
int afxo_keypair_derive(const afxo_ctx* c, const uint8_t master_secret[64], uint8_t out[128]);
int afxo_encrypt(const uint8_t keypair[128], const uint8_t plaintext[96], uint8_t out[64]);
int afxo_decrypt(const uint8_t keypair[128], const uint8_t ciphertext[64], uint8_t plaintext_out[96]);
int afxo_issue(const afxo_ctx* c, uint32_t n_attrs, const uint8_t* kinds, const uint8_t* values, const uint8_t t_wide[64],
               const uint8_t U_wide[64], const uint8_t rng_seed[32], uint8_t out_t[32], uint8_t out_U[32], uint8_t out_V[32],
               uint8_t out_challenge[32], uint8_t* out_responses);
int afxo_issuance_verify(const afxo_ctx* c, uint32_t n_attrs, const uint8_t* kinds, const uint8_t* values, const uint8_t t[32],
                         const uint8_t U[32], const uint8_t V[32], const uint8_t challenge[32], const uint8_t* responses,
                         uint32_t n_responses);
int afxo_show(const afxo_ctx* c, uint32_t n_attrs, const uint8_t* kinds, const uint8_t* values, const uint8_t t_in[32],
              const uint8_t U_in[32], const uint8_t V_in[32], const uint8_t* keypair, const uint8_t z_wide[64],
              const uint8_t rng_seed[32], const uint8_t* enc_seeds, afxo_presentation* out);
int afxo_verify_presentation(const afxo_ctx* c, const afxo_presentation* p);
int afxo_verify_encryption_proof(const afxo_ctx* c, const afxo_encproof* e);

/* batch forms over the same SoA layout as the C ABI; `threads` host threads, static split */
int afxo_verify_presentations_soa(const afxo_ctx* c, const afx_shape* shape, const afx_presentation_soa* batch, size_t count,
                                  uint8_t* status, int threads);

/* the same with every proof's recomputed challenge (trace[(r * count + i) * 32], r = 0 main proof, 1 + e e-th proof of encryption;
 * reached[r * count + i] = that verifier got to its commitments), and Issuer::issue / CredentialIssuance::verify over
 * struct-of-arrays batches (values [n][count][32]); oracle/batch.c */
int afxo_verify_presentations_soa_traced(const afxo_ctx* c, const afx_shape* shape, const afx_presentation_soa* batch, size_t count,
                                         uint8_t* status, uint8_t* trace, uint8_t* reached, int threads);
int afxo_issue_soa(const afxo_ctx* c, uint32_t n, const uint8_t* kinds, const uint8_t* values, const uint8_t* t_wide, const uint8_t* U_wide,
                   const uint8_t* seed, size_t count, uint8_t* t, uint8_t* U, uint8_t* V, uint8_t* challenge, uint8_t* responses,
                   uint8_t* status, int threads);
int afxo_verify_issuances_soa_traced(const afxo_ctx* c, uint32_t n, const uint8_t* kinds, const uint8_t* values, const uint8_t* t,
                                     const uint8_t* U, const uint8_t* V, const uint8_t* challenge, const uint8_t* responses, uint32_t nr,
                                     size_t count, uint8_t* status, uint8_t* trace, uint8_t* reached, int threads);

/* primitive wrappers for KATs */
int afxo_point_decode_encode(const uint8_t in[32], uint8_t out[32]);
void afxo_point_from_uniform(const uint8_t in[64], uint8_t out[32]);
int afxo_point_add(const uint8_t a[32], const uint8_t b[32], uint8_t out[32]);
int afxo_point_sub(const uint8_t a[32], const uint8_t b[32], uint8_t out[32]);
int afxo_point_scalarmult(const uint8_t s[32], const uint8_t a[32], uint8_t out[32]);
void afxo_basepoint(uint8_t out[32]);
int afxo_multiscalar(uint32_t n, const uint8_t* scalars, const uint8_t* points, int vartime, uint8_t out[32]);
void afxo_scalar_reduce_wide(const uint8_t in[64], uint8_t out[32]);
void afxo_scalar_muladd(const uint8_t a[32], const uint8_t b[32], const uint8_t c[32], uint8_t out[32]);
void afxo_scalar_neg(const uint8_t a[32], uint8_t out[32]);
int afxo_scalar_is_canonical(const uint8_t a[32]);
void afxo_sha512(uint8_t out[64], const uint8_t* msg, size_t len);
void afxo_keccak_f1600(uint8_t st[200]);
void afxo_merlin_simple(const uint8_t* label, size_t llen, const uint8_t* l1, size_t l1len, const uint8_t* m1, size_t m1len,
                        const uint8_t* l2, size_t l2len, uint8_t* out, size_t outlen);
long afxo_merlin_script(const uint8_t* script, size_t len, const uint8_t* fields, uint32_t n_fields, uint8_t* out, size_t cap);
void afxo_debug_last(uint8_t* commits, int* ncommit, uint8_t challenge[32]);
void afxo_debug_reset(void);
/* opt-in strict mode (not the reference's behaviour; see oracle/aeonflux.c) */
void afxo_ctx_set_strict(afxo_ctx* c, int strict);   /* ncommit = 0 until the next prove/verify reaches its commitments */
#endif
