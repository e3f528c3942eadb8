/*
 * aeonflux_gpu.h — C ABI of the MI355X batch engine for aeonflux's credential NIZKs.
 *
 * This is the drop-in boundary: the entry points a Rust `extern "C"` shim
 *  would bind behind `Issuer::issue` (/root/reference/src/issuer.rs:111-124),
 * `Issuer::verify` (src/issuer.rs:141-147), `CredentialIssuance::verify` (src/issuer.rs:48-57)
 * and `AnonymousCredential::show` (src/credential.rs:37-46).  The reference has no FFI/plugin
 * layer of its own (Cargo.toml:22 is commented out), so the batch forms below are new; each
 * item of a batch has exactly the semantics of one call of the cited reference method.
 *
 * Conventions
 *  - Sc = 32-byte little-endian canonical scalar mod l; Pt = 32-byte compressed ristretto255.
 *  - All per-item data is struct-of-arrays: a field `f` of a batch of `count` items is one
 *    contiguous array `[count][32]`; repeated fields are `[k][count][32]` (k-major).
 *  - Pointers in the *_soa structs are HOST pointers for afx_*() calls and DEVICE pointers for
 *    the afx_*_dev() calls (inputs already resident in HBM; no PCIe in the call).
 *  - Return value: AFX_OK or a negative AFX_E_* batch-level error.  Per-item results go to
 *    status[i] (AFX_ST_*).  Inputs that would make the reference panic (out-of-range indices,
 *    length mismatches: src/nizk/presentation.rs:81,100,346,351,407) and encodings the reference
 *    cannot hold in memory (non-canonical scalars, undecodable points) give
 *    AFX_ST_VERIFICATION_FAILURE, never a fault.
 *  - The engine has NO CPU fallback: every arithmetic step runs in HIP kernels on gfx950; calls
 *    fail with AFX_E_NO_DEVICE when no GPU is usable.
 *  - Every random draw the reference makes (caller csprng and zkp's hidden thread_rng()) is an
 *    explicit input, so runs are reproducible (SURVEY.md §8b "Randomness").
 */
#ifndef AEONFLUX_GPU_H
#define AEONFLUX_GPU_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define AFX_MAX_ATTRIBUTES 32

/* batch-level return codes */
#define AFX_OK 0
#define AFX_E_BAD_ARGS (-1)    /* null pointer, bad length, n mismatch, unsupported shape        */
#define AFX_E_BAD_PARAMS (-2)  /* SystemParameters / key bytes do not parse (parameters.rs:92-153) */
#define AFX_E_NO_DEVICE (-3)   /* no usable HIP device / kernel image (there is no CPU fallback)  */
#define AFX_E_HIP (-4)         /* a HIP runtime call failed; see afx_last_error()                 */
#define AFX_E_NO_KEY (-5)      /* operation needs the issuer secret key but ctx has none          */
#define AFX_E_NO_MEMORY (-6)   /* host allocation (or thread creation) failed inside the library    */

/* per-item status == the reference's CredentialError outcome (src/errors.rs:73-89) */
#define AFX_ST_OK 0
#define AFX_ST_VERIFICATION_FAILURE 1  /* CredentialError::VerificationFailure (errors.rs:152-156) */
#define AFX_ST_MAC_CREATION 2          /* CredentialError::MacCreation (amacs.rs:285-287)          */
#define AFX_ST_NO_SYMMETRIC_KEY 3      /* CredentialError::NoSymmetricKey (presentation.rs:150-157) */
#define AFX_ST_UNDECRYPTABLE 4         /* CredentialError::UndecryptableAttribute (symmetric.rs:285-288) */

/* amacs::Attribute (src/amacs.rs:168-179): kinds of a credential's attributes */
#define AFX_ATTR_PUBLIC_SCALAR 0
#define AFX_ATTR_SECRET_SCALAR 1
#define AFX_ATTR_PUBLIC_POINT 2
#define AFX_ATTR_EITHER_POINT 3
#define AFX_ATTR_SECRET_POINT 4

/* amacs::EncryptedAttribute (src/amacs.rs:207-217): kinds as sent in a presentation */
#define AFX_ENC_PUBLIC_SCALAR 0
#define AFX_ENC_SECRET_SCALAR 1
#define AFX_ENC_PUBLIC_POINT 2
#define AFX_ENC_SECRET_POINT 3

typedef struct afx_ctx afx_ctx;

/* The batch-uniform part of a ProofOfValidCredential (src/nizk/presentation.rs:118-127): everything
 * that is not a Sc or Pt.  Heterogeneous traffic is grouped by shape on the host. */
typedef struct {
  uint32_t n_attributes;                                /* encrypted_attributes.len()                  */
  uint8_t kinds[AFX_MAX_ATTRIBUTES];                    /* AFX_ENC_* per position                      */
  uint32_t n_responses;                                 /* proof.responses.len()                       */
  uint32_t n_hidden_scalars;                            /* hidden_scalar_indices.len()                 */
  uint16_t hidden_scalar_indices[AFX_MAX_ATTRIBUTES];
  uint32_t n_enc_proofs;                                /* proofs_of_encryption.len()                  */
  uint16_t enc_indices[AFX_MAX_ATTRIBUTES];             /* ProofOfEncryption.index (encryption.rs:36)  */
} afx_shape;

/* ProofOfEncryption (src/nizk/encryption.rs:32-41), SoA over the batch */
typedef struct {
  const uint8_t* challenge;  /* [count] Sc                      proof.challenge        */
  const uint8_t* responses;  /* [6][count] Sc                   proof.responses a,a0,a1,m3,z,z1 */
  const uint8_t* pk;         /* [count] Pt                      public_key.pk          */
  const uint8_t* E1;         /* [count] Pt                      ciphertext.E1          */
  const uint8_t* E2;         /* [count] Pt                      ciphertext.E2          */
  const uint8_t* C_y_1;      /* [count] Pt */
  const uint8_t* C_y_2;      /* [count] Pt */
  const uint8_t* C_y_3;      /* [count] Pt */
  const uint8_t* C_y_2p;     /* [count] Pt                      C_y_2_prime            */
} afx_encproof_soa;

/* ProofOfValidCredential (src/nizk/presentation.rs:118-127), SoA over the batch */
typedef struct {
  const uint8_t* challenge;    /* [count] Sc                            proof.challenge               */
  const uint8_t* responses;    /* [n_responses][count] Sc               proof.responses               */
  const uint8_t* C_x_0;        /* [count] Pt */
  const uint8_t* C_x_1;        /* [count] Pt */
  const uint8_t* C_V;          /* [count] Pt */
  const uint8_t* C_y;          /* [n_attributes][count] Pt */
  const uint8_t* attr_values;  /* [n_attributes][count] 32 B: Sc for PUBLIC_SCALAR rows, Pt for
                                  PUBLIC_POINT rows; rows of secret kinds are never read            */
  const afx_encproof_soa* enc; /* [n_enc_proofs] (host array of structs, also for *_dev calls)     */
} afx_presentation_soa;

/* ---- context ------------------------------------------------------------------------------- */

/* Build an engine context on HIP device `device`.
 *   sysparams      : SystemParameters::to_bytes layout (src/parameters.rs:155-184)
 *   amacs_key      : amacs::SecretKey::to_bytes layout (src/amacs.rs:110-125), or NULL/0 for a
 *                    user-side context (show / issuance-verify only).  Every y_i is read (the
 *                    reference's from_bytes re-reads one chunk, src/amacs.rs:148-150 — a bug).
 *   issuer_params  : C_W || I (64 B; intended IssuerParameters layout, src/issuer.rs:155,163)
 * Decompresses the generators, builds the fixed-base window tables in HBM.  A ctx may be called from any number of threads
 * at once, like the `&self` methods it stands behind: small host-pointer calls that arrive while the device is busy share
 * ONE set of kernel launches (afx_ctx_set_coalescing below); everything else takes the context in turn. */
int afx_ctx_create(afx_ctx** out, int device, const uint8_t* sysparams, size_t sysparams_len,
                   const uint8_t* amacs_key, size_t amacs_key_len, const uint8_t issuer_params[64]);
/* Overwrites device and host copies of the key and tables before freeing (Zeroize+Drop on
 * amacs::SecretKey, src/amacs.rs:64-82). */
void afx_ctx_destroy(afx_ctx* ctx);
const char* afx_last_error(void);
uint32_t afx_ctx_n_attributes(const afx_ctx* ctx);
/* The HIP stream (hipStream_t) every call on this ctx launches on; for event timing in bench.py. */
void* afx_ctx_stream(const afx_ctx* ctx);

/* Cross-call pipelining (off by default).  Off: every *_dev call runs on afx_ctx_stream(ctx), strictly ordered.
 * On: successive *_dev calls alternate between two internal streams (each with its own workspace), so the launch
 * tail and small kernels of one call overlap the next call's work; the caller guarantees the calls are independent
 * (they may share read-only inputs) and waits with afx_ctx_synchronize (or a device-wide synchronise). */
int afx_ctx_set_pipelining(afx_ctx* ctx, int enable);
int afx_ctx_synchronize(afx_ctx* ctx);

/* Strict mode (opt-in, off by default, NOT bit-compatible with the reference):
 *  - constraint #3 of the presentation proof (presentation.rs:267-273 prover, :427-433 verifier) pairs the j-th kept
 *    commitment with the generators and kind of its OWN attribute position instead of position j, in afx_show and in
 *    afx_verify_presentations; with hidden group elements only in trailing positions both modes give the same bytes,
 *    otherwise only strict-mode presentations verify, and only under a strict-mode verifier;
 *  - afx_verify_presentations requires exactly one proof of encryption per SECRET_POINT attribute, with
 *    enc_indices equal to those positions in increasing order (the reference verifies whatever is attached,
 *    presentation.rs:438-440).
 *  - the DLEQ the reference's README.md:121-122 lists as TODO: for every hidden group element i the presentation proof also
 *    shows, with its own nonce z, that C_y[i] - C_y_1 = z*(G_y[i] - G_y[0]), C_y_1 being the commitment inside that
 *    attribute's proof of encryption (C_y[i] = z*G_y[i] + M1, presentation.rs:173; C_y_1 = z*G_y[0] + M1, encryption.rs:70):
 *    the plaintext that is proven encrypted is the one the credential commits to.  Prover (afx_show) and verifier. */
int afx_ctx_set_strict(afx_ctx* ctx, int enable);

/* Schedule of the issuer key's scalars (x0, x1, y_i in Issuer::verify's Z and in Amac::tag).  Default (0): the host recodes
 * them to width-5 NAF and every lane runs that addition schedule - the fastest form, whose running time depends on the
 * key's NAF weight (a per-key constant, the same for every batch under that key; nothing depends on the items).  1: the key
 * scalars take the per-item window path instead (64 additions per term, whatever the key): running time independent of
 * the key, 3-5 % slower.  The reference computes these products with dalek's constant-time `*` / `multiscalar_mul`
 * (src/nizk/presentation.rs:342-351, src/amacs.rs:267-270); DESIGN.md section 1 "Secrets" (the threat model: docs/HISTORY.md section 1).
 * Results are identical in both modes. */
int afx_ctx_set_fixed_key_schedule(afx_ctx* ctx, int enable);

/* Secret-independent addressing.  The reference multiplies by secrets - the issuer key, the prover's nonces, blindings and
 * witnesses - with dalek's constant-time `*` / `multiscalar_mul` (src/amacs.rs:267-270, src/nizk/presentation.rs:162-184, zkp's
 * Prover), whose table lookups read every entry and select.  This engine's INSTRUCTION stream never depends on a per-item secret
 * in any mode; what the mode decides is whether the window digit of a secret may pick WHICH table entry a lane gathers from HBM:
 *   AFX_SECRETS_PROVER_SIDE (2, the default of a new context): not on the prover-side calls - afx_issue*, afx_show* and the
 *     symmetric-key helpers (afx_keypairs_derive, afx_encrypt, afx_decrypt): every scalar of every multiscalar job there but the
 *     constant 1 is treated as a secret.  What a user of the crate gets from its constant-time arithmetic.  Issuer::verify runs
 *     the fast tables (its only secret is the issuer key, see afx_ctx_set_fixed_key_schedule).
 *   AFX_SECRETS_EVERYWHERE (1): additionally on afx_verify_presentations*: every scalar of the job that computes Z - the issuer
 *     key's (which then also run the fixed schedule, whatever afx_ctx_set_fixed_key_schedule says) and the per-item products
 *     y_i * m_i of the key with revealed scalar attributes, whose digits would give the key away just the same.
 *   AFX_SECRETS_NOWHERE (0): the fastest tables everywhere (rounds 1-3 of this engine; a device of the engine's own, or inputs
 *     that are no secrets: synthetic benchmark data).  Prover-side calls of up to 2048 items run the secret-independent plan in
 *     this mode too: at those sizes it is the faster one (its chains run in segments over the bases' powers).
 * Where the mode applies no load's address is made from a digit of a secret.  A secret term on a per-item base runs 2-bit signed
 * windows: every addition reads both stored (affine) entries of its lane's table and keeps the digit's with selects - 128 additions
 * per term instead of 64.  A secret term on a generator runs 6-bit signed windows over positional tables (155 KB per generator,
 * part of every context): each lane of the wave loads one of the window's 32 multiples, the one its lane id names, and every lane
 * takes the multiple its digit names from the lane that holds it (ds_bpermute_b32: a register exchange, no memory access, source
 * lanes chosen so that no pattern of digits conflicts in the crossbar) - 43 additions per term instead of 20.  Results are
 * byte-identical in every mode.  Cost against mode 0, measured on one MI355X: issue -35 %, show -24 %;
 * verification unchanged in mode 2, -8 % in mode 1. */
#define AFX_SECRETS_NOWHERE 0
#define AFX_SECRETS_EVERYWHERE 1
#define AFX_SECRETS_PROVER_SIDE 2
int afx_ctx_set_secret_independent_addressing(afx_ctx* ctx, int mode);

/* Items per internal pass (tuning; 0 restores the default of 2^19).  A batch larger than this is processed in passes
 * of this many items, which bounds the device workspace (about 25-70 KB per item and pass, depending on the
 * statement); smaller values trade throughput for memory.  Accepted range 256 .. 2^22. */
int afx_ctx_set_chunk_items(afx_ctx* ctx, uint32_t items);

/* Small passes (default 4096 items; 0 switches the latency plans off, at most 2^16).  A call of few items leaves most of the
 * device idle, and its duration is that of its LONGEST chain of field operations - one Issuer::verify of the reference is one
 * presentation (src/issuer.rs:141-147).  Passes of at most this many items therefore give every variable-base term of every
 * multiscalar multiplication a chain of its own (a 3-term commitment becomes 3 lanes with 64 additions each instead of one
 * with 192), add the partial sums up afterwards, and encode an item's commitments in several rows.  More work in all, less
 * time per call (measured, C3 shape: 3.2 ms against 4.3 ms at 2^12 items, level at 2^13).  Passes of up to FOUR times this many items split
 * only the job that multiplies by the issuer key (Z: one grid row that cannot fill the device at such sizes) into one NAF
 * chain per term (5.0 ms against 5.8 ms at 2^13 items, 8.7 against 9.0 at 2^14).  Results are identical under every plan. */
int afx_ctx_set_small_batch_items(afx_ctx* ctx, uint32_t items);

/* Concurrent small calls (on by default).  Issuer::verify takes `&self`, keeps no state and is called one presentation at a time
 * from as many threads as a server has (src/issuer.rs:141-147); Issuer::issue (:111-124), CredentialIssuance::verify (:48-57) and
 * AnonymousCredential::show (src/credential.rs:37-46) likewise.  One such call costs one chain of field operations on a device
 * that is otherwise idle (0.8 ms), so calls that queue behind each other would cap a server at ~1.2 k calls/s whatever its thread
 * count.  Instead, host-pointer calls (afx_verify_presentations[_range,_wire,_wire_range], afx_issue[_range],
 * afx_verify_issuances[_range,_wire], afx_show[_range]) of at most 512 items that arrive while another call's kernels run are
 * COLLECTED: each stages its rows - calls of one statement, shape and mode into the free item slots of ONE pass, so 64 callers of
 * one shape are one pass of 64 items - and sleeps until the flush that carries its rows completes.  The caller that opened a
 * collection launches it as soon as the device is free, at the latest `max_wait_us` after it opened it (default 2000) or when it
 * holds `max_items` items (default 4096; at most 2^16).  A call that finds the device idle is launched at once: a single thread
 * sees the latency it saw before.  Results are byte-identical to separate calls; a flush that fails (AFX_E_HIP ...) fails every call
 * it carried with the same code and message, and the context stays usable.  max_items == 0 switches collection off (every call
 * then takes the context in turn).  Large calls, device-pointer calls and the afx_ctx_set_* functions wait for the collections
 * in flight and then have the context to themselves. */
int afx_ctx_set_coalescing(afx_ctx* ctx, uint32_t max_wait_us, uint32_t max_items);
typedef struct afx_coalescing_stats {
  uint64_t sessions;        /* sets of launches that carried collected calls                                    */
  uint64_t calls;           /* calls that went through them                                                     */
  uint64_t items;           /* ... and their items                                                              */
  uint64_t appended_calls;  /* calls that took free item slots of another call's pass (no plan of their own)    */
  uint64_t max_calls;       /* most calls one set of launches carried                                           */
  uint64_t leader_waits;    /* times a collection's opener slept because an earlier one was still computing     */
  uint64_t staging_ns;      /* time the context's lock was held while calls staged their rows (sum)             */
  uint64_t launch_ns;       /* ... and while collections were launched (sum): what serialises the callers       */
} afx_coalescing_stats;
int afx_ctx_get_coalescing_stats(afx_ctx* ctx, afx_coalescing_stats* out);

/* Plan variants (test aid; 0 = every choice automatic, the default).  Which of several equivalent layouts a SMALL pass takes is
 * decided by its size and by the width of the launch set it shares with other calls: secret scalars of a prover pass in 8, 4 or 1
 * segment(s); chains on four waves per item or one; a transcript on a wave or on 32 lanes; sums of many parts with a lane per part
 * or per item.  Every combination returns the same bytes.  The flags force the alternatives at any size, so that tests can compare
 * each of them with the CPU oracle (tests/test_gpu_plan_variants.py) instead of reaching them only through particular sizes and
 * concurrency.  AFX_VARIANT_SELFCHECK: every plan is assembled twice against different provisional addresses and the two copies,
 * relocated to the same place, must be equal byte for byte (also: AFX_PLAN_SELFCHECK=1 in the environment at context creation).
 * The flags are part of every plan's cache key.  Not a tuning interface: the automatic choice is the measured best. */
#define AFX_VARIANT_SEGMENTS_1 0x01u        /* small prover passes: whole chains                                       */
#define AFX_VARIANT_SEGMENTS_2 0x02u        /* ... two segments per secret scalar                                      */
#define AFX_VARIANT_SEGMENTS_4 0x04u        /* ... four (what passes of 257 .. 2048 items take by themselves)          */
#define AFX_VARIANT_ONE_WAVE_CHAINS 0x08u   /* no four-wave chains (what launch sets wider than 512 blocks take)       */
#define AFX_VARIANT_HASH_HALF_WAVE 0x10u    /* cooperative transcripts on 32 lanes per item (more than 2048 of them)   */
#define AFX_VARIANT_NO_POINTSUM_TREE 0x20u  /* sums of many parts on one lane per item                                 */
#define AFX_VARIANT_SELFCHECK 0x40u
#define AFX_VARIANT_ALL 0x7fu
int afx_ctx_set_plan_variants(afx_ctx* ctx, uint32_t flags);

/* Host copies of LARGE host-pointer calls (default 0: off).  Issuer::verify on a batch in host memory (src/issuer.rs:141-147
 * called over a vector of presentations) hands the engine ~2.4 KB per presentation in pageable memory.  A call of more than 16 MB
 * of rows is cut into slices of 2^17 items on two streams; each slice's rows are gathered into a pinned image by `threads` host
 * threads (the caller's own among them) that run on the CPUs of the NUMA node the device hangs off, and go to HBM in one
 * transfer per contiguous run while the previous slice computes; large results come back the same way.  0 (the default): the
 * runtime's own copies out of pageable memory, one per row, on the calling thread wherever it runs.  Measured on a two-socket
 * EPYC 9575F host (profiles/r06_host_pointer_numa.txt): with 0 the call keeps 95.6-96.6 % of the device-resident rate whether the
 * caller's thread and arrays sit on the device's node or the other one; the pool keeps 93-95.4 % (its gather is a second pass over
 * the rows that the runtime's pin-in-place transfer does not make).  It is for hosts whose runtime stages pageable copies through a
 * single thread far from the device; this pool's boxes do not.  At most 64; no more threads are started than the node has CPUs the
 * process may use.  Small calls are not affected (their rows already travel as one pinned image). */
int afx_ctx_set_host_copy_threads(afx_ctx* ctx, uint32_t threads);

/* Challenge trace (parity aid; off by default).  set(rows, count) allocates a device array [rows][count][32]; while it
 * exists, every verification call (presentations, proofs of encryption, issuances) of at most `count` items also
 * stores the challenge it RECOMPUTES for item i of proof r in cell (r, i): r = 0 for the main proof (or the only
 * one), r = 1 + e for the e-th attached proof of encryption; a call that needs more rows or items fails with
 * AFX_E_BAD_ARGS.  get() waits for the context's work and copies the array to host_out (rows * count * 32 bytes).
 * The value is what zkp's verify_compact compares with the proof's challenge (presentation.rs:435,
 * encryption.rs:209, issuance.rs:217), so equal traces mean every recomputed commitment of the item was equal too.
 * Items rejected before the transcript stage (non-canonical scalar, undecodable point) still get a value; it is
 * computed with the identity in place of the undecodable point and means nothing.  set(0, 0) frees the array. */
int afx_ctx_set_challenge_trace(afx_ctx* ctx, size_t rows, size_t count);
int afx_ctx_get_challenge_trace(afx_ctx* ctx, uint8_t* host_out);

/* Per-item operation counts of the context's most recent call (measurement aid: the compute-side figure beside the
 * HBM roofline is derived from these).  Counts are what ONE item executes, summed over all
 * the jobs of its statement; every lane executes the same schedule, so they do not depend on the data. */
typedef struct afx_plan_stats {
  uint64_t msm_jobs;           /* multiscalar multiplications (grid rows of k_msm plus chained jobs)            */
  uint64_t doublings;          /* point doublings in their shared doubling chains (4S + 3M, every fourth 4S + 4M) */
  uint64_t var_additions;      /* additions of a per-item window-table entry (8M, last of a window 7M)          */
  uint64_t fixed_additions;    /* additions of a generator-table entry (7M, last of a window 6M)                */
  uint64_t table_additions;    /* additions spent building the per-item window tables (8M)                      */
  uint64_t encodings;          /* ristretto255 encodings (1 inverse square root = 254S + 11M, plus ~14M)        */
  uint64_t decodings;          /* ristretto255 decodings (same size)                                            */
  uint64_t keccak_permutations;
  uint64_t field_mul, field_sq; /* GF(2^255-19) multiplications / squarings of all of the above, from the kernels' own
                                  schedule (tests/test_device_arith_on_host.py pins the per-block counts)        */
  uint64_t secret_terms;       /* terms of the multiscalar jobs that run with secret-independent addressing (0 unless
                                  afx_ctx_set_secret_independent_addressing is on)                               */
  uint64_t chain_mul, chain_sq; /* the share of field_mul / field_sq inside the inversion and square-root chains, which run in the
                                  10 x 25.5-bit form of the field (100 / 55 multiply-adds each instead of 98 / 62)              */
} afx_plan_stats;
int afx_ctx_get_plan_stats(afx_ctx* ctx, afx_plan_stats* out);

/* The cache of assembled plans (small host-pointer calls reuse the plan of their statement, shape, mode and padded size: about 0.3 ms
 * of host work per call).  At most 512 entries / 64 MB; the least recently used entry makes room for a new one, so a stream of
 * unusual shapes - a serialized batch names its own shape - cannot pin the cache against the shapes a server really sees. */
typedef struct afx_plan_cache_stats {
  uint64_t hits, misses;   /* small host-pointer calls that reused a plan / assembled (and cached) one                     */
  uint64_t evictions;      /* entries dropped to make room                                                                 */
  uint64_t entries, bytes; /* what the cache holds now                                                                     */
} afx_plan_cache_stats;
int afx_ctx_get_plan_cache_stats(afx_ctx* ctx, afx_plan_cache_stats* out);

/* Per-kernel device timing with HIP events on afx_ctx_stream(ctx) (measurement aid; off by default).
 * set_timing(ctx, 1) resets the counters and starts recording every launch; get_timing synchronises the
 * stream and returns the summed duration and launch count of one kernel: "k_msm_window", "k_msm_naf", "k_msm_fixed"
 * (the three multiscalar kernels: per-item windows, uniform width-5 NAF terms, fixed bases only), "k_msm_tables", or
 * "k_msm" for those four together; "k_hash", "k_decode", "k_pointop", "k_scalarop", "k_sccheck", "k_finish",
 * "k_from_uniform", "k_reduce_wide", "k_fill_u32". */
int afx_ctx_set_timing(afx_ctx* ctx, int enable);
int afx_ctx_get_timing(afx_ctx* ctx, const char* kernel, double* total_ms, uint64_t* launches);
/* The core clock the k_msm_window launches recorded since set_timing(ctx, 1) actually ran at, in MHz (0 if none ran): one lane of
 * each of 64 blocks per launch - spread evenly over the launch's block order, that is over its duration and over the eight XCDs -
 * reads the shader-clock counter and the constant 100 MHz counter around its chain; get_core_clock_mhz is the MEDIAN of the 64
 * ratios, get_core_clock_samples all of them in increasing order (at most `cap` written, *n_out = how many there are).  The path
 * runs at the socket power cap, so this is below the nominal clock the multiply-add peak is usually quoted at, and a launch's
 * first blocks run faster than its last. */
int afx_ctx_get_core_clock_mhz(afx_ctx* ctx, double* mhz);
int afx_ctx_get_core_clock_samples(afx_ctx* ctx, double* mhz_out, uint32_t cap, uint32_t* n_out);

/* ---- Issuer::verify (src/issuer.rs:141-147 -> src/nizk/presentation.rs:324-443) ------------- */

/* status[i] = AFX_ST_OK iff Issuer::verify(presentation_i).is_ok().  Host pointers. */
int afx_verify_presentations(afx_ctx* ctx, const afx_shape* shape, const afx_presentation_soa* batch,
                             size_t count, uint8_t* status);
/* Same, all SoA arrays and `status` in device memory; asynchronous on afx_ctx_stream(ctx). */
int afx_verify_presentations_dev(afx_ctx* ctx, const afx_shape* shape, const afx_presentation_soa* batch,
                                 size_t count, uint8_t* status_dev);

/* Items [first, first + n) of a batch of `total` presentations.  Every array of `batch` and `status` is the whole batch's
 * ([k][total][32], status[total]); the call reads and writes only the elements of its range, so several contexts (one per
 * GPU) can work on one batch from several host threads.  The range is staged to HBM in slices that alternate between
 * two streams: the copy of one slice overlaps the kernels of the previous one. */
int afx_verify_presentations_range(afx_ctx* ctx, const afx_shape* shape, const afx_presentation_soa* batch, size_t total,
                                   size_t first, size_t n, uint8_t* status);

/* ---- several GPUs behind one call -------------------------------------------------------------------
 * Issuer::verify and Issuer::issue are single `&self` methods in one process (src/issuer.rs:141-147, :111-124).  A group is
 * that issuer on a node of GPUs: one context per listed device (the same device may be listed twice), each with its own
 * copy of parameters, key and tables.  A group call splits [0, count) into contiguous ranges (afx_shard_bounds), one host
 * thread per member runs afx_*_range on its range, and every member writes its part of the caller's arrays.  No data moves
 * between devices; there is no collective.  Host pointers only.  Settings (strict mode, pass size ...) are per member:
 * afx_group_member().  A call of at most afx_ctx_set_small_batch_items items (member 0's setting) is not cut up - its duration
 * is one chain's either way - but handed whole to ONE member, the next in turn: small calls from several host threads then
 * spread over the devices. */
typedef struct afx_group afx_group;
int afx_group_create(afx_group** out, const int* devices, uint32_t n_devices, const uint8_t* sysparams, size_t sysparams_len,
                     const uint8_t* amacs_key, size_t amacs_key_len, const uint8_t issuer_params[64]);
void afx_group_destroy(afx_group* group);
uint32_t afx_group_size(const afx_group* group);
afx_ctx* afx_group_member(afx_group* group, uint32_t index);
/* member `index` of `members` takes items [*first, *first + *n): contiguous, the first count % members ranges one item longer */
void afx_shard_bounds(size_t count, uint32_t members, uint32_t index, size_t* first, size_t* n);
int afx_group_verify_presentations(afx_group* group, const afx_shape* shape, const afx_presentation_soa* batch, size_t count,
                                   uint8_t* status);

/* ---- wire format (the reference defines none: presentation.rs:117-127 holds decoded
 *      points and has no to_bytes) ------------------------------------------------------------------
 * A batch of same-shape presentations, in the crate's "u32le n || 32-byte items" style (parameters.rs:155-184):
 *   header : "AFXP" | u32le version (1) | u32le count | u32le cells_per_record
 *            | u32le n_attributes | u32le n_responses | u32le n_hidden_scalars | u32le n_enc_proofs
 *            | kinds[n_attributes] (u8) | hidden_scalar_indices[..] (u16le) | enc_indices[..] (u16le) | zero pad to 32 B
 *   records: count x cells_per_record x 32 bytes, array of structs, each record =
 *            challenge | responses[n_responses] | C_x_0 | C_x_1 | C_V | C_y[n_attributes]
 *            | value of every PUBLIC_SCALAR / PUBLIC_POINT attribute, in position order
 *            | per proof of encryption: challenge | responses[6] | pk | E1 | E2 | C_y_1 | C_y_2 | C_y_3 | C_y_2'
 * A single ProofOfValidCredential::to_bytes is the same with count = 1. */
size_t afx_wire_header_bytes(const afx_shape* shape);
uint32_t afx_wire_cells_per_record(const afx_shape* shape);
/* Parse the header (host).  Returns AFX_OK and fills shape/count/records offset, or AFX_E_BAD_ARGS. */
int afx_wire_parse(const uint8_t* blob, size_t len, afx_shape* shape_out, size_t* count_out, size_t* records_offset_out);
/* Write such a batch from the struct-of-arrays (HOST pointers) that afx_show returned: what a user sends to the issuer.  blob == NULL
 * only reports the length needed in *len_out.  Bytes only. */
int afx_wire_pack_presentations(const afx_shape* shape, const afx_presentation_soa* batch, size_t count, uint8_t* blob, size_t blob_cap,
                                size_t* len_out);
/* Issuer::verify over a serialized batch: the records are copied to HBM, transposed to struct-of-arrays by a
 * kernel, and verified.  status must hold `count` bytes (status_cap >= count). */
int afx_verify_presentations_wire(afx_ctx* ctx, const uint8_t* blob, size_t len, uint8_t* status, size_t status_cap, size_t* count_out);
/* Records [first, first + n) of the blob's batch (status is the whole batch's array: element first + i answers record first + i), and the
 * whole blob over a group's devices: the records are contiguous, so every member takes a byte range of the caller's blob. */
int afx_verify_presentations_wire_range(afx_ctx* ctx, const uint8_t* blob, size_t len, size_t first, size_t n, uint8_t* status, size_t status_cap,
                                        size_t* count_out);
int afx_group_verify_presentations_wire(afx_group* group, const uint8_t* blob, size_t len, uint8_t* status, size_t status_cap, size_t* count_out);

/* ---- presentations of DIFFERENT shapes in one call ------------------------------------------------------------------
 * Issuer::verify takes any presentation (src/issuer.rs:141-147): the shape - which attributes are hidden, how many proofs of
 * encryption ride along - is per presentation (src/nizk/presentation.rs:118-127, :293-309), and a server's request stream
 * mixes them.  The two entry points below group such a stream by shape inside the library, run one GPU batch per distinct
 * shape and put every status byte where its presentation stood in the caller's order.  Two shapes are the same group when
 * their used fields are equal (counts, kinds[0..n), hidden_scalar_indices[0..hs), enc_indices[0..ne)); array tails are ignored. */

/* Length in bytes (header + records) of the AFXP v1 section that starts at `blob`, from its header alone; AFX_E_BAD_ARGS if the
 * header is malformed or the section would run past `len`. */
int afx_wire_section_bytes(const uint8_t* blob, size_t len, size_t* section_len_out);
/* `blob` = AFXP v1 sections back to back, in arrival order; each section is a complete same-shape batch (count >= 1; one
 * serialized presentation is a section with count = 1).  Sections of equal shape are merged into one batch whatever their
 * position in the stream.  status[i] answers the i-th presentation of the stream; *count_out = their number.
 * A section whose shape the reference would panic on fails its own items only.  Records are copied once on the host when
 * a group spans several sections (bytes only; no arithmetic leaves the GPU). */
int afx_verify_presentations_mixed_wire(afx_ctx* ctx, const uint8_t* blob, size_t len, uint8_t* status, size_t status_cap, size_t* count_out);
/* ... every shape group split over a group's devices */
int afx_group_verify_presentations_mixed_wire(afx_group* group, const uint8_t* blob, size_t len, uint8_t* status, size_t status_cap, size_t* count_out);

/* The same over struct-of-arrays groups (host pointers).  `positions`, when not NULL, holds for each of the group's items its
 * index in the caller's order: status[positions[i]] answers item i of the group (every index < status_len, each used once
 * over all groups - checked before anything runs).  With positions == NULL the group's statuses are written contiguously,
 * after those of the groups before it.  Groups may repeat a shape. */
typedef struct {
  afx_shape shape;
  afx_presentation_soa batch;  /* the group's own arrays: [k][count][32]                        */
  size_t count;
  const uint64_t* positions;   /* [count] or NULL                                               */
} afx_presentation_group;
int afx_verify_presentations_mixed(afx_ctx* ctx, const afx_presentation_group* groups, size_t n_groups, uint8_t* status, size_t status_len);
/* ... and over a group of GPUs: every shape group is split over the members like afx_group_verify_presentations. */
int afx_group_verify_presentations_mixed(afx_group* group, const afx_presentation_group* groups, size_t n_groups, uint8_t* status,
                                         size_t status_len);

/* A batch of CredentialIssuance messages of one attribute layout (issuer.rs:42-45: proof + credential{amac, attributes};
 * the reference has no to_bytes for it either), same style:
 *   header : "AFXI" | u32le version (1) | u32le count | u32le cells_per_record | u32le n_attributes | u32le n_responses
 *            | kinds[n_attributes] (u8, AFX_ATTR_*) | zero pad to 32 B
 *   records: count x cells_per_record x 32 bytes, each record =
 *            t | U | V | challenge | responses[n_responses] | value of every attribute (Sc, or Pt = M1), position order
 * cells_per_record = 4 + n_responses + n_attributes. */
size_t afx_issuance_wire_header_bytes(uint32_t n_attributes);
int afx_issuance_wire_parse(const uint8_t* blob, size_t len, uint32_t* n_attributes_out, uint8_t kinds_out[AFX_MAX_ATTRIBUTES],
                            uint32_t* n_responses_out, size_t* count_out, size_t* records_offset_out);
/* CredentialIssuance::verify (issuer.rs:48-57) over a serialized batch; the context needs no issuer key. */
int afx_verify_issuances_wire(afx_ctx* ctx, const uint8_t* blob, size_t len, uint8_t* status, size_t status_cap, size_t* count_out);

/* IssuerParameters in the byte form the crate intends (C_W || I, 64 bytes; issuer.rs:155,163 - its own
 * to_bytes/from_bytes are unimplemented!(), parameters.rs:365-372): what afx_ctx_create was given, or what an
 * issuer context derived from its key. */
int afx_ctx_issuer_parameters(afx_ctx* ctx, uint8_t out[64]);

/* ProofOfEncryption::verify alone (src/nizk/encryption.rs:154-210); `index` = ProofOfEncryption.index */
int afx_verify_encryption_proofs(afx_ctx* ctx, uint16_t index, const afx_encproof_soa* batch, size_t count,
                                 uint8_t* status);
int afx_verify_encryption_proofs_dev(afx_ctx* ctx, uint16_t index, const afx_encproof_soa* batch, size_t count,
                                     uint8_t* status_dev);

/* ---- Issuer::issue (src/issuer.rs:111-124 = Amac::tag src/amacs.rs:276-294 +
 *      ProofOfIssuance::prove src/nizk/issuance.rs:40-129) ----------------------------------- */

/* Attributes of a batch of credential requests of one layout (user.rs:137-139). */
typedef struct {
  uint32_t n_attributes;              /* request.attributes.len(); != ctx n => every status MAC_CREATION */
  uint8_t kinds[AFX_MAX_ATTRIBUTES];  /* AFX_ATTR_* per position                                         */
  const uint8_t* values;              /* [n_attributes][count] 32 B: Sc for scalar kinds; Pt (M1) for point kinds */
} afx_attributes_soa;

typedef struct {
  const uint8_t* t_wide;    /* [count][64]  the 64 bytes Scalar::random draws        (amacs.rs:289)  */
  const uint8_t* U_wide;    /* [count][64]  the 64 bytes RistrettoPoint::random draws (amacs.rs:290) */
  const uint8_t* rng_seed;  /* [count][32]  the 32 bytes zkp's prove_compact draws from thread_rng() */
} afx_issue_randomness;

typedef struct {
  uint8_t* t;          /* [count] Sc    amac.t */
  uint8_t* U;          /* [count] Pt    amac.U */
  uint8_t* V;          /* [count] Pt    amac.V */
  uint8_t* challenge;  /* [count] Sc    ProofOfIssuance.0.challenge */
  uint8_t* responses;  /* [n+5][count] Sc  ProofOfIssuance.0.responses (w,w',x_0,x_1,y_0..y_{n-1},1) */
} afx_issuance_soa;

int afx_issue(afx_ctx* ctx, const afx_attributes_soa* requests, const afx_issue_randomness* rnd, size_t count,
              const afx_issuance_soa* out, uint8_t* status);
int afx_issue_dev(afx_ctx* ctx, const afx_attributes_soa* requests, const afx_issue_randomness* rnd, size_t count,
                  const afx_issuance_soa* out, uint8_t* status_dev);

/* Requests [first, first + n) of a batch of `total`; arrays are the whole batch's, as for afx_verify_presentations_range. */
int afx_issue_range(afx_ctx* ctx, const afx_attributes_soa* requests, const afx_issue_randomness* rnd, size_t total, size_t first,
                    size_t n, const afx_issuance_soa* out, uint8_t* status);
int afx_group_issue(afx_group* group, const afx_attributes_soa* requests, const afx_issue_randomness* rnd, size_t count,
                    const afx_issuance_soa* out, uint8_t* status);

/* CredentialIssuance::verify (src/issuer.rs:48-57 -> src/nizk/issuance.rs:132-218), user side. */
int afx_verify_issuances(afx_ctx* ctx, const afx_attributes_soa* attrs, const afx_issuance_soa* issuances,
                         uint32_t n_responses, size_t count, uint8_t* status);
int afx_verify_issuances_dev(afx_ctx* ctx, const afx_attributes_soa* attrs, const afx_issuance_soa* issuances,
                             uint32_t n_responses, size_t count, uint8_t* status_dev);
/* Issuances [first, first + n) of a batch of `total`, and the whole batch over a group's devices. */
int afx_verify_issuances_range(afx_ctx* ctx, const afx_attributes_soa* attrs, const afx_issuance_soa* issuances,
                               uint32_t n_responses, size_t total, size_t first, size_t n, uint8_t* status);
int afx_group_verify_issuances(afx_group* group, const afx_attributes_soa* attrs, const afx_issuance_soa* issuances,
                               uint32_t n_responses, size_t count, uint8_t* status);
/* Write an AFXI batch (the wire format above) from what afx_issue returned (HOST pointers): what the issuer sends back to its users.  blob == NULL only
 * reports the length needed. */
int afx_issuance_wire_pack(const afx_attributes_soa* attrs, const afx_issuance_soa* issuances, uint32_t n_responses, size_t count, uint8_t* blob,
                           size_t blob_cap, size_t* len_out);

/* ---- AnonymousCredential::show (src/credential.rs:37-46 -> src/nizk/presentation.rs:139-321) - */

/* A batch of credentials of one layout, as the user holds them. */
typedef struct {
  uint32_t n_attributes;
  uint8_t kinds[AFX_MAX_ATTRIBUTES];  /* AFX_ATTR_* per position (after hide_attribute/reveal_attribute) */
  const uint8_t* values;              /* [n][count] 32 B: Sc (scalar kinds) or Pt M1 (point kinds)        */
  const uint8_t* M2;                  /* [n][count] Pt: Plaintext.M2 rows for SECRET_POINT positions     */
  const uint8_t* m3;                  /* [n][count] Sc: Plaintext.m3 rows for SECRET_POINT positions     */
  const uint8_t* t;                   /* [count] Sc */
  const uint8_t* U;                   /* [count] Pt */
  const uint8_t* V;                   /* [count] Pt */
} afx_credentials_soa;

/* symmetric::Keypair (src/symmetric.rs:52-56,68-81); one keypair per item */
typedef struct {
  const uint8_t* a;   /* [count] Sc */
  const uint8_t* a0;  /* [count] Sc */
  const uint8_t* a1;  /* [count] Sc */
  const uint8_t* pk;  /* [count] Pt */
} afx_keypairs_soa;

typedef struct {
  const uint8_t* z_wide;     /* [count][64]   Scalar::random for the nonce z (presentation.rs:162)       */
  const uint8_t* rng_seed;   /* [count][32]   thread_rng() draw of the presentation's prove_compact      */
  const uint8_t* enc_seeds;  /* [n_secret_points][count][32]  same for each ProofOfEncryption, in order   */
} afx_show_randomness;

/* Output arrays mirror afx_presentation_soa / afx_encproof_soa but writable. */
typedef struct {
  uint8_t* challenge; uint8_t* responses;
  uint8_t* pk; uint8_t* E1; uint8_t* E2; uint8_t* C_y_1; uint8_t* C_y_2; uint8_t* C_y_3; uint8_t* C_y_2p;
} afx_encproof_out;
typedef struct {
  uint8_t* challenge; uint8_t* responses;      /* [3+hs][count] */
  uint8_t* C_x_0; uint8_t* C_x_1; uint8_t* C_V;
  uint8_t* C_y;                                /* [n][count] */
  uint8_t* attr_values;                        /* [n][count]: revealed values copied through */
  const afx_encproof_out* enc;                 /* [n_secret_points] */
} afx_presentation_out;

/* keypairs == NULL with a SECRET_POINT attribute => every status AFX_ST_NO_SYMMETRIC_KEY.
 * `shape_out` receives the presentation shape (kinds, hidden indices, enc indices). */
int afx_show(afx_ctx* ctx, const afx_credentials_soa* creds, const afx_keypairs_soa* keypairs,
             const afx_show_randomness* rnd, size_t count, const afx_presentation_out* out, afx_shape* shape_out,
             uint8_t* status);
int afx_show_dev(afx_ctx* ctx, const afx_credentials_soa* creds, const afx_keypairs_soa* keypairs,
                 const afx_show_randomness* rnd, size_t count, const afx_presentation_out* out, afx_shape* shape_out,
                 uint8_t* status_dev);
/* Credentials [first, first + n) of a batch of `total` (every output array indexed like the inputs; shape_out is the same
 * for every range of one batch), and the whole batch over a group's devices (a user-side group is created with
 * amacs_key == NULL). */
int afx_show_range(afx_ctx* ctx, const afx_credentials_soa* creds, const afx_keypairs_soa* keypairs,
                   const afx_show_randomness* rnd, size_t total, size_t first, size_t n, const afx_presentation_out* out,
                   afx_shape* shape_out, uint8_t* status);
int afx_group_show(afx_group* group, const afx_credentials_soa* creds, const afx_keypairs_soa* keypairs,
                   const afx_show_randomness* rnd, size_t count, const afx_presentation_out* out, afx_shape* shape_out,
                   uint8_t* status);

/* ---- requests, credentials and issuances of DIFFERENT layouts in one call -------------------------------------------
 * Issuer::issue takes any request (src/issuer.rs:111-124: the attribute kinds are per attribute, src/amacs.rs:168-179),
 * AnonymousCredential::show any credential in whatever state its hide_attribute / reveal_attribute calls left it
 * (src/credential.rs:37-46, :53-97) and CredentialIssuance::verify any issuance (src/issuer.rs:48-57).  As for
 * afx_verify_presentations_mixed above, the caller hands over one struct-of-arrays group per distinct layout; each group has its
 * own input AND output arrays ([k][count][32], item i of the group in element i), and only the status bytes go back to the
 * caller's order: status[positions[i]] answers item i of the group (positions == NULL: contiguous, after the groups before it;
 * every index < status_len and used once over all groups - checked before anything runs).  Groups may repeat a layout.  Small
 * groups are assembled into ONE set of kernel launches (mixed.cpp), so a stream of many layouts costs about one small call. */
typedef struct {
  afx_attributes_soa requests;   /* n_attributes != ctx n => the group's statuses are all MAC_CREATION (amacs.rs:285-287)      */
  afx_issue_randomness rnd;
  afx_issuance_soa out;          /* the group's own output arrays; responses [ctx n + 5][count][32]                           */
  size_t count;
  const uint64_t* positions;     /* [count] or NULL                                                                           */
} afx_issue_group;
int afx_issue_mixed(afx_ctx* ctx, const afx_issue_group* groups, size_t n_groups, uint8_t* status, size_t status_len);
int afx_group_issue_mixed(afx_group* group, const afx_issue_group* groups, size_t n_groups, uint8_t* status, size_t status_len);

typedef struct {
  afx_attributes_soa attrs;
  afx_issuance_soa issuances;    /* read only here                                                                            */
  uint32_t n_responses;          /* proof.responses.len() of the group's issuances                                            */
  size_t count;
  const uint64_t* positions;
} afx_issuance_group;
int afx_verify_issuances_mixed(afx_ctx* ctx, const afx_issuance_group* groups, size_t n_groups, uint8_t* status, size_t status_len);
int afx_group_verify_issuances_mixed(afx_group* group, const afx_issuance_group* groups, size_t n_groups, uint8_t* status, size_t status_len);

typedef struct {
  afx_credentials_soa creds;
  const afx_keypairs_soa* keypairs;  /* or NULL (see afx_show)                                                                */
  afx_show_randomness rnd;
  afx_presentation_out out;          /* the group's own output arrays                                                         */
  afx_shape shape_out;               /* written: the presentation shape of this group                                         */
  size_t count;
  const uint64_t* positions;
} afx_show_group;
int afx_show_mixed(afx_ctx* ctx, afx_show_group* groups, size_t n_groups, uint8_t* status, size_t status_len);
int afx_group_show_mixed(afx_group* group, afx_show_group* groups, size_t n_groups, uint8_t* status, size_t status_len);

/* ---- setup helpers (cold path; still GPU arithmetic) ---------------------------------------- */

/* IssuerParameters::generate (src/parameters.rs:349-362) and W = w*G_w (src/amacs.rs:104): given
 * sysparams and the key scalars laid out as amacs::SecretKey::to_bytes minus the trailing W
 * (u32 n || w || w' || x0 || x1 || y[n]), writes W (32 B) and C_W || I (64 B). */
int afx_issuer_keygen(int device, const uint8_t* sysparams, size_t sysparams_len, const uint8_t* key_scalars,
                      size_t key_scalars_len, uint8_t W_out[32], uint8_t issuer_params_out[64]);

/* SystemParameters::hash_and_pray (src/parameters.rs:196-326).  `rng_stream` stands in for csprng.fill_bytes: it is
 * consumed 32 bytes per attempt, in the reference's order (G_w, G_w', G_x0, G_x1, G_y.., G_m.., G_V, G_a, G_a0, G_a1);
 * the decompression tests run on the GPU.  Writes SystemParameters::to_bytes; *consumed_out = bytes of the stream used.
 * AFX_E_BAD_ARGS if the stream runs out, AFX_E_BAD_PARAMS for CredentialError::NoSystemParameters (duplicates). */
int afx_system_parameters_generate(int device, uint32_t n_attributes, const uint8_t* rng_stream, size_t stream_len,
                                   uint8_t* params_out, size_t params_cap, size_t* consumed_out);
/* impl From<&[u8; 30]> for Plaintext (src/symmetric.rs:135-143): M1 = encode_to_group (src/encoding.rs:56-70; counter out),
 * M2 = hash-to-group, m3 = hash-to-scalar.  msgs [count][30]; SHA-512 runs on the host, everything else on the GPU. */
int afx_plaintexts_from_bytes(afx_ctx* ctx, const uint8_t* msgs, size_t count, uint8_t* M1, uint8_t* M2, uint8_t* m3, uint32_t* counters);
/* Keypair::derive (src/symmetric.rs:197-215): master_secrets [count][64] -> a, a0, a1, pk ([count][32] each). */
int afx_keypairs_derive(afx_ctx* ctx, const uint8_t* master_secrets, size_t count, uint8_t* a, uint8_t* a0, uint8_t* a1, uint8_t* pk);
/* Keypair::encrypt (src/symmetric.rs:252-261) and Keypair::decrypt (:273-289; status AFX_ST_UNDECRYPTABLE on mismatch;
 * `messages` [count][30] = decode_from_group of the recovered M1, may be NULL). */
int afx_encrypt(afx_ctx* ctx, const afx_keypairs_soa* keypairs, const uint8_t* M1, const uint8_t* M2, const uint8_t* m3, size_t count,
                uint8_t* E1, uint8_t* E2, uint8_t* status);
int afx_decrypt(afx_ctx* ctx, const afx_keypairs_soa* keypairs, const uint8_t* E1, const uint8_t* E2, size_t count, uint8_t* M1,
                uint8_t* M2, uint8_t* m3, uint8_t* messages, uint8_t* status);

/* Batch ristretto255 primitives (dalek CompressedRistretto::decompress -> compress round trip,
 * RistrettoPoint::from_uniform_bytes, Scalar::from_bytes_mod_order_wide); used to build synthetic
 * attribute values and by the parity tests of the K* rows. */
int afx_points_from_uniform_bytes(afx_ctx* ctx, const uint8_t* wide /*[count][64]*/, size_t count, uint8_t* out /*[count] Pt*/);
int afx_scalars_from_wide_bytes(afx_ctx* ctx, const uint8_t* wide /*[count][64]*/, size_t count, uint8_t* out /*[count] Sc*/);
int afx_points_validate(afx_ctx* ctx, const uint8_t* pts /*[count] Pt*/, size_t count, uint8_t* ok /*[count]*/,
                        uint8_t* reencoded /*[count] Pt or NULL*/);
/* out[i] = sum_k scalars[k][i] * points[k][i]  (RistrettoPoint::multiscalar_mul, amacs.rs:270) */
int afx_multiscalar_mul(afx_ctx* ctx, uint32_t n_terms, const uint8_t* scalars /*[n_terms][count] Sc*/,
                        const uint8_t* points /*[n_terms][count] Pt*/, size_t count, uint8_t* out /*[count] Pt*/,
                        uint8_t* ok /*[count]*/);

/* A merlin transcript over a batch (the K* row's STROBE-128 / Keccak-f[1600] layer by itself: every proof of the crate is bound to a
 * merlin transcript through zkp's TranscriptProtocol [3P], e.g. src/nizk/presentation.rs:355-356 and :435).  The
 * transcript is given as a script - the operations of merlin::Transcript, byte strings length-prefixed with u32 LE:
 *     AFX_MERLIN_NEW          label                      Transcript::new(label); first, once
 *     AFX_MERLIN_APPEND       label, message             append_message(label, message): the same bytes for every item
 *     AFX_MERLIN_APPEND_FIELD label, u32 field index     append_message(label, fields[index][item]): a 32-byte per-item message
 *     AFX_MERLIN_CHALLENGE    label, u32 n (1 .. 64)     challenge_bytes(label, n bytes); the script ends with one
 * and out64[item] receives 64 bytes of which the first n are the LAST challenge (the rest is what the sponge's state held behind
 * them).  A challenge earlier in the script acts on the transcript as it must - label and length absorbed, the sponge permuted, its
 * bytes zeroed - but is not returned: a caller that chains challenges (each absorbed again) runs one call per challenge and feeds
 * the earlier ones back as fields.
 * Compiled and run exactly like the statements' own transcripts (strobe_sim.hpp, k_hash): all-constant leading blocks are absorbed
 * once on the host, everything from the first per-item field on by the kernel.  tests/test_gpu_primitives.py runs merlin's published
 * conformance vectors through it. */
#define AFX_MERLIN_NEW 1
#define AFX_MERLIN_APPEND 2
#define AFX_MERLIN_APPEND_FIELD 3
#define AFX_MERLIN_CHALLENGE 4
int afx_merlin_challenges(afx_ctx* ctx, const uint8_t* script, size_t script_len, const uint8_t* const* fields /* n_fields x [count][32] */,
                          uint32_t n_fields, size_t count, uint8_t* out64 /* [count][64] */);

#ifdef __cplusplus
}
#endif
#endif /* AEONFLUX_GPU_H */
