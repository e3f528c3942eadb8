// Plain-old-data "plan" structures shared by the host engine (engine.cpp) and the HIP kernels
// (kernels.hip).  A plan is the batch-uniform description of one proof statement: which points to
// decompress, which small multiscalar multiplications to run, and the STROBE/merlin byte schedule
// with 32-byte holes for per-item values.  Per-item data never appears here — only device pointers
// to struct-of-arrays batches.
#pragma once
#include <stdint.h>

// afx_ctx_set_plan_variants (include/aeonflux_gpu.h AFX_VARIANT_*): the bits the launch wrappers look at
#define AFX_KV_ONE_WAVE_CHAINS 0x08u     /* no four-wave chains (k_msm_quad, k_pointsum_quad, k_pointsum_tree) */
#define AFX_KV_HASH_HALF_WAVE 0x10u      /* cooperative transcripts on 32 lanes per item, never a wave each */
#define AFX_KV_NO_POINTSUM_TREE 0x20u    /* sums of many parts without the lane-per-part kernel */

#define AFX_CLOCK_SLOTS 64     /* blocks of a k_msm launch that read the clock counters (kernels.hip msm_body): (cycles, 100 MHz ticks) each */
#define AFX_MSM_MAX_TERMS 72   /* Z job: 2 + n + #public scalars <= 2 + 2n; issuance: n + 4 */
#define AFX_TABLE_ENTRIES 9            /* variable bases: 0*P (identity) .. 8*P, signed 4-bit windows */
#define AFX_TABLE_STORED 8             /* entries kept per table: 1*P .. 8*P (digit 0 reads one shared identity entry), or the 8 odd
                                          multiples of a width-5 NAF term */
/* positional tables: for every window position j, d * 2^(AFX_POS_BITS * j) * G for d = 0 .. 2^(AFX_POS_BITS-1), affine
 * niels.  A fixed-base term costs AFX_POS_WINDOWS additions and no doubling, wherever in the job it is added. */
/* Width chosen by measurement (same box, -DAFX_POS_BITS builds): 10 -> 13 bits: C3 +1.1 %, C5 +4.6 %, show +2.5 %; 14..16 bits no
 * further gain (the gathers leave the caches); 9.2 MB of tables per generator, 13 ms per context to build them. */
#ifndef AFX_POS_BITS
#define AFX_POS_BITS 13
#endif
#define AFX_POS_WINDOWS ((253 + AFX_POS_BITS - 1) / AFX_POS_BITS)
#define AFX_POS_ENTRIES ((1 << (AFX_POS_BITS - 1)) + 1)
#define AFX_POS_WINDOW_DWORDS ((AFX_POS_ENTRIES * AFX_NIELS_DWORDS + 3) & ~3)   /* 16-byte multiple */
#define AFX_POS_TABLE_DWORDS (AFX_POS_WINDOWS * AFX_POS_WINDOW_DWORDS)
/* Positional tables for SECRET scalars on fixed bases (afx_ctx_set_secret_independent_addressing): signed 6-bit windows.  The 32
 * stored multiples d = 1..32 of a window are read one per lane (an address that depends on the lane's id only) and every lane takes
 * the multiple its digit names from the lane that holds it (ds_bpermute_b32: no memory access; kernels.hip
 * msm_add_positional_secret) - no address depends on a digit.  43 additions per term instead of AFX_POS_WINDOWS; 155 KB per
 * generator, built at context creation. */
#define AFX_SEC_BITS 6
#define AFX_SEC_WINDOWS ((253 + AFX_SEC_BITS - 1) / AFX_SEC_BITS)
#define AFX_SEC_ENTRIES ((1 << (AFX_SEC_BITS - 1)) + 1)
#define AFX_SEC_WINDOW_DWORDS (AFX_SEC_ENTRIES * AFX_NIELS_DWORDS)
#define AFX_SEC_TABLE_DWORDS (AFX_SEC_WINDOWS * AFX_SEC_WINDOW_DWORDS)
/* Variable bases of a job that has a secret scalar on one of them (afx_msm_job.narrow): signed windows of AFX_SECVAR_BITS bits
 * instead of 4.  Every stored entry of the lane's table is read for every addition of such a term, which makes the chain HBM-bound
 * (profiles/r03_secret_mode_traffic.txt: 4.5 TB/s of table scans at 4 bits); a 2-bit window reads 2 entries for each of 128
 * additions instead of 8 for each of 64 - half the bytes for twice the additions, and a table of 2 entries instead of 8 to build.
 * Measured on one box (profiles/r03_secret_independent_second_pass.txt; 3 bits lay between 2 and 4).  The chain keeps the NEXT
 * addition's entries in registers while the current one computes (kernels.hip msm_chain_narrow): 32 dwords per stored entry, so
 * wider windows than 2 bits would not fit the register file. */
#ifndef AFX_SECVAR_BITS
#define AFX_SECVAR_BITS 2
#endif
#define AFX_SECVAR_WINDOWS ((256 + AFX_SECVAR_BITS - 1) / AFX_SECVAR_BITS)   /* the biased scalar has up to 256 (+ a window) bits */
#define AFX_SECVAR_STORED (1 << (AFX_SECVAR_BITS - 1))                      /* multiples 1 .. 2^(bits-1) */
#define AFX_DIGIT_WORDS 9              /* recoded scalar: up to 260 bits (253 + one window of bias) */
#define AFX_VAR_DWORDS 36              /* extended point: X,Y,Z,T x 9 limbs */
#define AFX_NIELS_DWORDS 28            /* affine niels: (y+x)/2, (y-x)/2, dxy x 9 limbs + 1 dword of padding = 7 x 16 bytes */
#define AFX_TABLE_ENTRY_DWORDS 32       /* a window-table entry: 4 field elements as canonical 256-bit words = 128 B = 2 HBM sectors */
#define AFX_VAR_TABLE_DWORDS (AFX_TABLE_STORED * AFX_TABLE_ENTRY_DWORDS)
#define AFX_BLOCK 256
#define AFX_HASH_COOP_GROUPS 4096       /* a small pass hashes with 32 lanes per (item, program) - k_hash_coop - while items x programs of
                                          the launch stay within this: beyond it the lane groups queue up and one lane per item is as
                                          fast (measured: issue, 1-2 programs per launch, gains up to 4096 items; show and verify,
                                          5 programs, up to ~800: tools/small_call_latency.py) */
/* field multiplications / squarings of one ristretto255 decoding / encoding as ge.cuh implements them
 * (measured on the host build of that header: tests/test_device_arith_on_host.py) */
#define AFX_DECODE_MUL 27
#define AFX_DECODE_SQ 257
#define AFX_ENCODE_MUL 32
#define AFX_ENCODE_SQ 255
/* of which inside the inversion / square-root chains, which run in the 10 x 25.5-bit form (fe10.cuh): z^((p-5)/8) of a decoding or
 * an encoding, z^(p-2) of a plain inversion (k_compress2x, k_negenc) */
#define AFX_CHAIN_SQRT_MUL 11
#define AFX_CHAIN_SQRT_SQ 251
#define AFX_CHAIN_INVERT_MUL 11
#define AFX_CHAIN_INVERT_SQ 254

/* per-item failure bits OR-ed into the `bad` word of an item */
#define AFX_BAD_DECODE 1u
#define AFX_BAD_IDENTITY 2u
#define AFX_BAD_SCALAR 4u
#define AFX_BAD_CHALLENGE 8u
#define AFX_BAD_SHAPE 16u

/* One PASS = one statement over one batch of `count` items: the item count (which is also the stride of every struct-of-arrays
 * variable of the pass), its per-item failure words and its window-table / recoded-scalar workspace.  A kernel launch carries an
 * array of passes and, per grid row, the index of the row's pass (row_pass; null = every row belongs to pass 0): the launches of
 * SEVERAL small calls - presentations of many shapes behind one Issuer::verify stream - are merged row by row into one launch
 * per kernel (engine.cpp PlanSet), so that a stream of n shapes costs one small call's dozen launches, not n dozens. */
typedef struct {
  uint32_t count;
  uint32_t pad;
  uint32_t* bad;         /* [count] failure bits (AFX_BAD_*)                                              */
  int32_t* table_ws;     /* window tables of the pass's k_msm launches                                    */
  uint32_t* digit_ws;    /* recoded scalars of the pass's k_msm launches                                  */
} afx_pass;

/* A grid row of a MERGED launch: which pass the row belongs to and where its job lies, as a byte offset from the launch's `jobs`
 * pointer (then the base of the blob holding every plan's job arrays where the plans left them: merging copies no job).  A plan's
 * own launch passes rows == null: row r runs jobs[r] of pass 0. */
typedef struct {
  uint32_t pass;
  uint32_t job_off;
} afx_row;

/* a grid row of k_compress2x / k_negenc: the jobs one lane walks for its item with a single field inversion */
typedef struct {
  uint32_t job_off;             /* first job of the walk: byte offset from the launch's `jobs` pointer                 */
  uint32_t n_jobs;
  uint32_t pass, pad;
  int32_t* prefix_ws;           /* scratch for the walk's prefix products: n_jobs * 9 * count dwords                   */
} afx_walk_row;

/* rows of the small utility launches that open and close a plan */
typedef struct { uint32_t* p; uint32_t v, n; } afx_fill_job;                                            /* p[0..n) = v                */
typedef struct { const uint32_t* bad; uint8_t* status; uint32_t count, fail_code; } afx_finish_job;     /* status[i] = bad[i] ? code : 0 */
typedef struct { const uint8_t* wide; uint8_t* out_enc; int32_t* out_var; } afx_uniform_job;            /* RistrettoPoint::from_uniform_bytes */
typedef struct { const uint8_t* wide; uint8_t* out; } afx_reduce_job;                                   /* Scalar::from_bytes_mod_order_wide  */

/* variable point storage: struct-of-arrays, limb (c*9+l) of item i at base[(c*9+l)*count + i] */
typedef int32_t* afx_var_t;

typedef struct {
  const uint8_t* enc;      /* [count][32] compressed input                                            */
  afx_var_t out;           /* extended coordinates out (or null)                                       */
  uint32_t reject_identity; /* the point is allocated into a transcript: identity encoding fails      */
  uint32_t elligator;       /* 1, 2: not a decoding - `enc` is a [count][64] array of uniform bytes and out = the Elligator map of each
                               record's first / second 32 bytes (a small pass runs from_uniform's two maps beside its decodings:
                               Assembler::from_uniform; only launches marked Launch::odd hold such jobs, k_decode_mixed runs them) */
} afx_decode_job;

typedef struct {
  const uint8_t* sc;       /* [count][32] scalar array to test for canonicity */
} afx_sccheck_job;

/* out = sa*A + sb*B, sa,sb in {-1,0,+1}; B may be a batch constant (extended coords, AFX_VAR_DWORDS dwords) */
typedef struct {
  const int32_t* a;        /* var (SoA)                           */
  const int32_t* b;        /* var (SoA), or null                  */
  const int32_t* b_const;  /* AFX_VAR_DWORDS uniform, or null     */
  int32_t sa, sb;
  afx_var_t out;           /* may be null                         */
  uint8_t* out_enc;        /* [count][32] compressed, may be null */
  uint32_t reject_identity; /* 1: the identity fails the item; 2: anything BUT the identity fails it */
} afx_pointop_job;

/* out[i] = a[i or uniform] * b[i] (+ c[i])  mod l ; optionally negated */
typedef struct {
  const uint8_t* a; uint32_t a_stride;   /* 32 per item, 0 = uniform */
  const uint8_t* b; uint32_t b_stride;
  const uint8_t* c; uint32_t c_stride;   /* null = no addend */
  uint32_t negate;
  uint8_t* out;                          /* [count][32] */
} afx_scalarop_job;

typedef struct {
  const uint8_t* scalar;   /* [count][32] per item, or one 32-byte scalar when scalar_stride == 0 */
  uint32_t scalar_stride;  /* 32 or 0 */
  int32_t fixed_idx;       /* >= 0: generator id (positional tables); -1: variable point           */
  const int32_t* var;      /* variable point (SoA) when fixed_idx < 0                              */
  uint32_t negate;         /* subtract the term                                                    */
  uint32_t table_slot;     /* variable terms: slot of this base's window table in table_ws (Assembler::msm; terms of one
                              launch list that share a base and a table kind share the table)              */
  uint32_t secret;         /* the scalar is a secret and the context runs with secret-independent addressing (Assembler::msm
                              sets it): on a per-item base both stored entries of the 2-bit window's affine table are read and
                              the digit's selected; a fixed base uses the 6-bit positional tables (AFX_SEC_*), a window's 32
                              multiples loaded one per lane and the digit's taken through the lane exchange (ds_bpermute_b32) */
  uint32_t dbl;            /* the base holds HALF the point the statement means (its producer left its half for k_compress2x,
                              afx_msm_job.leave_half): the term's scalar counts twice (Assembler::msm sets it)           */
  uint32_t win_off;        /* a SEGMENT of a secret scalar on a per-item base (Assembler::msm_split, small prover passes): the term's chain
                              runs the job's `wins` windows and window w takes digit w + win_off of the recoded scalar; the base is the
                              term's own point times 2^(AFX_SECVAR_BITS * win_off) (afx_powers_job made it).  0 otherwise.        */
  uint32_t pad;
} afx_msm_term;

typedef struct {
  uint32_t n_terms;
  uint32_t n_var;                       /* number of variable terms (they come first in term[])   */
  uint32_t n_uni;                       /* the first n_uni variable terms have a batch-constant scalar known to the host:
                                           width-5 NAF digits, identical schedule for every lane (Assembler::msm sets these four) */
  int32_t top_bit;                      /* n_uni != 0: highest bit position at which anything is added            */
  const uint32_t* naf_sched;            /* the nonzero NAF digits as a list of additions, highest bit position first:
                                           (bit << 16) | (term << 8) | (negative << 7) | table index 0..7 (digit = +-(2 idx + 1),
                                           sign of `negate` folded in); terminated by 0xffffffff.  Uniform => scalar loads */
  afx_msm_term term[AFX_MSM_MAX_TERMS];
  const int32_t* addend;                /* optional variable point added at the end               */
  uint32_t addend_negate;
  uint8_t* out_enc;                     /* [count][32] compressed result, may be null             */
  afx_var_t out_var;                    /* extended result, may be null                           */
  uint32_t reject_identity;
  uint32_t leave_half;                  /* host only, set by the statement code on a job whose result is BOTH encoded and used as a base
                                           by later multiscalar terms (and by nothing else): the job runs on halved scalars, out_var
                                           receives half the result, k_compress2x encodes the double, and every later term on that base
                                           doubles its scalar instead (afx_msm_term.dbl) - one inverse square root less */
  int32_t chain_to;                     /* host only: index (in the vector handed to Assembler::msm) of a job that consumes this
                                           job's out_var and therefore goes into a later launch; -1 none */
  afx_var_t half_var;                   /* non-null: the job runs on HALVED scalars (s/2 mod l for every term), stores the sum here and
                                           does not encode; k_compress2x then encodes TWICE the stored point, which needs one field
                                           inversion per item for all such jobs together instead of a square root each (Assembler::msm) */
  uint32_t digit_slot;                  /* first recoded-scalar slot of this job in digit_ws (one per term)   */
  uint32_t narrow;                      /* one of the variable terms has a secret scalar under secret-independent addressing: the job's
                                           variable terms run AFX_SECVAR_BITS-bit windows over tables of AFX_SECVAR_STORED entries
                                           (Assembler::msm sets it; only launches of the SEC kernel instances hold such jobs).
                                           2: the same over CACHED entries (a segmenting pass skips k_table_affine; four-wave chains only) */
  uint32_t wins;                        /* narrow jobs: how many windows the chain runs - AFX_SECVAR_WINDOWS (0 means that), or a segment's
                                           share when its terms are segments (afx_msm_term.win_off)                            */
} afx_msm_job;

/* The same job as the kernels read it (Assembler::msm_list writes this form into the plan's blob): the terms it HAS, in a side
 * array, instead of room for AFX_MSM_MAX_TERMS of them - 0.1 KB + 40 B a term instead of 2.9 KB a job (a small C3 verification
 * has 86 jobs of 1-3 terms: 14 KB of plan instead of 250 KB to assemble, relocate, copy and send). */
typedef struct {
  uint32_t n_terms, n_var, n_uni;
  int32_t top_bit;
  const uint32_t* naf_sched;
  int32_t term_off;                     /* the job's n_terms afx_msm_term entries lie at (const uint8_t*)job + term_off: an OFFSET, so that the
                                           kernels reach them from their own kernel argument (scalar loads the compiler can prove unclobbered;
                                           a loaded pointer made every term field a vector load and every branch on one an exec-mask branch) */
  uint32_t wins;                        /* narrow jobs: windows of the chain (afx_msm_job.wins, never 0 here) */
  const int32_t* addend;
  uint32_t addend_negate;
  uint32_t reject_identity;
  uint8_t* out_enc;
  afx_var_t out_var;
  afx_var_t half_var;
  uint32_t digit_slot;
  uint32_t narrow;
  uint32_t leave_half, pad;             /* host-side checks only (tests/hostsim)                  */
} afx_msm_djob;

#ifdef __cplusplus
static inline
#ifdef __HIPCC__
__host__ __device__
#endif
const afx_msm_term* afx_job_terms(const afx_msm_djob* j) { return (const afx_msm_term*)((const uint8_t*)j + j->term_off); }
#endif

/* one window table to build (k_msm_tables<kind>): the base and where the table goes.  Kinds (one launch each): 0 the multiples
 * 1..8 (signed 4-bit windows), item-major [item][entry] - a lane's digit picks one entry; 1 the odd multiples 1, 3, .., 15 (terms
 * that run a width-5 NAF); 2 the multiples 1..AFX_SECVAR_STORED of a narrow job.  Kinds 1 and 2 are entry-major,
 * [entry][16-byte piece][item]: every lane of a wave reads the same entry, so a wave's load is 1 KB contiguous. */
typedef struct {
  const int32_t* var;      /* variable point (SoA)                                                  */
  uint32_t table_slot;     /* slot in table_ws                                                      */
  uint32_t pad;
} afx_table_job;

/* k_powers: out[i][item] = 2^(step * (i + 1)) * src[item], i < n_out: the bases of a secret scalar's SEGMENTS on a per-item point
 * (afx_msm_term.win_off) - `step` doublings in a row between one and the next, a chain nobody waits for twice: every later stage of
 * the pass multiplies by short chains on these instead of a full-length one on src. */
#define AFX_POWERS_MAX 7
typedef struct {
  const int32_t* src;            /* variable point (SoA) */
  afx_var_t out[AFX_POWERS_MAX];
  uint32_t step, n_out;
} afx_powers_job;

/* k_compress2x: out_enc[item] = encoding of 2 * var[item] (ristretto255) */
typedef struct {
  const int32_t* var;      /* SoA extended point, the half of what is to be encoded                 */
  uint8_t* out_enc;        /* [count][32]                                                            */
  uint32_t reject_identity;
  uint32_t negate;         /* 1: encode -2 * var instead (the "-E1" beside an E1 that left its half); 2 (AFX_COMPRESS_PLAIN): encode var
                              ITSELF, with an inverse square root of its own - a point that is no half (a sum of decoded points), whose
                              encoding a small pass moves here from k_pointop so that it runs beside the other encodings instead of
                              before the chains (Assembler::pointop)                                                          */
} afx_compress_job;
#define AFX_COMPRESS_PLAIN 2u

/* k_negenc: out_enc[item] = encoding of -P, P = the point that `enc` decodes to (coordinates in `var`, Z = 1, as k_decode left
 * them): no square root, one field inversion per item for all its jobs (ge.cuh negenc_*) */
typedef struct {
  const uint8_t* enc;      /* [count][32] the encoding P was decoded from                            */
  const int32_t* var;      /* SoA extended P                                                          */
  uint8_t* out_enc;        /* [count][32]                                                             */
  uint32_t reject_identity;
  uint32_t pad;
} afx_negenc_job;

/* k_pointsum (small batches: Assembler::msm splits a job into one chain per term and sums the partial results here):
 * out = sum of parts (+- addend); stored to out_var and/or half_var (then encoded by k_compress2x), or encoded here */
typedef struct {
  const int32_t* const* parts;  /* device array of n_parts SoA extended points                     */
  uint32_t n_parts;
  uint32_t addend_negate;
  const int32_t* addend;        /* optional                                                         */
  afx_var_t out_var;            /* may be null                                                      */
  afx_var_t half_var;           /* may be null: the parts were computed on halved scalars           */
  uint8_t* out_enc;             /* [count][32], encoded in this kernel when half_var is null; may be null */
  uint32_t reject_identity;
  uint32_t pad;
} afx_pointsum_job;

/* one 8-byte word of a STROBE rate block: st = (st & keep) ^ c ^ (field_word & fmask) */
typedef struct {
  uint64_t c;
  uint64_t keep;
  uint64_t fmask;
  int32_t field;   /* -1: none; else index into the program's field pointer table ([count][32] arrays) */
  int8_t q;        /* 64-bit word of the field holding the lowest selected byte, -1..3             */
  uint8_t r;       /* byte rotation 0..7                                                            */
  uint16_t pad;
} afx_hash_word;

#define AFX_SQ_NONE 0
#define AFX_SQ_CHALLENGE_COMPARE 1   /* reduce 64 bytes mod l, compare with challenge[item]        */
#define AFX_SQ_SCALAR_OUT 2          /* reduce 64 bytes mod l, store to outs[squeeze_out][item]    */
#define AFX_SQ_WIDE_OUT 3            /* store the 64 bytes as they are to outs[squeeze_out] ([count][64]): a small pass reduces all of a
                                        transcript's blindings in ONE k_reduce_wide launch behind the hash, a lane each, instead of one
                                        after the other on the sponge's lane between its permutations (Assembler::hash)          */

typedef struct {
  afx_hash_word w[21];   /* bytes 0..167 of the block: rate (166) + the two pad bytes            */
  uint32_t squeeze;      /* AFX_SQ_* applied after the permutation                              */
  uint32_t squeeze_out;
} afx_hash_record;

typedef struct {
  const uint64_t* init_state;      /* 25 words, uniform; or null                                */
  const uint64_t* load_state;      /* per item SoA [25][count]; or null                         */
  uint64_t* save_state;            /* per item SoA [25][count]; or null                         */
  uint32_t n_records;
  const afx_hash_record* records;
  const uint8_t* const* fields;    /* device table of [count][32] array pointers                */
  uint8_t* const* outs;            /* device table of [count][32] output arrays                 */
  const uint8_t* challenge;        /* [count][32] for AFX_SQ_CHALLENGE_COMPARE                  */
  uint8_t* trace;                  /* optional [count][32]: the recomputed challenge is also stored here (parity aid) */
  uint32_t n_fields, n_outs;       /* host only: entries of the `fields` / `outs` pointer tables (Plan::relocate walks them)  */
} afx_hash_program;
