// Launch wrappers around the HIP kernels in kernels.hip (see that file for what each kernel does).
#pragma once
#include <hip/hip_runtime_api.h>
#include "plan.h"

hipError_t afxk_setup_generators(hipStream_t s, const uint8_t* enc, uint32_t ngen, int32_t* ext, uint8_t* neg_enc, uint32_t* ok);
// Plan launches: grid row r runs job jobs[r] of pass passes[0] (rows == null: a plan's own launch), or the job at byte offset
// rows[r].job_off from `jobs` of pass passes[rows[r].pass] (a launch merged from several plans; plan.h afx_row, afx_pass).
// `max_count` = the largest item count among the launch's passes (sizes the grid).
// mixed: the launch may hold Elligator jobs (afx_decode_job.elligator; Launch::odd)
hipError_t afxk_decode(hipStream_t s, const afx_decode_job* jobs, uint32_t njobs, const afx_row* rows, const afx_pass* passes, uint32_t max_count, int mixed);
hipError_t afxk_sccheck(hipStream_t s, const afx_sccheck_job* jobs, uint32_t njobs, const afx_row* rows, const afx_pass* passes, uint32_t max_count);
hipError_t afxk_pointop(hipStream_t s, const afx_pointop_job* jobs, uint32_t njobs, const afx_row* rows, const afx_pass* passes, uint32_t max_count);
hipError_t afxk_scalarop(hipStream_t s, const afx_scalarop_job* jobs, uint32_t njobs, const afx_row* rows, const afx_pass* passes, uint32_t max_count);
// secret != 0: the 6-bit tables of the secret-independent path (AFX_SEC_*) instead of the 13-bit ones
hipError_t afxk_setup_postables(hipStream_t s, const int32_t* ext, uint32_t ngen, int32_t* base_scratch, int32_t* postab, int secret);
// `variants` (afxk_msm, afxk_pointsum, afxk_hash_coop): the context's afx_ctx_set_plan_variants bits (plan.h AFX_KV_*) - which of two
// equivalent kernels a small launch takes when the automatic choice is overridden (tests).  A plan may only hold cached narrow tables
// (afx_msm_job.narrow == 2) while AFX_KV_ONE_WAVE_CHAINS is off.
// kind of table (plan.h afx_table_job): 0 multiples 1..8, 1 odd multiples 1..15 (NAF terms), 2 the short tables of narrow jobs,
// 3 the same in the cached form (no k_table_affine step: segmenting passes)
hipError_t afxk_msm_tables(hipStream_t s, int kind, const afx_table_job* jobs, uint32_t nrows, const afx_row* rows, const afx_pass* passes, uint32_t max_count);
// kind: 0 fixed bases only, 1 per-item windows, 2 uniform NAF terms (kernels.hip MSM_*)
// clock_probe: AFX_CLOCK_SLOTS pairs of 64-bit counters (shader-clock cycles, 100 MHz ticks): lane 0 of that many blocks of the launch adds its
// chain's span to its pair (kernels.hip msm_body); may be null
// secret: bit 0 - some term of the launch has afx_msm_term.secret set (sec_tables must then be the context's AFX_SEC_* tables);
//         bit 1 - some job's narrow tables hold cached entries (afx_msm_job.narrow == 2): the launch takes the four-wave chains at any size
// rows == null (a plan's own launch): pass_host = the HOST copy of the plan's pass, whose fields go as kernel arguments; merged launches: kinds 0 and 1 only
hipError_t afxk_msm(hipStream_t s, int kind, int encodes, int secret, const afx_msm_djob* jobs, uint32_t njobs, const int32_t* pos_tables, const int32_t* sec_tables,
                    const afx_row* rows, const afx_pass* passes, const afx_pass* pass_host, uint32_t max_count, unsigned long long* clock_probe, uint32_t variants);
// out_enc = encoding of twice each job's point; every row (plan.h afx_walk_row) shares one field inversion per item
// the tables of narrow jobs (secret scalars on per-item bases), second step: X, Y, Z -> affine entries, one inversion per item and row
hipError_t afxk_table_affine(hipStream_t s, const afx_table_job* jobs, const afx_walk_row* rows, uint32_t nrows, const afx_pass* passes, uint32_t max_count);
hipError_t afxk_compress2x(hipStream_t s, const afx_compress_job* jobs, const afx_walk_row* rows, uint32_t nrows, const afx_pass* passes, uint32_t max_count);
// out_enc = encoding of the negation of each job's decoded point
hipError_t afxk_negenc(hipStream_t s, const afx_negenc_job* jobs, const afx_walk_row* rows, uint32_t nrows, const afx_pass* passes, uint32_t max_count);
// many_parts: some job of the launch sums sixteen parts or more (Launch::odd): few items then take a lane per part (k_pointsum_tree)
hipError_t afxk_pointsum(hipStream_t s, const afx_pointsum_job* jobs, uint32_t njobs, const afx_row* rows, const afx_pass* passes, uint32_t max_count, int many_parts, uint32_t variants);
// out[i] = 2^(step (i + 1)) * src: the segment bases of a small prover pass (plan.h afx_powers_job)
hipError_t afxk_powers(hipStream_t s, const afx_powers_job* jobs, uint32_t njobs, const afx_row* rows, const afx_pass* passes, uint32_t max_count);
hipError_t afxk_hash(hipStream_t s, const afx_hash_program* progs, uint32_t nprogs, const afx_row* rows, const afx_pass* passes, uint32_t max_count);
// the same programs with 32 lanes per (item, program): small passes, where one lane's serial permutations are what a call waits for
hipError_t afxk_hash_coop(hipStream_t s, const afx_hash_program* progs, uint32_t nprogs, const afx_row* rows, const afx_pass* passes, uint32_t max_count, uint32_t variants);
hipError_t afxk_finish(hipStream_t s, const afx_finish_job* jobs, uint32_t njobs, const afx_row* rows, uint32_t max_count);
hipError_t afxk_fill_u32(hipStream_t s, const afx_fill_job* jobs, uint32_t njobs, const afx_row* rows, uint32_t max_n);
hipError_t afxk_from_uniform_jobs(hipStream_t s, const afx_uniform_job* jobs, uint32_t njobs, const afx_row* rows, const afx_pass* passes, uint32_t max_count);
hipError_t afxk_reduce_wide_jobs(hipStream_t s, const afx_reduce_job* jobs, uint32_t njobs, const afx_row* rows, const afx_pass* passes, uint32_t max_count);
// direct forms (the batch primitives of statements.cpp / statements_setup.cpp)
hipError_t afxk_from_uniform(hipStream_t s, const uint8_t* wide, uint8_t* out_enc, int32_t* out_var, uint32_t count);
hipError_t afxk_reduce_wide(hipStream_t s, const uint8_t* wide, uint8_t* out, uint32_t count);
hipError_t afxk_validate(hipStream_t s, const uint8_t* enc, uint8_t* ok, uint8_t* reenc, uint32_t count);
hipError_t afxk_aos_to_soa(hipStream_t s, const uint8_t* rec, uint8_t* soa, const uint32_t* row_of_cell, uint32_t cells, uint32_t count);
