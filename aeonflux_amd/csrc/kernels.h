// Launch wrappers around the HIP kernels in kernels.hip (see that file for what each kernel does).
#pragma once
#include <hip/hip_runtime_api.h>
#include "plan.h"

hipError_t afxk_setup_generators(hipStream_t s, const uint8_t* enc, uint32_t ngen, int32_t* ext, uint8_t* neg_enc, uint32_t* ok);
hipError_t afxk_decode(hipStream_t s, const afx_decode_job* jobs, uint32_t njobs, uint32_t* bad, uint32_t count);
hipError_t afxk_sccheck(hipStream_t s, const afx_sccheck_job* jobs, uint32_t njobs, uint32_t* bad, uint32_t count);
hipError_t afxk_pointop(hipStream_t s, const afx_pointop_job* jobs, uint32_t njobs, uint32_t* bad, uint32_t count);
hipError_t afxk_scalarop(hipStream_t s, const afx_scalarop_job* jobs, uint32_t njobs, uint32_t count);
// secret != 0: the 4-bit tables of the secret-independent path (AFX_SEC_*) instead of the 13-bit ones
hipError_t afxk_setup_postables(hipStream_t s, const int32_t* ext, uint32_t ngen, int32_t* base_scratch, int32_t* postab, int secret);
// kind of table (plan.h afx_table_job): 0 multiples 1..8, 1 odd multiples 1..15 (NAF terms), 2 the short tables of narrow jobs
hipError_t afxk_msm_tables(hipStream_t s, int kind, const afx_table_job* rows, uint32_t nrows, int32_t* table_ws, uint32_t count);
// kind: 0 fixed bases only, 1 per-item windows, 2 uniform NAF terms (kernels.hip MSM_*)
// clock_probe: two 64-bit counters (shader-clock cycles, 100 MHz ticks) one lane of the launch adds its chain's span to; may be null
// secret: some term of the launch has afx_msm_term.secret set (sec_tables must then be the context's 4-bit tables)
hipError_t afxk_msm(hipStream_t s, int kind, int encodes, int secret, const afx_msm_job* jobs, uint32_t njobs, const int32_t* pos_tables, const int32_t* sec_tables,
                    int32_t* table_ws, uint32_t* digit_ws, uint32_t* bad, uint32_t count, unsigned long long* clock_probe);
// out_enc = encoding of twice each job's point; prefix_ws: njobs * 9 * count dwords of scratch (one 9-limb field element per job and item)
// per_row: jobs per grid row, each row sharing one field inversion per item (0 = all jobs in one row)
hipError_t afxk_compress2x(hipStream_t s, const afx_compress_job* jobs, uint32_t njobs, uint32_t per_row, int32_t* prefix_ws, uint32_t* bad, uint32_t count);
// out_enc = encoding of the negation of each job's decoded point; prefix_ws: njobs * 9 * count dwords of scratch
hipError_t afxk_negenc(hipStream_t s, const afx_negenc_job* jobs, uint32_t njobs, int32_t* prefix_ws, uint32_t* bad, uint32_t count);
hipError_t afxk_pointsum(hipStream_t s, const afx_pointsum_job* jobs, uint32_t njobs, uint32_t* bad, uint32_t count);
hipError_t afxk_hash(hipStream_t s, const afx_hash_program* progs, uint32_t nprogs, uint32_t* bad, uint32_t count);
// the same programs with 32 lanes per (item, program): small passes, where one lane's serial permutations are what a call waits for
hipError_t afxk_hash_coop(hipStream_t s, const afx_hash_program* progs, uint32_t nprogs, uint32_t* bad, uint32_t count);
hipError_t afxk_finish(hipStream_t s, const uint32_t* bad, uint8_t* status, uint32_t count, uint32_t fail_all, uint8_t fail_code);
hipError_t afxk_fill_u32(hipStream_t s, uint32_t* p, uint32_t v, uint32_t n);
hipError_t afxk_from_uniform(hipStream_t s, const uint8_t* wide, uint8_t* out_enc, int32_t* out_var, uint32_t count);
hipError_t afxk_reduce_wide(hipStream_t s, const uint8_t* wide, uint8_t* out, uint32_t count);
hipError_t afxk_validate(hipStream_t s, const uint8_t* enc, uint8_t* ok, uint8_t* reenc, uint32_t count);
hipError_t afxk_aos_to_soa(hipStream_t s, const uint8_t* rec, uint8_t* soa, const uint32_t* row_of_cell, uint32_t cells, uint32_t count);
