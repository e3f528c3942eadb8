// TEST INFRASTRUCTURE: a fake HIP runtime + no-op kernel launchers, so the host half of the engine
// (context parsing, plan assembly, STROBE schedule compilation, C ABI argument handling) can be exercised
// on a machine without a GPU under ASan/UBSan.  Nothing here computes results: every "kernel" is a no-op
// and status bytes come back as whatever malloc'ed memory held.  Never shipped, never loaded by the product.
#include <hip/hip_runtime_api.h>
#include <stdlib.h>
#include <string.h>
#include "../../aeonflux_amd/csrc/kernels.h"

extern "C" {
// AFX_FAKE_HIP_DEVICES: how many devices the fake runtime reports (default 1)
hipError_t hipGetDeviceCount(int* n) { const char* d = getenv("AFX_FAKE_HIP_DEVICES"); *n = d ? atoi(d) : 1; return hipSuccess; }
hipError_t hipDeviceGetAttribute(int* v, hipDeviceAttribute_t, int) { *v = 256; return hipSuccess; }
hipError_t hipSetDevice(int) { return hipSuccess; }
// AFX_FAKE_HIP_MAX_ALLOC (bytes) makes larger single allocations fail, to exercise the engine's out-of-memory handling
hipError_t hipMalloc(void** p, size_t n) {
  const char* lim = getenv("AFX_FAKE_HIP_MAX_ALLOC");
  if (lim && n > strtoull(lim, nullptr, 10)) { *p = nullptr; return hipErrorOutOfMemory; }
  *p = calloc(1, n ? n : 1);
  return *p ? hipSuccess : hipErrorOutOfMemory;
}
hipError_t hipGetLastError(void) { return hipSuccess; }
hipError_t hipFree(void* p) { free(p); return hipSuccess; }
hipError_t hipHostMalloc(void** p, size_t n, unsigned) { *p = calloc(1, n ? n : 1); return hipSuccess; }
hipError_t hipHostFree(void* p) { free(p); return hipSuccess; }
hipError_t hipMemset(void* p, int v, size_t n) { memset(p, v, n); return hipSuccess; }
hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t) { memmove(d, s, n); return hipSuccess; }
hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) { memmove(d, s, n); return hipSuccess; }
hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t) { memset(d, v, n); return hipSuccess; }
hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) { *s = (hipStream_t)0x1; return hipSuccess; }
hipError_t hipStreamDestroy(hipStream_t) { return hipSuccess; }
hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
hipError_t hipDeviceSynchronize(void) { return hipSuccess; }
hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { *e = (hipEvent_t)0x1; return hipSuccess; }
hipError_t hipEventDestroy(hipEvent_t) { return hipSuccess; }
hipError_t hipEventCreate(hipEvent_t* e) { *e = (hipEvent_t)0x1; return hipSuccess; }
hipError_t hipEventElapsedTime(float* ms, hipEvent_t, hipEvent_t) { *ms = 0; return hipSuccess; }
hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return hipSuccess; }
hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
const char* hipGetErrorString(hipError_t) { return "fake"; }
}

// walk the job arrays like the kernels would, touching every pointer-sized field (ASan catches bad blobs)
static thread_local volatile uintptr_t sink;   // per thread: the group tests drive several contexts from several threads
hipError_t afxk_setup_generators(hipStream_t, const uint8_t*, uint32_t ngen, int32_t*, uint8_t*, uint32_t* ok) {
  for (uint32_t i = 0; i < ngen; i++) ok[i] = 1;
  return hipSuccess;
}
hipError_t afxk_decode(hipStream_t, const afx_decode_job* j, uint32_t n, uint32_t*, uint32_t) { for (uint32_t i = 0; i < n; i++) sink += (uintptr_t)j[i].enc; return hipSuccess; }
hipError_t afxk_sccheck(hipStream_t, const afx_sccheck_job* j, uint32_t n, uint32_t*, uint32_t) { for (uint32_t i = 0; i < n; i++) sink += (uintptr_t)j[i].sc; return hipSuccess; }
hipError_t afxk_pointop(hipStream_t, const afx_pointop_job* j, uint32_t n, uint32_t*, uint32_t) { for (uint32_t i = 0; i < n; i++) sink += (uintptr_t)j[i].a; return hipSuccess; }
hipError_t afxk_scalarop(hipStream_t, const afx_scalarop_job* j, uint32_t n, uint32_t) { for (uint32_t i = 0; i < n; i++) sink += (uintptr_t)j[i].a; return hipSuccess; }
hipError_t afxk_setup_postables(hipStream_t, const int32_t*, uint32_t, int32_t*, int32_t*, int) { return hipSuccess; }
hipError_t afxk_msm_tables(hipStream_t, int kind, const afx_table_job* r, uint32_t n, int32_t*, uint32_t) {
  if (kind < 0 || kind > 2) return hipErrorInvalidValue;
  for (uint32_t i = 0; i < n; i++) sink += (uintptr_t)r[i].var + r[i].table_slot;
  return hipSuccess;
}
hipError_t afxk_msm(hipStream_t, int kind, int encodes, int secret, const afx_msm_job* j, uint32_t n, const int32_t*, const int32_t* sec_tables, int32_t*, uint32_t*, uint32_t*, uint32_t,
                    unsigned long long* probe) {
  if (probe) { probe[0] += 2250; probe[1] += 100; }
  if (secret && !sec_tables) return hipErrorInvalidValue;
  int any_secret = 0;
  for (uint32_t i = 0; i < n; i++)
    for (uint32_t t = 0; t < j[i].n_terms; t++) any_secret |= j[i].term[t].secret != 0;
  if (any_secret != (secret != 0)) return hipErrorInvalidValue;   // the launch's flag is the OR of its terms' flags
  if (kind < 0 || kind > 2) return hipErrorInvalidValue;
  for (uint32_t i = 0; i < n; i++) {
    if (!encodes && j[i].out_enc && !j[i].half_var) return hipErrorInvalidValue;   // a job that encodes in the kernel needs the encoding launch
    if ((j[i].n_var == 0 ? 0 : j[i].n_uni ? 2 : 1) != kind && !(j[i].n_var == 0 && kind == 1)) return hipErrorInvalidValue;   // every job in its own class's launch (fixed-only jobs may ride in the windowed one)
    if (j[i].leave_half && (j[i].out_var || !j[i].half_var)) return hipErrorInvalidValue;    // a job that leaves its half stores only the half
    for (uint32_t t = 0; t < j[i].n_uni; t++) if (j[i].term[t].dbl) return hipErrorInvalidValue;   // NAF schedules never run on a half base
    int secret_var = 0;
    for (uint32_t t = 0; t < j[i].n_var; t++) secret_var |= j[i].term[t].secret != 0;
    if ((j[i].narrow != 0) != (secret_var != 0) || (j[i].narrow && kind != 1)) return hipErrorInvalidValue;   // narrow windows exactly where a variable base carries a secret
    for (uint32_t t = 0; t < j[i].n_terms; t++) sink += (uintptr_t)j[i].term[t].scalar;
    if (j[i].n_uni) {   // the NAF schedule lives in the plan blob: read it to its terminator
      for (const uint32_t* e = j[i].naf_sched; ; e++) { sink += *e; if (*e == 0xffffffffu) break; }
    }
  }
  return hipSuccess;
}
hipError_t afxk_negenc(hipStream_t, const afx_negenc_job* j, uint32_t n, int32_t* ws, uint32_t*, uint32_t count) {
  for (uint32_t i = 0; i < n; i++) { sink += (uintptr_t)j[i].enc + (uintptr_t)j[i].var + (uintptr_t)j[i].out_enc; ws[(size_t)i * 9 * count] = 1; ws[((size_t)i * 9 + 8) * count + count - 1] = 1; }
  return hipSuccess;
}
hipError_t afxk_pointsum(hipStream_t, const afx_pointsum_job* j, uint32_t n, uint32_t*, uint32_t) {
  for (uint32_t i = 0; i < n; i++) {
    if (j[i].n_parts < 2 || (!j[i].out_var && !j[i].half_var && !j[i].out_enc)) return hipErrorInvalidValue;
    for (uint32_t k = 0; k < j[i].n_parts; k++) sink += (uintptr_t)j[i].parts[k];
  }
  return hipSuccess;
}
hipError_t afxk_compress2x(hipStream_t, const afx_compress_job* j, uint32_t n, uint32_t per_row, int32_t* ws, uint32_t*, uint32_t count) {
  if (per_row > n) return hipErrorInvalidValue;
  for (uint32_t i = 0; i < n; i++) { sink += (uintptr_t)j[i].var + (uintptr_t)j[i].out_enc; ws[(size_t)i * 9 * count] = 1; ws[((size_t)i * 9 + 8) * count + count - 1] = 1; }
  return hipSuccess;
}
hipError_t afxk_hash(hipStream_t, const afx_hash_program* p, uint32_t n, uint32_t*, uint32_t) {
  for (uint32_t i = 0; i < n; i++)
    for (uint32_t r = 0; r < p[i].n_records; r++)
      for (int w = 0; w < 21; w++)
        if (p[i].records[r].w[w].field >= 0) sink += (uintptr_t)p[i].fields[p[i].records[r].w[w].field];
  return hipSuccess;
}
hipError_t afxk_hash_coop(hipStream_t s, const afx_hash_program* p, uint32_t n, uint32_t* bad, uint32_t count) {
  if ((uint64_t)count * n > AFX_HASH_COOP_GROUPS) return hipErrorInvalidValue;   // only small passes hash with a lane group per item
  return afxk_hash(s, p, n, bad, count);
}
hipError_t afxk_finish(hipStream_t, const uint32_t*, uint8_t* status, uint32_t count, uint32_t, uint8_t) { memset(status, 0x5a, count); return hipSuccess; }
hipError_t afxk_fill_u32(hipStream_t, uint32_t* p, uint32_t v, uint32_t n) { for (uint32_t i = 0; i < n; i++) p[i] = v; return hipSuccess; }
hipError_t afxk_from_uniform(hipStream_t, const uint8_t*, uint8_t*, int32_t*, uint32_t) { return hipSuccess; }
hipError_t afxk_reduce_wide(hipStream_t, const uint8_t*, uint8_t*, uint32_t) { return hipSuccess; }
hipError_t afxk_validate(hipStream_t, const uint8_t*, uint8_t*, uint8_t*, uint32_t) { return hipSuccess; }
hipError_t afxk_aos_to_soa(hipStream_t, const uint8_t*, uint8_t*, const uint32_t* m, uint32_t cells, uint32_t) { for (uint32_t i = 0; i < cells; i++) sink += m[i]; return hipSuccess; }
