// TEST INFRASTRUCTURE: a fake HIP runtime + no-op kernel launchers, so the host half of the engine
// (context parsing, plan assembly, STROBE schedule compilation, C ABI argument handling) can be exercised
// on a machine without a GPU under ASan/UBSan.  Nothing here computes results: every "kernel" is a no-op
// and status bytes come back as whatever malloc'ed memory held.  Never shipped, never loaded by the product.
#include <hip/hip_runtime_api.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sched.h>
#include <time.h>
#include <atomic>
#include <map>
#include <mutex>
#include "../../aeonflux_amd/csrc/kernels.h"

// Knobs a test sets WHILE other threads are inside the library (environment variables cannot be changed then):
//   sync_us    every hipStreamSynchronize sleeps this long - a launch set "computes" for a while, so that calls of other threads
//              arrive meanwhile and are collected (afx_ctx::co)
//   fail_next  the next `fail_next` finishing launches fail (a flush that fails with calls of several threads in it)
//   echo       a pass's status bytes are the first byte of each item of the first scalar array the pass checks (the challenge
//              row of a verification): a test sees that every caller gets the result of ITS OWN rows
static std::atomic<int> g_sync_us{ 0 }, g_fail_next{ 0 }, g_echo{ 0 };
extern "C" void afx_fake_set(const char* what, int v) {
  if (!strcmp(what, "sync_us")) g_sync_us = v;
  else if (!strcmp(what, "fail_next")) g_fail_next = v;
  else if (!strcmp(what, "echo")) g_echo = v;
}
// the CPUs the thread that launched a device's latest finishing kernel was allowed on (bit k = CPU k, the first 64): a test reads
// where a group's member threads ran (group.cpp PinScope)
static std::atomic<unsigned long long> g_affinity[64];
extern "C" unsigned long long afx_fake_affinity(int device) { return device >= 0 && device < 64 ? g_affinity[device].load() : 0; }
static std::mutex echo_mu;   // (a global under a mutex: launches of two lanes run on two threads)
static std::map<const void*, const uint8_t*> echo_src;   // pass (its failure words) -> first scalar array checked

extern "C" {
// AFX_FAKE_HIP_DEVICES: how many devices the fake runtime reports (default 1)
hipError_t hipGetDeviceCount(int* n) { const char* d = getenv("AFX_FAKE_HIP_DEVICES"); *n = d ? atoi(d) : 1; return hipSuccess; }
hipError_t hipDeviceGetAttribute(int* v, hipDeviceAttribute_t, int) { *v = 256; return hipSuccess; }
static thread_local int cur_device = 0;
hipError_t hipSetDevice(int d) { cur_device = d; return hipSuccess; }
hipError_t hipDeviceGetPCIBusId(char* out, int len, int d) { snprintf(out, (size_t)len, "0000:%02x:00.0", d & 0xff); return hipSuccess; }
// AFX_FAKE_HIP_FAIL_DEVICE: the finishing launch of every plan on that device fails (a member of a group that breaks mid-call)
static bool device_fails() { const char* f = getenv("AFX_FAKE_HIP_FAIL_DEVICE"); return f && atoi(f) == cur_device; }
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
hipError_t hipStreamSynchronize(hipStream_t) {
  static const int env_us = [] { const char* e = getenv("AFX_FAKE_HIP_SYNC_US"); return e ? atoi(e) : 0; }();   // (for drivers that cannot call afx_fake_set)
  const int us = g_sync_us.load() ? g_sync_us.load() : env_us;
  if (us > 0) { struct timespec ts = { us / 1000000, (long)(us % 1000000) * 1000 }; nanosleep(&ts, nullptr); }
  return hipSuccess;
}
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
// Every plan launch comes with its passes and (merged launches) its row table: the fake kernels resolve each grid row to its job the
// way the real ones do (plan.h afx_row) and touch every pointer-sized field - a row table or a pass that points outside the blob,
// or a job pointer that escaped relocation (plans are assembled against non-canonical provisional addresses), faults here under ASan.
template <class T>
static const T& job_of(const T* jobs, const afx_row* rows, uint32_t r) { return rows ? *(const T*)((const uint8_t*)jobs + rows[r].job_off) : jobs[r]; }
static const afx_pass& pass_of(const afx_pass* passes, const afx_row* rows, uint32_t r) { return passes[rows ? rows[r].pass : 0]; }
static bool canonical(const void* p) { return ((uintptr_t)p >> 47) == 0; }   // provisional plan addresses have bit 62 set
#define CHECK_PTR(p) do { if (!canonical(p)) return hipErrorInvalidDevicePointer; sink += (uintptr_t)(p); } while (0)
static hipError_t check_pass(const afx_pass& P, uint32_t max_count) {
  if (P.count == 0 || P.count > max_count) return hipErrorInvalidValue;
  if (!canonical(P.bad) || !canonical(P.table_ws) || !canonical(P.digit_ws)) return hipErrorInvalidDevicePointer;
  P.bad[0] |= 0; P.bad[P.count - 1] |= 0;   // the failure words are real memory of the pass's workspace
  return hipSuccess;
}
hipError_t afxk_decode(hipStream_t, const afx_decode_job* j, uint32_t n, const afx_row* rows, const afx_pass* passes, uint32_t max_count, int mixed) {
  uint32_t maps = 0;
  for (uint32_t i = 0; i < n; i++) {
    hipError_t e = check_pass(pass_of(passes, rows, i), max_count); if (e) return e; CHECK_PTR(job_of(j, rows, i).enc); CHECK_PTR(job_of(j, rows, i).out);
    const afx_decode_job& d = job_of(j, rows, i);
    if (d.elligator > 2 || (d.elligator && (!mixed || !d.out))) return hipErrorInvalidValue;   // Elligator jobs only in launches marked for the kernel that knows them
    maps += d.elligator != 0;
  }
  if (mixed && !maps) return hipErrorInvalidValue;   // ... and a launch marked for it holds some (the mark travels with the jobs' own field)
  return hipSuccess;
}
hipError_t afxk_sccheck(hipStream_t, const afx_sccheck_job* j, uint32_t n, const afx_row* rows, const afx_pass* passes, uint32_t max_count) {
  std::map<const void*, const uint8_t*> first;   // per pass: the first array this launch checks (a launch set that failed leaves nothing behind)
  for (uint32_t i = 0; i < n; i++) {
    hipError_t e = check_pass(pass_of(passes, rows, i), max_count); if (e) return e;
    CHECK_PTR(job_of(j, rows, i).sc);
    first.emplace(pass_of(passes, rows, i).bad, job_of(j, rows, i).sc);
  }
  if (g_echo.load()) { std::lock_guard<std::mutex> lk(echo_mu); for (auto& kv : first) echo_src[kv.first] = kv.second; }
  return hipSuccess;
}
hipError_t afxk_pointop(hipStream_t, const afx_pointop_job* j, uint32_t n, const afx_row* rows, const afx_pass* passes, uint32_t max_count) {
  for (uint32_t i = 0; i < n; i++) {
    hipError_t e = check_pass(pass_of(passes, rows, i), max_count); if (e) return e;
    const afx_pointop_job& q = job_of(j, rows, i);
    CHECK_PTR(q.a); CHECK_PTR(q.b); CHECK_PTR(q.b_const); CHECK_PTR(q.out); CHECK_PTR(q.out_enc);
  }
  return hipSuccess;
}
hipError_t afxk_scalarop(hipStream_t, const afx_scalarop_job* j, uint32_t n, const afx_row* rows, const afx_pass* passes, uint32_t max_count) {
  for (uint32_t i = 0; i < n; i++) {
    hipError_t e = check_pass(pass_of(passes, rows, i), max_count); if (e) return e;
    const afx_scalarop_job& q = job_of(j, rows, i);
    CHECK_PTR(q.a); CHECK_PTR(q.b); CHECK_PTR(q.c); CHECK_PTR(q.out);
  }
  return hipSuccess;
}
hipError_t afxk_setup_postables(hipStream_t, const int32_t*, uint32_t, int32_t*, int32_t*, int) { return hipSuccess; }
hipError_t afxk_msm_tables(hipStream_t, int kind, const afx_table_job* j, uint32_t n, const afx_row* rows, const afx_pass* passes, uint32_t max_count) {
  if (kind < 0 || kind > 3) return hipErrorInvalidValue;
  for (uint32_t i = 0; i < n; i++) { hipError_t e = check_pass(pass_of(passes, rows, i), max_count); if (e) return e; CHECK_PTR(job_of(j, rows, i).var); sink += job_of(j, rows, i).table_slot; }
  return hipSuccess;
}
hipError_t afxk_msm(hipStream_t, int kind, int encodes, int secret, const afx_msm_djob* jobs, uint32_t n, const int32_t*, const int32_t* sec_tables, const afx_row* rows,
                    const afx_pass* passes, const afx_pass* pass_host, uint32_t max_count, unsigned long long* probe, uint32_t variants) {
  if (!rows) {   // a plan's own launch: the pass travels as kernel arguments, from its host copy - which must equal the device's
    if (!pass_host || memcmp(pass_host, passes, sizeof(afx_pass)) != 0) return hipErrorInvalidValue;
  } else if (kind == 2) return hipErrorInvalidValue;   // no merged NAF launches
  if (probe) { probe[0] += 2250; probe[1] += 100; }
  if (secret && !sec_tables) return hipErrorInvalidValue;
  int any_secret = 0;
  for (uint32_t i = 0; i < n; i++)
    for (uint32_t t = 0; t < job_of(jobs, rows, i).n_terms; t++) any_secret |= afx_job_terms(&job_of(jobs, rows, i))[t].secret != 0;
  if (any_secret != (secret != 0)) return hipErrorInvalidValue;   // the launch's flag is the OR of its terms' flags
  int any_cached = 0;
  for (uint32_t i = 0; i < n; i++) any_cached |= job_of(jobs, rows, i).narrow == 2;
  if (any_cached != ((secret & 2) != 0) || (any_cached && (encodes || kind != 1))) return hipErrorInvalidValue;   // ... bit 1 of its jobs' cached tables
  if (kind < 0 || kind > 2) return hipErrorInvalidValue;
  for (uint32_t i = 0; i < n; i++) {
    hipError_t e = check_pass(pass_of(passes, rows, i), max_count); if (e) return e;
    const afx_msm_djob& j = job_of(jobs, rows, i);
    const afx_msm_term* term = afx_job_terms(&j);
    if (!encodes && j.out_enc && !j.half_var) return hipErrorInvalidValue;   // a job that encodes in the kernel needs the encoding launch
    if ((j.n_var == 0 ? 0 : j.n_uni ? 2 : 1) != kind && !(j.n_var == 0 && kind == 1)) return hipErrorInvalidValue;   // every job in its own class's launch (fixed-only jobs may ride in the windowed one)
    if (j.leave_half && (j.out_var || !j.half_var)) return hipErrorInvalidValue;    // a job that leaves its half stores only the half
    for (uint32_t t = 0; t < j.n_uni; t++) if (term[t].dbl) return hipErrorInvalidValue;   // NAF schedules never run on a half base
    int secret_var = 0;
    for (uint32_t t = 0; t < j.n_var; t++) secret_var |= term[t].secret != 0;
    if ((j.narrow != 0) != (secret_var != 0) || (j.narrow && kind != 1)) return hipErrorInvalidValue;   // narrow windows exactly where a variable base carries a secret
    for (uint32_t t = 0; t < j.n_terms; t++) { CHECK_PTR(term[t].scalar); CHECK_PTR(term[t].var); }
    // a chain's windows: all of them, or a segment's share with every variable term a segment inside the recoded scalar
    const uint32_t all_wins = j.narrow ? AFX_SECVAR_WINDOWS : 64u;   // 2-bit windows of a secret on a per-item base, 4-bit ones otherwise
    if (j.wins == 0 || j.wins > all_wins) return hipErrorInvalidValue;
    for (uint32_t t = 0; t < j.n_terms; t++) {
      if (term[t].win_off && !(t < j.n_var && t >= j.n_uni)) return hipErrorInvalidValue;
      if (t < j.n_var && term[t].win_off + j.wins > all_wins) return hipErrorInvalidValue;
    }
    CHECK_PTR(j.addend); CHECK_PTR(j.out_enc); CHECK_PTR(j.out_var); CHECK_PTR(j.half_var);
    if (j.n_uni) {   // the NAF schedule lives in the plan blob: read it to its terminator
      for (const uint32_t* e2 = j.naf_sched; ; e2++) { sink += *e2; if (*e2 == 0xffffffffu) break; }
    }
  }
  return hipSuccess;
}
static hipError_t walk_rows(const uint8_t* jobs, size_t job_size, const afx_walk_row* rows, uint32_t nrows, const afx_pass* passes, uint32_t max_count, int kind) {
  for (uint32_t r = 0; r < nrows; r++) {
    const afx_pass& P = passes[rows[r].pass];
    hipError_t e = check_pass(P, max_count); if (e) return e;
    if (rows[r].n_jobs == 0 || !canonical(rows[r].prefix_ws)) return hipErrorInvalidValue;
    for (uint32_t i = 0; i < rows[r].n_jobs; i++) {
      const uint8_t* q = jobs + rows[r].job_off + (size_t)i * job_size;
      if (kind == 0) { const afx_compress_job* c = (const afx_compress_job*)q; CHECK_PTR(c->var); CHECK_PTR(c->out_enc); }
      else { const afx_negenc_job* c = (const afx_negenc_job*)q; CHECK_PTR(c->enc); CHECK_PTR(c->var); CHECK_PTR(c->out_enc); }
    }
    // the walk's scratch: n_jobs field elements per item
    rows[r].prefix_ws[0] = 1;
    rows[r].prefix_ws[(size_t)rows[r].n_jobs * 9 * P.count - 1] = 1;
  }
  return hipSuccess;
}
// the tables of narrow jobs made affine: every table of the row lies inside the pass's table workspace (its first and last dword are
// touched: the sanitizers see a slot past the allocation); the prefix products live inside the entries, so no scratch of its own
hipError_t afxk_table_affine(hipStream_t, const afx_table_job* jobs, const afx_walk_row* rows, uint32_t nrows, const afx_pass* passes, uint32_t max_count) {
  for (uint32_t r = 0; r < nrows; r++) {
    const afx_pass& P = passes[rows[r].pass];
    hipError_t e = check_pass(P, max_count); if (e) return e;
    if (rows[r].n_jobs == 0 || rows[r].prefix_ws != nullptr) return hipErrorInvalidValue;
    for (uint32_t i = 0; i < rows[r].n_jobs; i++) {
      const afx_table_job& t = *(const afx_table_job*)((const uint8_t*)jobs + rows[r].job_off + (size_t)i * sizeof(afx_table_job));
      CHECK_PTR(t.var);
      int32_t* slot = P.table_ws + (size_t)t.table_slot * P.count * AFX_VAR_TABLE_DWORDS;
      slot[0] += 1;
      slot[(size_t)AFX_SECVAR_STORED * P.count * AFX_TABLE_ENTRY_DWORDS - 1] += 1;
    }
  }
  return hipSuccess;
}
hipError_t afxk_negenc(hipStream_t, const afx_negenc_job* j, const afx_walk_row* rows, uint32_t nrows, const afx_pass* passes, uint32_t max_count) {
  return walk_rows((const uint8_t*)j, sizeof(afx_negenc_job), rows, nrows, passes, max_count, 1);
}
hipError_t afxk_compress2x(hipStream_t, const afx_compress_job* j, const afx_walk_row* rows, uint32_t nrows, const afx_pass* passes, uint32_t max_count) {
  return walk_rows((const uint8_t*)j, sizeof(afx_compress_job), rows, nrows, passes, max_count, 0);
}
hipError_t afxk_pointsum(hipStream_t, const afx_pointsum_job* jobs, uint32_t n, const afx_row* rows, const afx_pass* passes, uint32_t max_count, int, uint32_t) {
  for (uint32_t i = 0; i < n; i++) {
    hipError_t e = check_pass(pass_of(passes, rows, i), max_count); if (e) return e;
    const afx_pointsum_job& j = job_of(jobs, rows, i);
    if (j.n_parts < 2 || (!j.out_var && !j.half_var && !j.out_enc)) return hipErrorInvalidValue;
    for (uint32_t k = 0; k < j.n_parts; k++) CHECK_PTR(j.parts[k]);
    CHECK_PTR(j.addend); CHECK_PTR(j.out_var); CHECK_PTR(j.half_var); CHECK_PTR(j.out_enc);
  }
  return hipSuccess;
}
hipError_t afxk_powers(hipStream_t, const afx_powers_job* jobs, uint32_t n, const afx_row* rows, const afx_pass* passes, uint32_t max_count) {
  for (uint32_t i = 0; i < n; i++) {
    hipError_t e = check_pass(pass_of(passes, rows, i), max_count); if (e) return e;
    const afx_powers_job& j = job_of(jobs, rows, i);
    if (j.n_out == 0 || j.n_out > AFX_POWERS_MAX || j.step == 0 || j.step * (j.n_out + 1) > 2 * AFX_SECVAR_WINDOWS || !j.src) return hipErrorInvalidValue;
    CHECK_PTR(j.src);
    for (uint32_t k = 0; k < j.n_out; k++) { if (!j.out[k]) return hipErrorInvalidValue; CHECK_PTR(j.out[k]); }
  }
  return hipSuccess;
}
hipError_t afxk_hash(hipStream_t, const afx_hash_program* p, uint32_t n, const afx_row* rows, const afx_pass* passes, uint32_t max_count) {
  for (uint32_t i = 0; i < n; i++) {
    hipError_t e = check_pass(pass_of(passes, rows, i), max_count); if (e) return e;
    const afx_hash_program& q = job_of(p, rows, i);
    sink += q.init_state[0] + q.init_state[24];
    for (uint32_t r = 0; r < q.n_records; r++)
      for (int w = 0; w < 21; w++)
        if (q.records[r].w[w].field >= 0) { if ((uint32_t)q.records[r].w[w].field >= q.n_fields) return hipErrorInvalidValue; CHECK_PTR(q.fields[q.records[r].w[w].field]); }
    for (uint32_t k = 0; k < q.n_outs; k++) CHECK_PTR(q.outs[k]);
    CHECK_PTR(q.challenge);
  }
  return hipSuccess;
}
hipError_t afxk_hash_coop(hipStream_t s, const afx_hash_program* p, uint32_t n, const afx_row* rows, const afx_pass* passes, uint32_t max_count, uint32_t) {
  uint64_t groups = 0;
  for (uint32_t i = 0; i < n; i++) groups += pass_of(passes, rows, i).count;
  if (groups > AFX_HASH_COOP_GROUPS) return hipErrorInvalidValue;   // only small passes hash with a lane group per item
  return afxk_hash(s, p, n, rows, passes, max_count);
}
hipError_t afxk_finish(hipStream_t, const afx_finish_job* j, uint32_t n, const afx_row* rows, uint32_t max_count) {
  if (device_fails()) return hipErrorLaunchFailure;
  {
    cpu_set_t set;
    unsigned long long m = 0;
    if (cur_device >= 0 && cur_device < 64 && sched_getaffinity(0, sizeof set, &set) == 0) {
      for (int k = 0; k < 64; k++) if (CPU_ISSET(k, &set)) m |= 1ull << k;
      g_affinity[cur_device] = m;
    }
  }
  for (int f = g_fail_next.load(); f > 0; f = g_fail_next.load())
    if (g_fail_next.compare_exchange_strong(f, f - 1)) {
      std::lock_guard<std::mutex> lk(echo_mu);
      for (uint32_t i = 0; i < n; i++) echo_src.erase(job_of(j, rows, i).bad);
      return hipErrorLaunchFailure;
    }
  for (uint32_t i = 0; i < n; i++) {
    const afx_finish_job& q = job_of(j, rows, i);
    if (q.count == 0 || q.count > max_count || !canonical(q.bad) || !canonical(q.status)) return hipErrorInvalidValue;
    sink += q.bad[0] + q.bad[q.count - 1];
    memset(q.status, 0x5a, q.count);
    std::lock_guard<std::mutex> lk(echo_mu);
    auto src = echo_src.find(q.bad);
    if (src != echo_src.end()) {
      if (g_echo.load())
        for (uint32_t k = 0; k < q.count; k++) q.status[k] = src->second[32 * (size_t)k];
      echo_src.erase(src);
    }
  }
  return hipSuccess;
}
hipError_t afxk_fill_u32(hipStream_t, const afx_fill_job* j, uint32_t n, const afx_row* rows, uint32_t max_n) {
  for (uint32_t i = 0; i < n; i++) {
    const afx_fill_job& q = job_of(j, rows, i);
    if (q.n > max_n || !canonical(q.p)) return hipErrorInvalidValue;
    for (uint32_t k = 0; k < q.n; k++) q.p[k] = q.v;
    { std::lock_guard<std::mutex> lk(echo_mu); echo_src.erase(q.p); }   // a plan starts here: nothing an earlier (failed) launch set left behind at this address
  }
  return hipSuccess;
}
hipError_t afxk_from_uniform_jobs(hipStream_t, const afx_uniform_job* j, uint32_t n, const afx_row* rows, const afx_pass* passes, uint32_t max_count) {
  for (uint32_t i = 0; i < n; i++) { hipError_t e = check_pass(pass_of(passes, rows, i), max_count); if (e) return e; CHECK_PTR(job_of(j, rows, i).wide); CHECK_PTR(job_of(j, rows, i).out_enc); CHECK_PTR(job_of(j, rows, i).out_var);
  }
  return hipSuccess;
}
hipError_t afxk_reduce_wide_jobs(hipStream_t, const afx_reduce_job* j, uint32_t n, const afx_row* rows, const afx_pass* passes, uint32_t max_count) {
  for (uint32_t i = 0; i < n; i++) { hipError_t e = check_pass(pass_of(passes, rows, i), max_count); if (e) return e; CHECK_PTR(job_of(j, rows, i).wide); CHECK_PTR(job_of(j, rows, i).out); }
  return hipSuccess;
}
hipError_t afxk_from_uniform(hipStream_t, const uint8_t*, uint8_t*, int32_t*, uint32_t) { return hipSuccess; }
hipError_t afxk_reduce_wide(hipStream_t, const uint8_t*, uint8_t*, uint32_t) { return hipSuccess; }
hipError_t afxk_validate(hipStream_t, const uint8_t*, uint8_t*, uint8_t*, uint32_t) { return hipSuccess; }
// (the one fake kernel that moves data: records to rows, so that the echo knob also reaches the serialized front ends)
hipError_t afxk_aos_to_soa(hipStream_t, const uint8_t* rec, uint8_t* soa, const uint32_t* m, uint32_t cells, uint32_t n) {
  for (uint32_t i = 0; i < n; i++)
    for (uint32_t c = 0; c < cells; c++) memcpy(soa + ((size_t)m[c] * n + i) * 32, rec + ((size_t)i * cells + c) * 32, 32);
  return hipSuccess;
}
