// The aeonflux statements, batch form, and the C ABI of include/aeonflux_gpu.h.
//
// Each build_* function restates one reference method as a list of GPU launches over a batch; the
// transcript labels, allocation order and constraint lists are those of
//   /root/reference/src/nizk/presentation.rs:324-443 (verify) and :139-321 (prove),
//   /root/reference/src/nizk/encryption.rs:154-210 (verify) and :58-142 (prove),
//   /root/reference/src/nizk/issuance.rs:132-218 (verify) and :40-129 (prove),
//   /root/reference/src/amacs.rs:225-294 (Messages, Amac::tag / compute_V).
// No arithmetic happens on the host: the host only lays out launches and transcript byte schedules.
#include <string.h>
#include <algorithm>
#include <functional>
#include <memory>
#include <stdexcept>
#include <vector>
#include "engine.hpp"
#include <stdlib.h>
#include "statements.hpp"

using namespace afx;

static const uint8_t SC_L_BYTES[32] = { 0xed, 0xd3, 0xf5, 0x5c, 0x1a, 0x63, 0x12, 0x58, 0xd6, 0x9c, 0xf7, 0xa2, 0xde, 0xf9, 0xde, 0x14,
                                        0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0x10 };
static bool host_scalar_is_canonical(const uint8_t* s) {
  for (int i = 31; i >= 0; i--) {
    if (s[i] < SC_L_BYTES[i]) return true;
    if (s[i] > SC_L_BYTES[i]) return false;
  }
  return false;
}
static uint32_t rd32(const uint8_t* b) { return (uint32_t)b[0] | ((uint32_t)b[1] << 8) | ((uint32_t)b[2] << 16) | ((uint32_t)b[3] << 24); }
static size_t sizeof_system_parameters(uint32_t n) { return n < 3 ? 32 * (5 + 3 + (size_t)n + 4) + 4 : 32 * (5 + 2 * (size_t)n + 4) + 4; }  // parameters.rs:34-40
static size_t sizeof_secret_key(uint32_t n) { return 32 * (5 + (size_t)n) + 4; }                                                           // amacs.rs:44-46


extern "C" const char* afx_last_error(void) { return afx::last_error(); }
extern "C" uint32_t afx_ctx_n_attributes(const afx_ctx* ctx) { return ctx ? ctx->n : 0; }
extern "C" void* afx_ctx_stream(const afx_ctx* ctx) { return ctx ? (void*)ctx->stream : nullptr; }

// indexed by afx::LaunchKind.  "k_msm" (afx_ctx_get_timing) = the three chain kernels + the table kernel together
static const char* const KIND_NAMES[] = { "k_fill_u32", "k_decode", "k_sccheck", "k_pointop", "k_scalarop", "k_msm_window", "k_hash",
                                          "k_from_uniform", "k_reduce_wide", "copy", "k_finish", "k_msm_fixed", "k_msm_naf", "k_msm_tables", "k_compress2x", "k_pointsum", "k_negenc", "k_table_affine", "k_powers" };
static_assert(sizeof KIND_NAMES / sizeof KIND_NAMES[0] == afx::L_KINDS, "one name per launch kind");
static int drain_timing(afx_ctx* c) {
  for (auto& L : c->lane)
    if (L.stream) AFX_HIP(hipStreamSynchronize(L.stream));
  for (auto& t : c->timed) {
    float ms = 0;
    AFX_HIP(hipEventElapsedTime(&ms, t.start, t.stop));
    c->kind_ms[t.kind] += ms;
    c->kind_launches[t.kind] += 1;
    c->event_pool.push_back(t.start);
    c->event_pool.push_back(t.stop);
  }
  c->timed.clear();
  return AFX_OK;
}
extern "C" int afx_ctx_set_pipelining(afx_ctx* c, int enable) try {
  if (!c) { set_error("null argument"); return AFX_E_BAD_ARGS; }
  CtxLock lock(c);
  AFX_HIP(hipSetDevice(c->device));
  for (auto& L : c->lane)
    if (L.stream) AFX_HIP(hipStreamSynchronize(L.stream));
  c->pipelining = enable != 0;
  c->lane_next = 0;
  for (auto& L : c->lane) L.msm_recorded = false;
  return AFX_OK;
} catch (...) { return afx::exception_rc(); }
extern "C" int afx_ctx_set_challenge_trace(afx_ctx* c, size_t rows, size_t count) try {
  if (!c) { set_error("null argument"); return AFX_E_BAD_ARGS; }
  CtxLock lock(c);
  AFX_HIP(hipSetDevice(c->device));
  for (auto& L : c->lane)
    if (L.stream) AFX_HIP(hipStreamSynchronize(L.stream));
  { std::lock_guard<std::mutex> sg(c->settings_mu); c->trace = nullptr; }
  c->trace_rows = c->trace_count = 0;
  if (rows == 0 || count == 0) { c->trace_buf.release(false); return AFX_OK; }
  int rc = c->trace_buf.ensure(rows * count * 32);
  if (rc) return rc;
  AFX_HIP(hipMemsetAsync(c->trace_buf.p, 0, rows * count * 32, c->stream));
  { std::lock_guard<std::mutex> sg(c->settings_mu); c->trace = (uint8_t*)c->trace_buf.p; }
  c->trace_rows = rows;
  c->trace_count = count;
  return AFX_OK;
} catch (...) { return afx::exception_rc(); }
extern "C" int afx_ctx_get_challenge_trace(afx_ctx* c, uint8_t* host_out) try {
  if (!c || !host_out) { set_error("null argument"); return AFX_E_BAD_ARGS; }
  CtxLock lock(c);
  if (!c->trace) { set_error("challenge trace is off"); return AFX_E_BAD_ARGS; }
  AFX_HIP(hipSetDevice(c->device));
  for (auto& L : c->lane)
    if (L.stream) AFX_HIP(hipStreamSynchronize(L.stream));
  AFX_HIP(hipMemcpy(host_out, c->trace, c->trace_rows * c->trace_count * 32, hipMemcpyDeviceToHost));
  return AFX_OK;
} catch (...) { return afx::exception_rc(); }
extern "C" int afx_ctx_set_chunk_items(afx_ctx* c, uint32_t items) try {
  if (!c || (items != 0 && (items < 256 || items > (1u << 22)))) { set_error("chunk size out of range"); return AFX_E_BAD_ARGS; }
  CtxLock lock(c);
  { std::lock_guard<std::mutex> sg(c->settings_mu); c->chunk_items = items; }
  return AFX_OK;
} catch (...) { return afx::exception_rc(); }
extern "C" int afx_ctx_set_small_batch_items(afx_ctx* c, uint32_t items) try {
  if (!c || items > (1u << 16)) { set_error("small-batch threshold out of range"); return AFX_E_BAD_ARGS; }
  CtxLock lock(c);
  { std::lock_guard<std::mutex> sg(c->settings_mu); c->small_batch_items = items; }
  return AFX_OK;
} catch (...) { return afx::exception_rc(); }
extern "C" int afx_ctx_set_coalescing(afx_ctx* c, uint32_t max_wait_us, uint32_t max_items) try {
  if (!c || max_items > (1u << 16) || max_wait_us > 1000000u) { set_error("coalescing bounds out of range"); return AFX_E_BAD_ARGS; }
  CtxLock lock(c);   // (waits for the sessions in flight)
  c->co.enabled = max_items != 0;
  c->co.max_wait_us = max_wait_us;
  if (max_items) c->co.max_items = max_items;
  return AFX_OK;
} catch (...) { return afx::exception_rc(); }
extern "C" int afx_ctx_set_plan_variants(afx_ctx* c, uint32_t flags) try {
  const uint32_t seg = flags & (AFX_VARIANT_SEGMENTS_1 | AFX_VARIANT_SEGMENTS_2 | AFX_VARIANT_SEGMENTS_4);
  if (!c || (flags & ~(uint32_t)AFX_VARIANT_ALL) || (seg & (seg - 1))) { set_error("unknown plan variant flags (at most one AFX_VARIANT_SEGMENTS_*)"); return AFX_E_BAD_ARGS; }
  CtxLock lock(c);
  { std::lock_guard<std::mutex> sg(c->settings_mu); c->variants = flags & ~(uint32_t)AFX_VARIANT_SELFCHECK; c->plan_selfcheck = (flags & AFX_VARIANT_SELFCHECK) != 0; }
  return AFX_OK;
} catch (...) { return afx::exception_rc(); }
extern "C" int afx_ctx_set_host_copy_threads(afx_ctx* c, uint32_t threads) try {
  if (!c || threads > 64) { set_error("host copy threads out of range (0 .. 64)"); return AFX_E_BAD_ARGS; }
  CtxLock lock(c);
  AFX_HIP(hipSetDevice(c->device));
  for (auto& L : c->lane)
    if (L.stream) AFX_HIP(hipStreamSynchronize(L.stream));
  c->copy_pool.reset();   // (made again, with the new count, by the next large host-pointer call)
  c->host_copy_threads = threads;
  return AFX_OK;
} catch (...) { return afx::exception_rc(); }
extern "C" int afx_ctx_get_coalescing_stats(afx_ctx* c, afx_coalescing_stats* out) try {
  if (!c || !out) { set_error("null argument"); return AFX_E_BAD_ARGS; }
  CtxLock lock(c, true);   // (a reader: no need to wait for the sessions in flight)
  const afx_ctx::Coalesce& co = c->co;
  *out = afx_coalescing_stats{ co.n_sessions, co.n_calls, co.n_items, co.n_appended, co.n_max_calls, co.n_waited_flushes, co.staging_ns, co.launch_ns };
  return AFX_OK;
} catch (...) { return afx::exception_rc(); }
extern "C" int afx_ctx_set_strict(afx_ctx* c, int enable) try {
  if (!c) { set_error("null argument"); return AFX_E_BAD_ARGS; }
  CtxLock lock(c);
  { std::lock_guard<std::mutex> sg(c->settings_mu); c->strict = enable != 0; }
  return AFX_OK;
} catch (...) { return afx::exception_rc(); }
extern "C" int afx_ctx_set_fixed_key_schedule(afx_ctx* c, int enable) try {
  if (!c) { set_error("null argument"); return AFX_E_BAD_ARGS; }
  CtxLock lock(c);
  { std::lock_guard<std::mutex> sg(c->settings_mu); c->fixed_key_schedule = enable != 0; }
  return AFX_OK;
} catch (...) { return afx::exception_rc(); }
// the generators' 6-bit positional tables (AFX_SEC_*), once per context: window bases through lane 0's workspace, like the 13-bit ones
static int build_secret_tables(afx_ctx* c) {
  if (c->d_sec_tables.p || c->ngen == 0) return AFX_OK;
  AFX_HIP(hipSetDevice(c->device));
  for (auto& L : c->lane)
    if (L.stream) AFX_HIP(hipStreamSynchronize(L.stream));
  int rc;
  if ((rc = c->d_sec_tables.ensure(sizeof(int32_t) * AFX_SEC_TABLE_DWORDS * (size_t)c->ngen)) ||
      (rc = c->lane[0].ws.ensure(sizeof(int32_t) * AFX_VAR_DWORDS * (size_t)c->ngen * AFX_SEC_WINDOWS)))
    return rc;
  AFX_HIP(afxk_setup_postables(c->stream, (const int32_t*)c->d_gen_ext.p, c->ngen, (int32_t*)c->lane[0].ws.p, (int32_t*)c->d_sec_tables.p, 1));
  AFX_HIP(hipStreamSynchronize(c->stream));
  return AFX_OK;
}
extern "C" int afx_ctx_set_secret_independent_addressing(afx_ctx* c, int mode) try {
  if (!c) { set_error("null argument"); return AFX_E_BAD_ARGS; }
  if (mode < 0 || mode > 2) { set_error("mode must be 0 (nowhere), 1 (everywhere) or 2 (the prover-side calls: the default)"); return AFX_E_BAD_ARGS; }
  CtxLock lock(c);
  if (mode) { const int rc = build_secret_tables(c); if (rc) return rc; }
  { std::lock_guard<std::mutex> sg(c->settings_mu); c->secret_mode = mode; }
  return AFX_OK;
} catch (...) { return afx::exception_rc(); }
extern "C" int afx_ctx_get_plan_stats(afx_ctx* c, afx_plan_stats* out) try {
  if (!c || !out) { set_error("null argument"); return AFX_E_BAD_ARGS; }
  CtxLock lock(c);
  *out = c->last_stats;
  return AFX_OK;
} catch (...) { return afx::exception_rc(); }
extern "C" int afx_ctx_get_plan_cache_stats(afx_ctx* c, afx_plan_cache_stats* out) try {
  if (!c || !out) { set_error("null argument"); return AFX_E_BAD_ARGS; }
  CtxLock lock(c, true);
  *out = afx_plan_cache_stats{ c->plan_cache_hits, c->plan_cache_misses, c->plan_cache_evictions, (uint64_t)c->plan_cache.size(), (uint64_t)c->plan_cache_bytes };
  return AFX_OK;
} catch (...) { return afx::exception_rc(); }
extern "C" int afx_ctx_synchronize(afx_ctx* c) try {
  if (!c) { set_error("null argument"); return AFX_E_BAD_ARGS; }
  CtxLock lock(c);
  AFX_HIP(hipSetDevice(c->device));
  for (auto& L : c->lane)
    if (L.stream) AFX_HIP(hipStreamSynchronize(L.stream));
  return AFX_OK;
} catch (...) { return afx::exception_rc(); }
extern "C" int afx_ctx_set_timing(afx_ctx* c, int enable) try {
  if (!c) { set_error("null argument"); return AFX_E_BAD_ARGS; }
  CtxLock lock(c);
  AFX_HIP(hipSetDevice(c->device));
  int rc = drain_timing(c);
  if (rc) return rc;
  for (int k = 0; k < (int)afx::L_KINDS; k++) { c->kind_ms[k] = 0; c->kind_launches[k] = 0; }
  if (c->d_consts.p) AFX_HIP(hipMemsetAsync(c->clock_probe(), 0, 16 * AFX_CLOCK_SLOTS, c->stream));
  { std::lock_guard<std::mutex> sg(c->settings_mu); c->timing = enable != 0; }
  return AFX_OK;
} catch (...) { return afx::exception_rc(); }
// the clock of every probing block that ran (kernels.hip msm_body), sorted
static int clock_samples(afx_ctx* c, std::vector<double>& out) {
  AFX_HIP(hipSetDevice(c->device));
  for (auto& L : c->lane)
    if (L.stream) AFX_HIP(hipStreamSynchronize(L.stream));
  unsigned long long v[2 * AFX_CLOCK_SLOTS];
  AFX_HIP(hipMemcpy(v, c->clock_probe(), sizeof v, hipMemcpyDeviceToHost));
  out.clear();
  for (int k = 0; k < AFX_CLOCK_SLOTS; k++)
    if (v[2 * k + 1]) out.push_back(100.0 * (double)v[2 * k] / (double)v[2 * k + 1]);
  std::sort(out.begin(), out.end());
  return AFX_OK;
}
extern "C" int afx_ctx_get_core_clock_mhz(afx_ctx* c, double* mhz) try {
  if (!c || !mhz) { set_error("null argument"); return AFX_E_BAD_ARGS; }
  CtxLock lock(c);
  std::vector<double> s;
  const int rc = clock_samples(c, s);
  if (rc) return rc;
  *mhz = s.empty() ? 0.0 : s.size() % 2 ? s[s.size() / 2] : 0.5 * (s[s.size() / 2 - 1] + s[s.size() / 2]);   // the median
  return AFX_OK;
} catch (...) { return afx::exception_rc(); }
extern "C" int afx_ctx_get_core_clock_samples(afx_ctx* c, double* mhz_out, uint32_t cap, uint32_t* n_out) try {
  if (!c || !n_out || (!mhz_out && cap)) { set_error("null argument"); return AFX_E_BAD_ARGS; }
  CtxLock lock(c);
  std::vector<double> s;
  const int rc = clock_samples(c, s);
  if (rc) return rc;
  *n_out = (uint32_t)s.size();
  for (uint32_t k = 0; k < cap && k < s.size(); k++) mhz_out[k] = s[k];
  return AFX_OK;
} catch (...) { return afx::exception_rc(); }
extern "C" int afx_ctx_get_timing(afx_ctx* c, const char* kernel, double* total_ms, uint64_t* launches) try {
  if (!c || !kernel || !total_ms || !launches) { set_error("null argument"); return AFX_E_BAD_ARGS; }
  CtxLock lock(c);
  AFX_HIP(hipSetDevice(c->device));
  int rc = drain_timing(c);
  if (rc) return rc;
  if (strcmp(kernel, "k_msm") == 0) {
    *total_ms = 0; *launches = 0;
    for (int k : { (int)L_MSM_WINDOW, (int)L_MSM_FIXED, (int)L_MSM_NAF, (int)L_MSM_TABLES }) { *total_ms += c->kind_ms[k]; *launches += c->kind_launches[k]; }
    return AFX_OK;
  }
  for (size_t k = 0; k < sizeof KIND_NAMES / sizeof KIND_NAMES[0]; k++)
    if (strcmp(kernel, KIND_NAMES[k]) == 0) { *total_ms = c->kind_ms[k]; *launches = c->kind_launches[k]; return AFX_OK; }
  set_error("unknown kernel name");
  return AFX_E_BAD_ARGS;
} catch (...) { return afx::exception_rc(); }

extern "C" void afx_ctx_destroy(afx_ctx* c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  for (auto& L : c->lane)
    if (L.stream) (void)hipStreamSynchronize(L.stream);
  // Zeroize + Drop of amacs::SecretKey (src/amacs.rs:64-82): wipe every copy of the key and anything derived
  c->d_key.release(true);
  for (auto& L : c->lane) L.ws.release(true);
  for (auto& L : c->lane) {
    L.staging.release(true);
    L.staging_out.release(true);
    if (L.pin) { memset(L.pin, 0, L.pin_cap); (void)hipHostFree(L.pin); L.pin = nullptr; L.pin_cap = 0; }
    if (L.pin_in) { memset(L.pin_in, 0, L.pin_in_cap); (void)hipHostFree(L.pin_in); L.pin_in = nullptr; L.pin_in_cap = 0; }
    if (L.pin_in_done) { (void)hipEventDestroy(L.pin_in_done); L.pin_in_done = nullptr; }
  }
  c->trace_buf.release(false);
  c->d_pos_tables.release(true);
  c->d_sec_tables.release(false);
  c->d_gen_ext.release(true);
  c->d_gen_enc.release(false);
  c->d_consts.release(false);
  for (auto& L : c->lane)
    for (int i = 0; i < 2; i++) {
      L.blob_dev[i].release(true);
      if (L.blob_host[i]) { memset(L.blob_host[i], 0, L.blob_host_cap[i]); (void)hipHostFree(L.blob_host[i]); }
      if (L.blob_event[i]) (void)hipEventDestroy(L.blob_event[i]);
    }
  for (auto& L : c->lane)
    if (L.msm_done) (void)hipEventDestroy(L.msm_done);
  for (auto& t : c->timed) { (void)hipEventDestroy(t.start); (void)hipEventDestroy(t.stop); }
  for (hipEvent_t e : c->event_pool) (void)hipEventDestroy(e);
  for (auto& k : c->host_key) { volatile uint8_t* p = k.data(); for (int i = 0; i < 32; i++) p[i] = 0; }
  // folded transcript states of prover plans derive from the key (and their cache keys hold the witnesses' bytes)
  for (auto& kv : c->folded_states) {
    volatile uint8_t* p = (volatile uint8_t*)kv.second.data();
    for (size_t i = 0; i < sizeof(uint64_t) * 25; i++) p[i] = 0;
    volatile char* q = const_cast<volatile char*>(kv.first.data());
    for (size_t i = 0; i < kv.first.size(); i++) q[i] = 0;
  }
  c->plan_cache.clear();   // ~Plan wipes the blobs (prover plans hold witnesses and blinding schedules)
  for (auto& L : c->lane)
    if (L.stream) (void)hipStreamDestroy(L.stream);
  delete c;
}

int afx_ctx_create_impl(afx_ctx** out, int device, const uint8_t* sp, size_t splen, const uint8_t* key, size_t klen,
                           const uint8_t* key_scalars_only, const uint8_t* issuer_params) {
  if (!out || !sp) { set_error("null argument"); return AFX_E_BAD_ARGS; }
  *out = nullptr;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device < 0 || device >= ndev) {
    set_error("no usable HIP device (this engine has no CPU fallback)");
    return AFX_E_NO_DEVICE;
  }
  if (splen < 4) { set_error("SystemParameters too short"); return AFX_E_BAD_PARAMS; }
  const uint32_t n = rd32(sp);
  if (n == 0 || n > AFX_MAX_ATTRIBUTES || splen != sizeof_system_parameters(n)) { set_error("SystemParameters length / attribute count"); return AFX_E_BAD_PARAMS; }
  std::unique_ptr<afx_ctx, void (*)(afx_ctx*)> c(new afx_ctx(), afx_ctx_destroy);
  { const char* sc = getenv("AFX_PLAN_SELFCHECK"); c->plan_selfcheck = sc && sc[0] == '1'; }   // tests: every plan assembled twice and compared
  c->device = device;
  c->n = n;
  c->g = n < 3 ? 3 : n;
  for (afx::DevBuf* pub : { &c->trace_buf, &c->d_gen_enc, &c->d_consts, &c->d_pos_tables, &c->d_sec_tables, &c->d_gen_ext }) pub->sensitive = false;   // public data
  AFX_HIP(hipSetDevice(device));
  {
    int cus = 0;
    AFX_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device));
    if (cus > 0) c->n_cu = (uint32_t)cus;
  }
  for (auto& L : c->lane) AFX_HIP(hipStreamCreateWithFlags(&L.stream, hipStreamNonBlocking));
  c->stream = c->lane[0].stream;
  const uint32_t np = 9 + c->g + n;
  c->ngen = np + 3;
  c->gen_enc.assign(c->ngen, Enc{});
  for (uint32_t i = 0; i < np; i++) memcpy(c->gen_enc[i].data(), sp + 4 + 32 * (size_t)i, 32);
  if (issuer_params) {   // C_W || I  (src/issuer.rs:155,163)
    memcpy(c->gen_enc[c->id_CW()].data(), issuer_params, 32);
    memcpy(c->gen_enc[c->id_I()].data(), issuer_params + 32, 32);
  }
  const uint8_t* ks = nullptr;
  if (key && klen) {
    if (klen != sizeof_secret_key(n) || rd32(key) != n) { set_error("amacs key length / attribute count"); return AFX_E_BAD_PARAMS; }
    ks = key + 4;
    memcpy(c->gen_enc[c->id_W()].data(), key + 4 + 32 * (size_t)(4 + n), 32);
  } else if (key_scalars_only) {
    ks = key_scalars_only + 4;
  }
  if (ks) {
    c->host_key.assign(4 + n, Enc{});
    for (uint32_t i = 0; i < 4 + n; i++) {
      if (!host_scalar_is_canonical(ks + 32 * (size_t)i)) { set_error("non-canonical key scalar"); return AFX_E_BAD_PARAMS; }   // amacs.rs:141-149
      memcpy(c->host_key[i].data(), ks + 32 * (size_t)i, 32);
    }
    c->has_key = true;
  }
  int rc;
  if ((rc = c->d_gen_enc.ensure(32 * (size_t)c->ngen)) ||
      (rc = c->d_pos_tables.ensure(sizeof(int32_t) * AFX_POS_TABLE_DWORDS * (size_t)c->ngen)) ||
      (rc = c->d_gen_ext.ensure(sizeof(int32_t) * AFX_VAR_DWORDS * (size_t)c->ngen)) || (rc = c->d_key.ensure(32 * (size_t)(4 + n))) ||
      (rc = c->d_consts.ensure(2048)) || (rc = c->lane[0].staging.ensure(32 * (size_t)c->ngen + 4 * (size_t)c->ngen)))
    return rc;
  for (int k = 0; k < afx_ctx::AFX_LANES; k++)
    for (int i = 0; i < 2; i++) {
      afx_ctx::Lane& L = c->lane[k];
      AFX_HIP(hipEventCreateWithFlags(&L.blob_event[i], hipEventDisableTiming));
      if (k >= 2) continue;   // the lanes only collected small calls use get their plan buffers when the first session lands on them (engine.cpp grow_blob)
      if ((rc = L.blob_dev[i].ensure(BLOB_CAP))) return rc;
      AFX_HIP(hipHostMalloc(&L.blob_host[i], BLOB_CAP, hipHostMallocDefault));
      L.blob_host_cap[i] = BLOB_CAP;
    }
  for (auto& L : c->lane) AFX_HIP(hipEventCreateWithFlags(&L.msm_done, hipEventDisableTiming));
  std::vector<uint8_t> flat(32 * (size_t)c->ngen);
  for (uint32_t i = 0; i < c->ngen; i++) memcpy(flat.data() + 32 * (size_t)i, c->gen_enc[i].data(), 32);
  AFX_HIP(hipMemcpyAsync(c->d_gen_enc.p, flat.data(), flat.size(), hipMemcpyHostToDevice, c->stream));
  uint8_t* d_neg = (uint8_t*)c->lane[0].staging.p;
  uint32_t* d_ok = (uint32_t*)((uint8_t*)c->lane[0].staging.p + 32 * (size_t)c->ngen);
  AFX_HIP(afxk_setup_generators(c->stream, (const uint8_t*)c->d_gen_enc.p, c->ngen, (int32_t*)c->d_gen_ext.p, d_neg, d_ok));
  // window bases 2^(AFX_POS_BITS*j) * G go through lane 0's workspace (free at this point), then the tables
  if ((rc = c->lane[0].ws.ensure(sizeof(int32_t) * AFX_VAR_DWORDS * (size_t)c->ngen * AFX_POS_WINDOWS))) return rc;
  AFX_HIP(afxk_setup_postables(c->stream, (const int32_t*)c->d_gen_ext.p, c->ngen, (int32_t*)c->lane[0].ws.p, (int32_t*)c->d_pos_tables.p, 0));
  std::vector<uint8_t> neg(32 * (size_t)c->ngen);
  std::vector<uint32_t> ok(c->ngen);
  AFX_HIP(hipMemcpyAsync(neg.data(), d_neg, neg.size(), hipMemcpyDeviceToHost, c->stream));
  AFX_HIP(hipMemcpyAsync(ok.data(), d_ok, 4 * (size_t)c->ngen, hipMemcpyDeviceToHost, c->stream));
  if (c->has_key) {
    std::vector<uint8_t> kflat(32 * (size_t)(4 + n));
    for (uint32_t i = 0; i < 4 + n; i++) memcpy(kflat.data() + 32 * (size_t)i, c->host_key[i].data(), 32);
    AFX_HIP(hipMemcpyAsync(c->d_key.p, kflat.data(), kflat.size(), hipMemcpyHostToDevice, c->stream));
    AFX_HIP(hipStreamSynchronize(c->stream));
    volatile uint8_t* p = kflat.data();
    for (size_t i = 0; i < kflat.size(); i++) p[i] = 0;
  }
  uint8_t consts[2048];
  memset(consts, 0, sizeof consts);
  consts[0] = 1;   // Scalar::one()
  AFX_HIP(hipMemcpyAsync(c->d_consts.p, consts, sizeof consts, hipMemcpyHostToDevice, c->stream));
  AFX_HIP(hipStreamSynchronize(c->stream));
  for (uint32_t i = 0; i < c->ngen; i++)
    if (!ok[i]) { set_error("generator / key point " + std::to_string(i) + " does not decompress"); return AFX_E_BAD_PARAMS; }   // parameters.rs:77-89
  c->gen_neg_enc.assign(c->ngen, Enc{});
  for (uint32_t i = 0; i < c->ngen; i++) memcpy(c->gen_neg_enc[i].data(), neg.data() + 32 * (size_t)i, 32);
  // the default mode keeps secrets out of table addresses on the prover-side calls: their generator tables (63 KB each) are part of every context
  if ((rc = build_secret_tables(c.get()))) return rc;
  *out = c.release();
  return AFX_OK;
}

extern "C" int afx_ctx_create(afx_ctx** out, int device, const uint8_t* sysparams, size_t sysparams_len, const uint8_t* amacs_key,
                              size_t amacs_key_len, const uint8_t issuer_params[64]) try {
  if (!issuer_params) { set_error("issuer_params is required"); return AFX_E_BAD_ARGS; }
  return afx_ctx_create_impl(out, device, sysparams, sysparams_len, amacs_key, amacs_key_len, nullptr, issuer_params);
} catch (...) { return afx::exception_rc(); }

// ProofOfEncryption::verify, src/nizk/encryption.rs:154-210
static void add_encproof_verify(Assembler& as, JobSets& js, uint16_t index, const afx_encproof_soa& e, size_t total, size_t off, uint32_t trace_row) {
  afx_ctx* c = as.ctx;
  auto row = [&](const uint8_t* base, size_t k) { return base + (k * total + off) * 32; };
  if (index >= c->n) { as.fail_all = true; return; }   // G_m[self.index] panics, encryption.rs:179
  js.sccheck.push_back({ row(e.challenge, 0) });
  for (int k = 0; k < 6; k++) js.sccheck.push_back({ row(e.responses, k) });
  int32_t *v_pk = as.new_var(), *v_E1 = as.new_var(), *v_E2 = as.new_var(), *v_Cy1 = as.new_var(), *v_Cy2 = as.new_var(),
          *v_Cy3 = as.new_var(), *v_Cy2p = as.new_var(), *v_D1 = as.new_var();
  uint8_t *e_D1 = as.new_enc(), *e_D2 = as.new_enc();
  js.decode.push_back({ row(e.pk, 0), v_pk, 1 });
  js.decode.push_back({ row(e.E1, 0), v_E1, 1 });
  js.decode.push_back({ row(e.E2, 0), v_E2, 0 });
  js.decode.push_back({ row(e.C_y_1, 0), v_Cy1, 0 });
  js.decode.push_back({ row(e.C_y_2, 0), v_Cy2, 1 });
  js.decode.push_back({ row(e.C_y_3, 0), v_Cy3, 1 });
  js.decode.push_back({ row(e.C_y_2p, 0), v_Cy2p, 1 });
  afx_pointop_job d1 = { v_Cy1, v_E2, nullptr, +1, -1, v_D1, e_D1, 1 };    // C_y_1 - E2   (:183)
  js.pointop.push_back(d1);
  // -E1 (:185): only its encoding is needed (the term a*(-E1) runs as -(a*E1) on E1's table, below), and the encoding of the
  // negation of a DECODED point needs no square root: one inversion per item for all its proofs of encryption (k_negenc)
  // (a small pass has idle lanes and waits for its longest stage: there every -E1 keeps a lane and a square root of its own,
  // in the launch that computes C_y_1 - E2 anyway, instead of a launch that walks the item's proofs one after the other)
  if (as.small()) js.pointop.push_back({ v_E1, nullptr, nullptr, -1, 0, nullptr, e_D2, 1 });
  else js.negenc.push_back({ row(e.E1, 0), v_E1, e_D2, 1, 0 });
  auto resp = [&](int k) { ScalarVar s; s.dev = row(e.responses, k); s.stride = 32; return s; };
  SchnorrBuilder v(as, "2019/1416 anonymous credentials", "2019/1416 proof of encryption");
  const int a = v.allocate_scalar("a", resp(0));
  const int a0 = v.allocate_scalar("a0", resp(1));
  const int a1 = v.allocate_scalar("a1", resp(2));
  const int m3 = v.allocate_scalar("m3", resp(3));
  const int z = v.allocate_scalar("z", resp(4));
  const int z1 = v.allocate_scalar("z1", resp(5));
  const int pk = v.allocate_point("pk", PointVar::Var(v_pk, row(e.pk, 0)));
  const int G_a = v.allocate_point("G_a", PointVar::Const(c->id_Ga()));
  const int G_a_0 = v.allocate_point("G_a_0", PointVar::Const(c->id_Ga0()));
  const int G_a_1 = v.allocate_point("G_a_1", PointVar::Const(c->id_Ga1()));
  const int G_y_1 = v.allocate_point("G_y_1", PointVar::Const(c->id_Gy(0)));
  const int G_y_2 = v.allocate_point("G_y_2", PointVar::Const(c->id_Gy(1)));
  const int G_y_3 = v.allocate_point("G_y_3", PointVar::Const(c->id_Gy(2)));
  const int G_m_3 = v.allocate_point("G_m_3", PointVar::Const(c->id_Gm(index)));
  const int C_y_2 = v.allocate_point("C_y_2", PointVar::Var(v_Cy2, row(e.C_y_2, 0)));
  const int C_y_3 = v.allocate_point("C_y_3", PointVar::Var(v_Cy3, row(e.C_y_3, 0)));
  const int C_y_2p = v.allocate_point("C_y_2'", PointVar::Var(v_Cy2p, row(e.C_y_2p, 0)));
  const int C_y_1_minus_E2 = v.allocate_point("C_y_1-E2", PointVar::Var(v_D1, e_D1));
  const int E1 = v.allocate_point("E1", PointVar::Var(v_E1, row(e.E1, 0)));
  const int minus_E1 = v.allocate_point("-E1", PointVar::NegOf(v_E1, e_D2));   // a*(-E1) runs as -(a*E1) on E1's window table
  v.constrain(pk, { { a, G_a }, { a0, G_a_0 }, { a1, G_a_1 } });
  v.constrain(C_y_1_minus_E2, { { z, G_y_1 }, { a, minus_E1 } });
  v.constrain(C_y_2p, { { a1, C_y_2 } });
  v.constrain(E1, { { a0, C_y_2 }, { m3, C_y_2p }, { z1, G_y_2 } });
  v.constrain(C_y_3, { { z, G_y_3 }, { m3, G_m_3 } });
  v.verify_compact(row(e.challenge, 0), trace_row, total, off, js.msm1, js.hash);
}


// Shapes on which the reference indexes out of range (panics) or zkp rejects every proof: every item fails, and no
// per-item array is touched.  Also yields the compact C_y list and, per compact index, which hidden-scalar slot
// constraint #3 looks up (presentation.rs:427-433 uses the compact index as an original position).
static bool presentation_shape_rejects(const afx_ctx* c, const afx_shape& sh, uint32_t keep[AFX_MAX_ATTRIBUTES], uint32_t* k_out,
                                       int hidden_slot[AFX_MAX_ATTRIBUTES], uint32_t pos[AFX_MAX_ATTRIBUTES]) {
  const uint32_t n = sh.n_attributes, hs = sh.n_hidden_scalars;
  if (n > c->n || n > AFX_MAX_ATTRIBUTES || hs > AFX_MAX_ATTRIBUTES || sh.n_enc_proofs > AFX_MAX_ATTRIBUTES ||
      sh.n_responses != 3 + hs /* verify_compact: responses.len() != num_scalars */)
    return true;
  for (uint32_t i = 0; i < n; i++)
    if (sh.kinds[i] > AFX_ENC_SECRET_POINT) return true;
  for (uint32_t j = 0; j < hs; j++)
    if (sh.hidden_scalar_indices[j] >= c->n) return true;   // G_m[*i], presentation.rs:407
  for (uint32_t e = 0; e < sh.n_enc_proofs; e++)
    if (sh.enc_indices[e] >= c->n) return true;             // G_m[self.index], encryption.rs:179
  uint32_t k = 0;
  for (uint32_t i = 0; i < n; i++)
    if (sh.kinds[i] != AFX_ENC_SECRET_POINT) keep[k++] = i;
  if (c->strict) {
    // one proof of encryption per hidden group element, in position order
    uint32_t e = 0;
    for (uint32_t i = 0; i < n; i++)
      if (sh.kinds[i] == AFX_ENC_SECRET_POINT) {
        if (e >= sh.n_enc_proofs || sh.enc_indices[e] != i) return true;
        e++;
      }
    if (e != sh.n_enc_proofs) return true;
  }
  // pos[j]: the attribute position whose kind and generators constraint #3 uses for the j-th kept commitment: j itself
  // in the reference (presentation.rs:427-433, the compact index used as an original position), keep[j] in strict mode
  for (uint32_t j = 0; j < k; j++) {
    const uint32_t p = c->strict ? keep[j] : j;
    pos[j] = p;
    hidden_slot[j] = -1;
    if (sh.kinds[p] == AFX_ENC_SECRET_POINT) continue;
    if (p >= c->g) return true;
    if (sh.kinds[p] == AFX_ENC_SECRET_SCALAR) {
      for (uint32_t h = 0; h < hs; h++)
        if (sh.hidden_scalar_indices[h] == p) { hidden_slot[j] = (int)h; break; }
      if (hidden_slot[j] < 0) return true;   // H_s[i] / G_m[i] lookup panics (:81, :100)
    }
  }
  *k_out = k;
  return false;
}

// Issuer::verify -> ProofOfValidCredential::verify, src/nizk/presentation.rs:324-443
static void build_presentation_verify(Assembler& as, const afx_shape& sh, const afx_presentation_soa& b, size_t total, size_t off,
                                      uint8_t* status_dev) {
  afx_ctx* c = as.ctx;
  JobSets js;
  auto row = [&](const uint8_t* base, size_t k) { return base + (k * total + off) * 32; };
  const uint32_t n = sh.n_attributes, hs = sh.n_hidden_scalars;
  uint32_t keep[AFX_MAX_ATTRIBUTES], pos[AFX_MAX_ATTRIBUTES], k = 0;
  int hidden_slot[AFX_MAX_ATTRIBUTES];
  if (presentation_shape_rejects(c, sh, keep, &k, hidden_slot, pos)) as.fail_all = true;
  if (as.fail_all) { emit(as, js, status_dev, AFX_ST_VERIFICATION_FAILURE); return; }

  js.sccheck.push_back({ row(b.challenge, 0) });
  for (uint32_t r = 0; r < sh.n_responses; r++) js.sccheck.push_back({ row(b.responses, r) });
  int32_t *v_Cx0 = as.new_var(), *v_Cx1 = as.new_var(), *v_CV = as.new_var(), *v_A = as.new_var(), *v_Z = as.new_var();
  uint8_t* e_Z = as.new_enc();
  js.decode.push_back({ row(b.C_x_0, 0), v_Cx0, 1 });
  js.decode.push_back({ row(b.C_x_1, 0), v_Cx1, 1 });
  js.decode.push_back({ row(b.C_V, 0), v_CV, 0 });
  int32_t* v_Cy[AFX_MAX_ATTRIBUTES];
  // Z = C_V - W - x0*C_x0 - x1*C_x1 - sum y_i * X_i                                  (:342-352)
  std::vector<afx_msm_term> zterms;
  zterms.push_back(mk_term(c->key_x0(), 0, v_Cx0, -1, true));
  zterms.push_back(mk_term(c->key_x1(), 0, v_Cx1, -1, true));
  for (uint32_t i = 0; i < n; i++) {
    v_Cy[i] = as.new_var();
    js.decode.push_back({ row(b.C_y, i), v_Cy[i], sh.kinds[i] != AFX_ENC_SECRET_POINT ? 1u : 0u });
    const int32_t* X = v_Cy[i];
    if (sh.kinds[i] == AFX_ENC_PUBLIC_SCALAR) {
      // y_i*(C_y_i + m_i*G_m_i) = y_i*C_y_i + (y_i*m_i)*G_m_i : the second part is a fixed-base term
      js.sccheck.push_back({ row(b.attr_values, i) });
      uint8_t* ym = as.new_enc();
      afx_scalarop_job so;
      memset(&so, 0, sizeof so);
      so.a = c->key_y(i); so.a_stride = 0; so.b = row(b.attr_values, i); so.b_stride = 32; so.out = ym;
      js.scalarop.push_back(so);
      zterms.push_back(mk_term(ym, 32, nullptr, (int32_t)c->id_Gm(i), true));
    } else if (sh.kinds[i] == AFX_ENC_PUBLIC_POINT) {
      int32_t *v_M = as.new_var(), *v_X = as.new_var();
      js.decode.push_back({ row(b.attr_values, i), v_M, 0 });
      afx_pointop_job po = { v_Cy[i], v_M, nullptr, +1, +1, v_X, nullptr, 0 };
      js.pointop.push_back(po);
      X = v_X;
    }
    zterms.push_back(mk_term(c->key_y(i), 0, X, -1, true));
  }
  afx_pointop_job pa = { v_CV, nullptr, c->gen_ext(c->id_W()), +1, -1, v_A, nullptr, 0 };   // C_V - W
  js.pointop.push_back(pa);
  afx_msm_job zj;
  memset(&zj, 0, sizeof zj);
  set_terms(zj, zterms);
  zj.addend = v_A;
  zj.out_var = v_Z;
  zj.out_enc = e_Z;
  zj.reject_identity = 1;
  const size_t z_index = js.msm1.size();
  js.msm1.push_back(zj);

  auto resp = [&](uint32_t r) { ScalarVar s; s.dev = row(b.responses, r); s.stride = 32; return s; };
  SchnorrBuilder v(as, "2019/1416 anonymous credential", "2019/1416 presentation proof");
  const int z = v.allocate_scalar("z", resp(0));
  const int z_0 = v.allocate_scalar("z_0", resp(1));
  const int t = v.allocate_scalar("t", resp(2));
  int H_s[AFX_MAX_ATTRIBUTES];
  for (uint32_t j = 0; j < hs; j++) H_s[j] = v.allocate_scalar("m", resp(3 + j));
  const int I = v.allocate_point("I", PointVar::Const(c->id_I()));
  const int C_x_1 = v.allocate_point("C_x_1", PointVar::Var(v_Cx1, row(b.C_x_1, 0)));
  const int C_x_0 = v.allocate_point("C_x_0", PointVar::Var(v_Cx0, row(b.C_x_0, 0)));
  const int G_x_0 = v.allocate_point("G_x_0", PointVar::Const(c->id_Gx0()));
  const int G_x_1 = v.allocate_point("G_x_1", PointVar::Const(c->id_Gx1()));
  int C_y[AFX_MAX_ATTRIBUTES], G_y[AFX_MAX_ATTRIBUTES], G_m[AFX_MAX_ATTRIBUTES];
  for (uint32_t j = 0; j < k; j++) C_y[j] = v.allocate_point("C_y", PointVar::Var(v_Cy[keep[j]], row(b.C_y, keep[j])));
  for (uint32_t i = 0; i < c->g; i++) G_y[i] = v.allocate_point("G_y", PointVar::Const(c->id_Gy(i)));
  for (uint32_t j = 0; j < hs; j++) G_m[j] = v.allocate_point("G_m", PointVar::Const(c->id_Gm(sh.hidden_scalar_indices[j])));
  // strict mode: C_y[i] - C_y_1 = z*G_y[i] + z*(-G_y[0]) for every hidden group element i and the C_y_1 of its proof of encryption
  // (the DLEQ of README.md:121-122; the oracle states the reason).  At position 0 the difference must be the identity.
  int D[AFX_MAX_ATTRIBUTES], D_pos[AFX_MAX_ATTRIBUTES], nD = 0, neg_G_y_1 = -1;
  if (c->strict)
    for (uint32_t e = 0; e < sh.n_enc_proofs; e++) {   // presentation_shape_rejects: one proof per hidden group element, in position order
      const uint32_t i = sh.enc_indices[e];
      int32_t *v_C1 = as.new_var(), *v_D = as.new_var();
      uint8_t* e_D = as.new_enc();
      js.decode.push_back({ row(b.enc[e].C_y_1, 0), v_C1, 0 });
      afx_pointop_job dj = { v_Cy[i], v_C1, nullptr, +1, -1, v_D, e_D, i == 0 ? 2u /* must BE the identity */ : 1u };
      js.pointop.push_back(dj);
      if (i == 0) continue;
      if (neg_G_y_1 < 0) neg_G_y_1 = v.allocate_point("-G_y_1", PointVar::Const(c->id_Gy(0), true));
      D[nD] = v.allocate_point("C_y-C_y_1", PointVar::Var(v_D, e_D));
      D_pos[nD++] = (int)i;
    }
  // Small passes: constraint #1 recomputes R = z*I - c*Z, and a chain on Z itself could only start when Z's chains have ended -
  // two chain latencies in a row, most of what a small call waits for.  Z is a sum the verifier knows term by term, so there
  // -c*Z runs as c*x0*C_x_0 + c*x1*C_x_1 + sum c*y_i*X_i - c*(C_V - W): chains on the decoded inputs, beside Z's own.
  PointVar pZ = PointVar::Var(v_Z, e_Z);
  // (not with secret-independent addressing: the products c*x0 ... are the key times a public factor, and the expanded terms would
  // have to scan their tables like Z's own do - the second chain is the cheaper price there)
  const bool expand_Z = as.small() && !c->secure_plan(false);
  if (expand_Z) {
    pZ.parts.push_back({ nullptr, 0, false, v_A, -1 });
    for (const afx_msm_term& t : zterms) pZ.parts.push_back({ t.scalar, t.scalar_stride, true, t.var, t.fixed_idx });
  }
  const int Z = v.allocate_point("Z", pZ);
  v.constrain(Z, { { z, I } });
  v.constrain(C_x_1, { { t, C_x_0 }, { z_0, G_x_0 }, { z, G_x_1 } });
  for (uint32_t j = 0; j < k; j++) {
    const uint32_t p = pos[j];
    if (sh.kinds[p] == AFX_ENC_SECRET_POINT) continue;
    if (sh.kinds[p] == AFX_ENC_SECRET_SCALAR) v.constrain(C_y[j], { { z, G_y[p] }, { H_s[hidden_slot[j]], G_m[hidden_slot[j]] } });
    else v.constrain(C_y[j], { { z, G_y[p] } });
  }
  for (int d = 0; d < nD; d++) v.constrain(D[d], { { z, G_y[D_pos[d]] }, { z, neg_G_y_1 } });
  // one launch for everything: the lane that finishes Z goes straight on to constraint #1 (Z = z*I), the only job that needs it
  const size_t first_constraint = js.msm1.size();
  v.verify_compact(row(b.challenge, 0), 0, total, off, js.msm1, js.hash, nullptr, expand_Z ? &js.scalarop2 : nullptr);
  if (!expand_Z) js.msm1[z_index].chain_to = (int32_t)first_constraint;
  // proofs of encryption: verified independently, whatever their number (:438-440)
  for (uint32_t e = 0; e < sh.n_enc_proofs && !as.fail_all; e++) add_encproof_verify(as, js, sh.enc_indices[e], b.enc[e], total, off, 1 + e);
  emit(as, js, status_dev, AFX_ST_VERIFICATION_FAILURE);
}

extern "C" int afx_verify_presentations_dev(afx_ctx* ctx, const afx_shape* shape, const afx_presentation_soa* batch, size_t count,
                                            uint8_t* status_dev) try {
  CtxLock lock__(ctx);
  if (!ctx || !shape || !batch || !status_dev) { set_error("null argument"); return AFX_E_BAD_ARGS; }
  if (!ctx->has_key) { set_error("Issuer::verify needs the issuer key"); return AFX_E_NO_KEY; }
  if (count == 0) return AFX_OK;
  if (shape->n_enc_proofs && shape->n_enc_proofs <= AFX_MAX_ATTRIBUTES && !batch->enc) { set_error("enc proofs missing"); return AFX_E_BAD_ARGS; }
  const afx_shape sh = *shape;
  const afx_presentation_soa b = *batch;
  {
    // the arrays a well-formed request reads (a shape every item fails on reads none): a null one is a bad call, not a GPU fault
    uint32_t keep[AFX_MAX_ATTRIBUTES], pos[AFX_MAX_ATTRIBUTES], k = 0;
    int slot[AFX_MAX_ATTRIBUTES];
    if (!presentation_shape_rejects(ctx, sh, keep, &k, slot, pos)) {
      bool missing = !b.challenge || !b.C_x_0 || !b.C_x_1 || !b.C_V || (sh.n_attributes && !b.C_y) || (sh.n_responses && !b.responses);
      for (uint32_t i = 0; i < sh.n_attributes; i++)
        if ((sh.kinds[i] == AFX_ENC_PUBLIC_SCALAR || sh.kinds[i] == AFX_ENC_PUBLIC_POINT) && !b.attr_values) missing = true;
      for (uint32_t e = 0; e < sh.n_enc_proofs; e++) {
        const afx_encproof_soa& q = b.enc[e];
        missing |= !q.challenge || !q.responses || !q.pk || !q.E1 || !q.E2 || !q.C_y_1 || !q.C_y_2 || !q.C_y_3 || !q.C_y_2p;
      }
      if (missing) { set_error("null batch array"); return AFX_E_BAD_ARGS; }
    }
  }
  std::vector<afx_encproof_soa> encs;
  if (b.enc && sh.n_enc_proofs <= AFX_MAX_ATTRIBUTES) encs.assign(b.enc, b.enc + sh.n_enc_proofs);
  // the plan's size depends on the shape and the mode, not on the arrays: remember it (the trace changes nothing in size,
  // but a too-small trace buffer is a plan error that the sizing run reports, so traced calls are not cached)
  const afx_shape csh = canonical_shape(sh);
  const PlanKey key = ctx->trace ? PlanKey() : plan_key("verify_presentations", &csh, sizeof csh, mode_flags(ctx));
  return run_chunked(ctx, count, [&](Assembler& as, size_t off, uint32_t) {
    afx_presentation_soa bb = b;
    bb.enc = encs.data();
    build_presentation_verify(as, sh, bb, count, off, status_dev + off);
  }, key);
} catch (...) { return afx::exception_rc(); }

extern "C" int afx_verify_encryption_proofs_dev(afx_ctx* ctx, uint16_t index, const afx_encproof_soa* batch, size_t count, uint8_t* status_dev) try {
  CtxLock lock__(ctx);
  if (!ctx || !batch || !status_dev) { set_error("null argument"); return AFX_E_BAD_ARGS; }
  if (count == 0) return AFX_OK;
  const afx_encproof_soa e = *batch;
  if (index < ctx->n && (!e.challenge || !e.responses || !e.pk || !e.E1 || !e.E2 || !e.C_y_1 || !e.C_y_2 || !e.C_y_3 || !e.C_y_2p)) {
    set_error("null batch array");
    return AFX_E_BAD_ARGS;
  }
  return run_chunked(ctx, count, [&](Assembler& as, size_t off, uint32_t) {
    JobSets js;
    add_encproof_verify(as, js, index, e, count, off, 0);
    emit(as, js, status_dev + off, AFX_ST_VERIFICATION_FAILURE);
  });
} catch (...) { return afx::exception_rc(); }

// ------------------------------------------------------------------------------------------------
// host-pointer front ends: stage the SoA batch into HBM, run the *_dev form, fetch the status bytes
// ------------------------------------------------------------------------------------------------

static void stage_encproof(Stager& st, const afx_encproof_soa& e, size_t total, size_t first, size_t n, size_t dn, size_t offs[9]) {
  const uint8_t* f[9] = { e.challenge, e.responses, e.pk, e.E1, e.E2, e.C_y_1, e.C_y_2, e.C_y_3, e.C_y_2p };
  for (int i = 0; i < 9; i++) offs[i] = st.add_rows(f[i], i == 1 ? 6 : 1, 32, total, first, n, dn);
}
static afx_encproof_soa dev_encproof(const Stager& st, const size_t offs[9]) {
  afx_encproof_soa d = { st.dev(offs[0]), st.dev(offs[1]), st.dev(offs[2]), st.dev(offs[3]), st.dev(offs[4]),
                         st.dev(offs[5]), st.dev(offs[6]), st.dev(offs[7]), st.dev(offs[8]) };
  return d;
}

// Items [first, first + n) of a batch of `total` presentations held in host memory; `status` is the batch's status array
// (element first + i answers item first + i), so that callers sharding one batch over several contexts share every array.
// The range is cut into slices that alternate between the context's two lanes: while one slice is verified, the next is
// copied to HBM (">= 2 chunks per device to overlap H2D with compute").
static int verify_presentations_host(afx_ctx* ctx, const afx_shape* shape, const afx_presentation_soa* b, size_t total, size_t first, size_t n,
                                     uint8_t* status) {
  if (n == 0) return AFX_OK;
  AFX_HIP(hipSetDevice(ctx->device));
  if (!ctx->has_key) { set_error("Issuer::verify needs the issuer key"); return AFX_E_NO_KEY; }
  {
    // a shape every item fails on says nothing reliable about the arrays' extents: answer without reading them
    uint32_t keep[AFX_MAX_ATTRIBUTES], pos[AFX_MAX_ATTRIBUTES], k = 0;
    int slot[AFX_MAX_ATTRIBUTES];
    if (presentation_shape_rejects(ctx, *shape, keep, &k, slot, pos)) { memset(status + first, AFX_ST_VERIFICATION_FAILURE, n); return AFX_OK; }
  }
  const uint32_t na = shape->n_attributes, nr = shape->n_responses, ne = shape->n_enc_proofs;
  if (!b->challenge || !b->C_x_0 || !b->C_x_1 || !b->C_V || (na && !b->C_y) || (nr && !b->responses) || (ne && !b->enc)) { set_error("null batch array"); return AFX_E_BAD_ARGS; }
  // what makes two calls one pass (statements.hpp host_pipe): the statement, the shape's used fields, the mode, the optional array
  const afx_shape jsh = canonical_shape(*shape);
  const PlanKey jkey = plan_key("V", &jsh, sizeof jsh, mode_flags(ctx) | (b->attr_values ? (uint64_t)1 << 63 : 0));
  return host_pipe(ctx, n, [&](Stager& st, size_t off, size_t sn) -> int {
    const size_t f0 = first + off;
    const size_t dn = st.dev_items(sn);   // the pass's size on the device: small calls are padded to the size their plan is kept for
    const size_t o_ch = st.add_rows(b->challenge, 1, 32, total, f0, sn, dn), o_rs = st.add_rows(b->responses, nr, 32, total, f0, sn, dn),
                 o_x0 = st.add_rows(b->C_x_0, 1, 32, total, f0, sn, dn), o_x1 = st.add_rows(b->C_x_1, 1, 32, total, f0, sn, dn),
                 o_cv = st.add_rows(b->C_V, 1, 32, total, f0, sn, dn), o_cy = st.add_rows(b->C_y, na, 32, total, f0, sn, dn),
                 o_av = b->attr_values ? st.add_rows(b->attr_values, na, 32, total, f0, sn, dn) : st.reserve(0);
    std::vector<std::array<size_t, 9>> eo(ne);
    for (uint32_t e = 0; e < ne; e++) stage_encproof(st, b->enc[e], total, f0, sn, dn, eo[e].data());
    const size_t o_st = st.add(nullptr, dn);
    st.plan_fetch(status, o_st, 1, 1, total, f0, sn, dn);
    int rc = st.upload();
    if (rc) return rc;
    std::vector<afx_encproof_soa> de(ne);
    for (uint32_t e = 0; e < ne; e++) de[e] = dev_encproof(st, eo[e].data());
    afx_presentation_soa d = { st.dev(o_ch), st.dev(o_rs), st.dev(o_x0), st.dev(o_x1), st.dev(o_cv), st.dev(o_cy), b->attr_values ? st.dev(o_av) : nullptr, de.data() };
    if ((rc = afx_verify_presentations_dev(ctx, shape, &d, dn, st.dev(o_st)))) return rc;
    return st.fetch_all();
  }, jkey);
}

extern "C" int afx_verify_presentations_range(afx_ctx* ctx, const afx_shape* shape, const afx_presentation_soa* b, size_t total, size_t first,
                                              size_t n, uint8_t* status) try {
  CtxLock lock__(ctx, true);
  if (!ctx || !shape || !b || !status) { set_error("null argument"); return AFX_E_BAD_ARGS; }
  if (first > total || n > total - first) { set_error("range outside the batch"); return AFX_E_BAD_ARGS; }
  return verify_presentations_host(ctx, shape, b, total, first, n, status);
} catch (...) { return afx::exception_rc(); }
extern "C" int afx_verify_presentations(afx_ctx* ctx, const afx_shape* shape, const afx_presentation_soa* b, size_t count, uint8_t* status) try {
  return afx_verify_presentations_range(ctx, shape, b, count, 0, count, status);
} catch (...) { return afx::exception_rc(); }

extern "C" int afx_verify_encryption_proofs(afx_ctx* ctx, uint16_t index, const afx_encproof_soa* b, size_t count, uint8_t* status) try {
  CtxLock lock__(ctx);
  if (!ctx || !b || !status) { set_error("null argument"); return AFX_E_BAD_ARGS; }
  if (count == 0) return AFX_OK;
  AFX_HIP(hipSetDevice(ctx->device));
  Stager st(ctx);
  size_t eo[9];
  stage_encproof(st, *b, count, 0, count, count, eo);
  const size_t o_st = st.add(nullptr, count);
  int rc = st.upload();
  if (rc) return rc;
  afx_encproof_soa d = dev_encproof(st, eo);
  if ((rc = afx_verify_encryption_proofs_dev(ctx, index, &d, count, st.dev(o_st)))) return rc;
  AFX_HIP(hipMemcpyAsync(status, st.dev(o_st), count, hipMemcpyDeviceToHost, ctx->stream));
  AFX_HIP(hipStreamSynchronize(ctx->stream));
  return AFX_OK;
} catch (...) { return afx::exception_rc(); }

// ------------------------------------------------------------------------------------------------
// batch primitives (K* rows): from_uniform_bytes, from_bytes_mod_order_wide, decompress/compress, MSM
// ------------------------------------------------------------------------------------------------
extern "C" int afx_points_from_uniform_bytes(afx_ctx* ctx, const uint8_t* wide, size_t count, uint8_t* out) try {
  CtxLock lock__(ctx);
  if (!ctx || !wide || !out) { set_error("null argument"); return AFX_E_BAD_ARGS; }
  if (count == 0) return AFX_OK;
  AFX_HIP(hipSetDevice(ctx->device));
  Stager st(ctx);
  const size_t o_in = st.add(wide, 64 * count), o_out = st.add(nullptr, 32 * count);
  int rc = st.upload();
  if (rc) return rc;
  AFX_HIP(afxk_from_uniform(ctx->stream, st.dev(o_in), st.dev(o_out), nullptr, (uint32_t)count));
  AFX_HIP(hipMemcpyAsync(out, st.dev(o_out), 32 * count, hipMemcpyDeviceToHost, ctx->stream));
  AFX_HIP(hipStreamSynchronize(ctx->stream));
  return AFX_OK;
} catch (...) { return afx::exception_rc(); }
extern "C" int afx_scalars_from_wide_bytes(afx_ctx* ctx, const uint8_t* wide, size_t count, uint8_t* out) try {
  CtxLock lock__(ctx);
  if (!ctx || !wide || !out) { set_error("null argument"); return AFX_E_BAD_ARGS; }
  if (count == 0) return AFX_OK;
  AFX_HIP(hipSetDevice(ctx->device));
  Stager st(ctx);
  const size_t o_in = st.add(wide, 64 * count), o_out = st.add(nullptr, 32 * count);
  int rc = st.upload();
  if (rc) return rc;
  AFX_HIP(afxk_reduce_wide(ctx->stream, st.dev(o_in), st.dev(o_out), (uint32_t)count));
  AFX_HIP(hipMemcpyAsync(out, st.dev(o_out), 32 * count, hipMemcpyDeviceToHost, ctx->stream));
  AFX_HIP(hipStreamSynchronize(ctx->stream));
  return AFX_OK;
} catch (...) { return afx::exception_rc(); }
// A merlin transcript over a batch, from a script (include/aeonflux_gpu.h): compiled by StrobeSim like the statements' own transcripts,
// run by k_hash.  What it is for: merlin's published conformance vectors - and any transcript a caller wants to cross-check - on the
// GPU's STROBE / Keccak path itself rather than through a statement.
extern "C" int afx_merlin_challenges(afx_ctx* ctx, const uint8_t* script, size_t script_len, const uint8_t* const* fields, uint32_t n_fields, size_t count,
                                     uint8_t* out64) try {
  CtxLock lock__(ctx);
  if (!ctx || !script || !out64 || (n_fields && !fields)) { set_error("null argument"); return AFX_E_BAD_ARGS; }
  if (n_fields > 4096) { set_error("too many fields"); return AFX_E_BAD_ARGS; }
  for (uint32_t k = 0; k < n_fields; k++) if (!fields[k]) { set_error("null field array"); return AFX_E_BAD_ARGS; }
  if (count == 0) return AFX_OK;
  // ---- the script: op, then length-prefixed (u32 LE) byte strings
  size_t at = 0;
  auto u32 = [&](uint32_t& v) { if (script_len - at < 4) return false; v = rd32(script + at); at += 4; return true; };
  auto bytes = [&](const uint8_t*& p, uint32_t& n) { if (!u32(n) || script_len - at < n) return false; p = script + at; at += n; return true; };
  const uint8_t* lab = nullptr;
  uint32_t llen = 0;
  if (script_len < 1 || script[at++] != AFX_MERLIN_NEW || !bytes(lab, llen)) { set_error("transcript script: must open with AFX_MERLIN_NEW"); return AFX_E_BAD_ARGS; }
  StrobeSim sim(lab, llen);
  // (a challenge that is not the script's last operation is replayed for its effect on the state; only the last one's bytes come back)
  bool closed = false;
  const uint8_t* pend_lab = nullptr;
  uint32_t pend_llen = 0, pend_n = 0;
  while (at < script_len) {
    if (closed) { sim.challenge_discard(pend_lab, pend_llen, pend_n); closed = false; }
    const uint8_t op = script[at++];
    if (!bytes(lab, llen)) { set_error("transcript script: truncated label"); return AFX_E_BAD_ARGS; }
    if (op == AFX_MERLIN_APPEND) {
      const uint8_t* msg = nullptr;
      uint32_t mlen = 0;
      if (!bytes(msg, mlen)) { set_error("transcript script: truncated message"); return AFX_E_BAD_ARGS; }
      sim.append_message_const(lab, llen, msg, mlen);
    } else if (op == AFX_MERLIN_APPEND_FIELD) {
      uint32_t f = 0;
      if (!u32(f) || f >= n_fields) { set_error("transcript script: field index out of range"); return AFX_E_BAD_ARGS; }
      sim.append_message_hole32(lab, llen, (int)f);
    } else if (op == AFX_MERLIN_CHALLENGE) {
      uint32_t n = 0;
      if (!u32(n) || n == 0 || n > 64) { set_error("transcript script: a challenge is 1 .. 64 bytes"); return AFX_E_BAD_ARGS; }
      pend_lab = lab; pend_llen = llen; pend_n = n;
      closed = true;
    } else { set_error("transcript script: unknown operation"); return AFX_E_BAD_ARGS; }
  }
  if (!closed) { set_error("transcript script: the last operation must be a challenge"); return AFX_E_BAD_ARGS; }
  sim.challenge_final(pend_lab, pend_llen, pend_n, AFX_SQ_WIDE_OUT, 0);
  AFX_HIP(hipSetDevice(ctx->device));
  Stager st(ctx);
  std::vector<size_t> o_f(n_fields);
  for (uint32_t k = 0; k < n_fields; k++) o_f[k] = st.add(fields[k], 32 * count);
  const size_t o_out = st.add(nullptr, 64 * count), o_st = st.add(nullptr, count);
  int rc = st.upload();
  if (rc) return rc;
  rc = run_chunked(ctx, count, [&](Assembler& as, size_t off, uint32_t) {
    std::vector<const uint8_t*> f(n_fields);
    for (uint32_t k = 0; k < n_fields; k++) f[k] = st.dev(o_f[k]) + off * 32;
    afx_hash_program p = make_hash_program(as, sim, f);
    uint8_t* outs[1] = { st.dev(o_out) + off * 64 };
    p.outs = as.put_ptrs(outs, 1);
    p.n_outs = 1;
    JobSets js;
    js.hash.push_back(p);
    emit(as, js, st.dev(o_st) + off, 1);
  });
  if (rc) return rc;
  AFX_HIP(hipMemcpyAsync(out64, st.dev(o_out), 64 * count, hipMemcpyDeviceToHost, ctx->stream));
  AFX_HIP(hipStreamSynchronize(ctx->stream));
  return AFX_OK;
} catch (...) { return afx::exception_rc(); }
extern "C" int afx_points_validate(afx_ctx* ctx, const uint8_t* pts, size_t count, uint8_t* ok, uint8_t* reencoded) try {
  CtxLock lock__(ctx);
  if (!ctx || !pts || !ok) { set_error("null argument"); return AFX_E_BAD_ARGS; }
  if (count == 0) return AFX_OK;
  AFX_HIP(hipSetDevice(ctx->device));
  Stager st(ctx);
  const size_t o_in = st.add(pts, 32 * count), o_ok = st.add(nullptr, count), o_re = st.add(nullptr, 32 * count);
  int rc = st.upload();
  if (rc) return rc;
  AFX_HIP(afxk_validate(ctx->stream, st.dev(o_in), st.dev(o_ok), reencoded ? st.dev(o_re) : nullptr, (uint32_t)count));
  AFX_HIP(hipMemcpyAsync(ok, st.dev(o_ok), count, hipMemcpyDeviceToHost, ctx->stream));
  if (reencoded) AFX_HIP(hipMemcpyAsync(reencoded, st.dev(o_re), 32 * count, hipMemcpyDeviceToHost, ctx->stream));
  AFX_HIP(hipStreamSynchronize(ctx->stream));
  return AFX_OK;
} catch (...) { return afx::exception_rc(); }
extern "C" int afx_multiscalar_mul(afx_ctx* ctx, uint32_t n_terms, const uint8_t* scalars, const uint8_t* points, size_t count, uint8_t* out, uint8_t* ok) try {
  CtxLock lock__(ctx);
  if (!ctx || !scalars || !points || !out || !ok) { set_error("null argument"); return AFX_E_BAD_ARGS; }
  if (n_terms == 0 || n_terms > AFX_MSM_MAX_TERMS) { set_error("n_terms out of range"); return AFX_E_BAD_ARGS; }
  if (count == 0) return AFX_OK;
  AFX_HIP(hipSetDevice(ctx->device));
  Stager st(ctx);
  const size_t o_s = st.add(scalars, 32 * count * n_terms), o_p = st.add(points, 32 * count * n_terms), o_out = st.add(nullptr, 32 * count),
               o_ok = st.add(nullptr, count);
  int rc = st.upload();
  if (rc) return rc;
  rc = run_chunked(ctx, count, [&](Assembler& as, size_t off, uint32_t) {
    JobSets js;
    std::vector<afx_msm_term> terms;
    for (uint32_t k = 0; k < n_terms; k++) {
      int32_t* v = as.new_var();
      js.sccheck.push_back({ st.dev(o_s) + (k * count + off) * 32 });
      js.decode.push_back({ st.dev(o_p) + (k * count + off) * 32, v, 0 });
      terms.push_back(mk_term(st.dev(o_s) + (k * count + off) * 32, 32, v, -1, false));
    }
    afx_msm_job j;
    memset(&j, 0, sizeof j);
    set_terms(j, terms);
    j.out_enc = st.dev(o_out) + off * 32;
    js.msm1.push_back(j);
    emit(as, js, st.dev(o_ok) + off, 1);
  });
  if (rc) return rc;
  AFX_HIP(hipMemcpyAsync(out, st.dev(o_out), 32 * count, hipMemcpyDeviceToHost, ctx->stream));
  std::vector<uint8_t> bad(count);
  AFX_HIP(hipMemcpyAsync(bad.data(), st.dev(o_ok), count, hipMemcpyDeviceToHost, ctx->stream));
  AFX_HIP(hipStreamSynchronize(ctx->stream));
  for (size_t i = 0; i < count; i++) ok[i] = bad[i] ? 0 : 1;
  return AFX_OK;
} catch (...) { return afx::exception_rc(); }

// ------------------------------------------------------------------------------------------------
// wire format: header parsing on the host, AoS -> SoA transposition on the GPU
// ------------------------------------------------------------------------------------------------
static bool wire_shape_ok(const afx_shape& sh) {
  if (sh.n_attributes > AFX_MAX_ATTRIBUTES || sh.n_responses > 3 + AFX_MAX_ATTRIBUTES || sh.n_hidden_scalars > AFX_MAX_ATTRIBUTES ||
      sh.n_enc_proofs > AFX_MAX_ATTRIBUTES)
    return false;
  for (uint32_t i = 0; i < sh.n_attributes; i++)
    if (sh.kinds[i] > AFX_ENC_SECRET_POINT) return false;
  return true;
}
extern "C" uint32_t afx_wire_cells_per_record(const afx_shape* sh) {
  if (!sh || !wire_shape_ok(*sh)) return 0;
  uint32_t pub = 0;
  for (uint32_t i = 0; i < sh->n_attributes; i++) pub += (sh->kinds[i] == AFX_ENC_PUBLIC_SCALAR || sh->kinds[i] == AFX_ENC_PUBLIC_POINT);
  return 1 + sh->n_responses + 3 + sh->n_attributes + pub + 14 * sh->n_enc_proofs;
}
extern "C" size_t afx_wire_header_bytes(const afx_shape* sh) {
  if (!sh || !wire_shape_ok(*sh)) return 0;
  const size_t raw = 32 + sh->n_attributes + 2 * (size_t)sh->n_hidden_scalars + 2 * (size_t)sh->n_enc_proofs;
  return (raw + 31) & ~size_t(31);
}
extern "C" int afx_wire_parse(const uint8_t* blob, size_t len, afx_shape* shape_out, size_t* count_out, size_t* records_offset_out) try {
  if (!blob || !shape_out || !count_out || !records_offset_out) { set_error("null argument"); return AFX_E_BAD_ARGS; }
  if (len < 32 || memcmp(blob, "AFXP", 4) != 0 || rd32(blob + 4) != 1) { set_error("not an AFXP v1 batch"); return AFX_E_BAD_ARGS; }
  afx_shape sh;
  memset(&sh, 0, sizeof sh);
  const uint32_t count = rd32(blob + 8), cells = rd32(blob + 12);
  sh.n_attributes = rd32(blob + 16); sh.n_responses = rd32(blob + 20); sh.n_hidden_scalars = rd32(blob + 24); sh.n_enc_proofs = rd32(blob + 28);
  if (sh.n_attributes > AFX_MAX_ATTRIBUTES || sh.n_responses > 3 + AFX_MAX_ATTRIBUTES || sh.n_hidden_scalars > AFX_MAX_ATTRIBUTES || sh.n_enc_proofs > AFX_MAX_ATTRIBUTES) {
    set_error("shape field out of range");
    return AFX_E_BAD_ARGS;
  }
  const size_t raw = 32 + sh.n_attributes + 2 * (size_t)sh.n_hidden_scalars + 2 * (size_t)sh.n_enc_proofs, hdr = (raw + 31) & ~size_t(31);
  if (len < hdr) { set_error("truncated header"); return AFX_E_BAD_ARGS; }
  const uint8_t* p = blob + 32;
  for (uint32_t i = 0; i < sh.n_attributes; i++) sh.kinds[i] = *p++;
  for (uint32_t i = 0; i < sh.n_hidden_scalars; i++) { sh.hidden_scalar_indices[i] = (uint16_t)(p[0] | (p[1] << 8)); p += 2; }
  for (uint32_t i = 0; i < sh.n_enc_proofs; i++) { sh.enc_indices[i] = (uint16_t)(p[0] | (p[1] << 8)); p += 2; }
  if (!wire_shape_ok(sh) || cells != afx_wire_cells_per_record(&sh)) { set_error("cells_per_record does not match the shape"); return AFX_E_BAD_ARGS; }
  if ((len - hdr) / 32 / (cells ? cells : 1) < count || len != hdr + (size_t)count * cells * 32) { set_error("record area length"); return AFX_E_BAD_ARGS; }
  *shape_out = sh;
  *count_out = count;
  *records_offset_out = hdr;
  return AFX_OK;
} catch (...) { return afx::exception_rc(); }
// records [first, first + n) of a parsed AFXP batch; status is the whole batch's array (element first + i answers record first + i)
static int verify_wire_records(afx_ctx* ctx, const afx_shape& sh, const uint8_t* records, size_t count, size_t first, size_t n, uint8_t* status) {
  if (n == 0) return AFX_OK;
  AFX_HIP(hipSetDevice(ctx->device));
  const uint32_t cells = afx_wire_cells_per_record(&sh);
  // SoA rows: challenge | responses | C_x_0 C_x_1 C_V | C_y[n] | attr_values[n] (secret rows stay unused) | enc proofs
  const uint32_t na = sh.n_attributes;
  const uint32_t row_resp = 1, row_cx = row_resp + sh.n_responses, row_cy = row_cx + 3, row_av = row_cy + na, row_enc = row_av + na;
  const uint32_t rows = row_enc + 14 * sh.n_enc_proofs;
  std::vector<uint32_t> row_of_cell;
  for (uint32_t r = 0; r < row_av; r++) row_of_cell.push_back(r);
  for (uint32_t i = 0; i < na; i++)
    if (sh.kinds[i] == AFX_ENC_PUBLIC_SCALAR || sh.kinds[i] == AFX_ENC_PUBLIC_POINT) row_of_cell.push_back(row_av + i);
  for (uint32_t r = row_enc; r < rows; r++) row_of_cell.push_back(r);
  if (row_of_cell.size() != cells) { set_error("internal: wire cell map"); return AFX_E_BAD_ARGS; }
  // records are contiguous: a slice of the batch is a byte range of the blob; slices alternate between the two lanes like
  // those of the column-array calls (statements.hpp host_pipe), each transposed on the GPU into its own SoA scratch
  const afx_shape jsh = canonical_shape(sh);
  const PlanKey jkey = plan_key("VW", &jsh, sizeof jsh, mode_flags(ctx));
  return host_pipe(ctx, n, [&](Stager& st, size_t off, size_t sn) -> int {
    const size_t f0 = first + off;
    st.layout_tag = 1;
    const size_t dn = st.dev_items(sn);
    const size_t o_rec = st.add_rows(records, 1, (size_t)cells * 32, count, f0, sn, dn), o_map = st.add((const uint8_t*)row_of_cell.data(), 4 * (size_t)cells),
                 o_soa = st.reserve(dn * rows * 32), o_st = st.add(nullptr, dn);
    st.plan_fetch(status, o_st, 1, 1, count, f0, sn, dn);
    int rc2 = st.upload();
    if (rc2) return rc2;
    if (!st.app) {   // (a call that took item slots of an earlier call's pass: that call's transposition covers them)
      // the transposition reads what the upload brings: under a Session it waits for the session's one upload
      hipStream_t strm = st.stream();
      const uint8_t* rec_d = st.dev(o_rec);
      uint8_t* soa_d = st.dev(o_soa);
      const uint32_t* map_d = (const uint32_t*)st.dev(o_map);
      const uint32_t cells_ = cells, dn_ = (uint32_t)dn;
      auto transpose = [=]() -> int { AFX_HIP(afxk_aos_to_soa(strm, rec_d, soa_d, map_d, cells_, dn_)); return AFX_OK; };
      if (st.ses) st.ses->pre.push_back(transpose);
      else if ((rc2 = transpose())) return rc2;
    }
    auto rowp = [&](uint32_t r) { return (const uint8_t*)st.dev(o_soa) + (size_t)r * dn * 32; };
    afx_presentation_soa d;
    d.challenge = rowp(0);
    d.responses = rowp(row_resp);
    d.C_x_0 = rowp(row_cx); d.C_x_1 = rowp(row_cx + 1); d.C_V = rowp(row_cx + 2);
    d.C_y = rowp(row_cy);
    d.attr_values = rowp(row_av);
    std::vector<afx_encproof_soa> encs(sh.n_enc_proofs);
    for (uint32_t e = 0; e < sh.n_enc_proofs; e++) {
      const uint32_t r = row_enc + 14 * e;
      encs[e] = { rowp(r), rowp(r + 1), rowp(r + 7), rowp(r + 8), rowp(r + 9), rowp(r + 10), rowp(r + 11), rowp(r + 12), rowp(r + 13) };
    }
    d.enc = encs.data();
    if ((rc2 = afx_verify_presentations_dev(ctx, &sh, &d, dn, st.dev(o_st)))) return rc2;
    return st.fetch_all();
  }, jkey);
}
extern "C" int afx_verify_presentations_wire_range(afx_ctx* ctx, const uint8_t* blob, size_t len, size_t first, size_t n, uint8_t* status, size_t status_cap,
                                                   size_t* count_out) try {
  CtxLock lock__(ctx, true);
  if (!ctx || !status || !count_out) { set_error("null argument"); return AFX_E_BAD_ARGS; }
  afx_shape sh;
  size_t count = 0, off = 0;
  int rc = afx_wire_parse(blob, len, &sh, &count, &off);
  if (rc) return rc;
  *count_out = count;
  if (first > count || n > count - first) { set_error("range outside the batch"); return AFX_E_BAD_ARGS; }
  if (n == 0) return AFX_OK;
  if (status_cap < count) { set_error("status buffer too small"); return AFX_E_BAD_ARGS; }
  if (count > 0xffffffffu / 64) { set_error("batch too large for one call"); return AFX_E_BAD_ARGS; }
  if (!ctx->has_key) { set_error("Issuer::verify needs the issuer key"); return AFX_E_NO_KEY; }
  return verify_wire_records(ctx, sh, blob + off, count, first, n, status);
} catch (...) { return afx::exception_rc(); }
extern "C" int afx_verify_presentations_wire(afx_ctx* ctx, const uint8_t* blob, size_t len, uint8_t* status, size_t status_cap, size_t* count_out) try {
  if (!count_out) { set_error("null argument"); return AFX_E_BAD_ARGS; }
  size_t count = 0, hdr = 0;
  afx_shape sh;
  int rc = afx_wire_parse(blob, len, &sh, &count, &hdr);
  if (rc) return rc;
  *count_out = count;
  return afx_verify_presentations_wire_range(ctx, blob, len, 0, count, status, status_cap, count_out);
} catch (...) { return afx::exception_rc(); }

// ---- CredentialIssuance batches ("AFXI" v1) -----------------------------------------------------
extern "C" size_t afx_issuance_wire_header_bytes(uint32_t n_attributes) {
  if (n_attributes > AFX_MAX_ATTRIBUTES) return 0;
  return (24 + (size_t)n_attributes + 31) & ~size_t(31);
}
extern "C" int afx_issuance_wire_parse(const uint8_t* blob, size_t len, uint32_t* n_out, uint8_t kinds_out[AFX_MAX_ATTRIBUTES], uint32_t* nr_out,
                                       size_t* count_out, size_t* records_offset_out) try {
  if (!blob || !n_out || !kinds_out || !nr_out || !count_out || !records_offset_out) { set_error("null argument"); return AFX_E_BAD_ARGS; }
  if (len < 24 || memcmp(blob, "AFXI", 4) != 0 || rd32(blob + 4) != 1) { set_error("not an AFXI v1 batch"); return AFX_E_BAD_ARGS; }
  const uint32_t count = rd32(blob + 8), cells = rd32(blob + 12), n = rd32(blob + 16), nr = rd32(blob + 20);
  if (n > AFX_MAX_ATTRIBUTES || nr > AFX_MAX_ATTRIBUTES + 5) { set_error("layout field out of range"); return AFX_E_BAD_ARGS; }
  const size_t hdr = afx_issuance_wire_header_bytes(n);
  if (len < hdr) { set_error("truncated header"); return AFX_E_BAD_ARGS; }
  for (uint32_t i = 0; i < n; i++)
    if (blob[24 + i] > AFX_ATTR_SECRET_POINT) { set_error("attribute kind out of range"); return AFX_E_BAD_ARGS; }
  if (cells != 4 + nr + n) { set_error("cells_per_record does not match the layout"); return AFX_E_BAD_ARGS; }
  if ((len - hdr) / 32 / cells < count || len != hdr + (size_t)count * cells * 32) { set_error("record area length"); return AFX_E_BAD_ARGS; }
  memset(kinds_out, 0, AFX_MAX_ATTRIBUTES);
  memcpy(kinds_out, blob + 24, n);
  *n_out = n; *nr_out = nr; *count_out = count; *records_offset_out = hdr;
  return AFX_OK;
} catch (...) { return afx::exception_rc(); }
extern "C" int afx_verify_issuances_wire(afx_ctx* ctx, const uint8_t* blob, size_t len, uint8_t* status, size_t status_cap, size_t* count_out) try {
  CtxLock lock__(ctx, true);
  if (!ctx || !status || !count_out) { set_error("null argument"); return AFX_E_BAD_ARGS; }
  afx_attributes_soa at;
  memset(&at, 0, sizeof at);
  uint32_t nr = 0;
  size_t count = 0, off = 0;
  int rc = afx_issuance_wire_parse(blob, len, &at.n_attributes, at.kinds, &nr, &count, &off);
  if (rc) return rc;
  *count_out = count;
  if (count == 0) return AFX_OK;
  if (status_cap < count) { set_error("status buffer too small"); return AFX_E_BAD_ARGS; }
  if (count > 0xffffffffu / 64) { set_error("batch too large for one call"); return AFX_E_BAD_ARGS; }
  AFX_HIP(hipSetDevice(ctx->device));
  const uint32_t n = at.n_attributes, cells = 4 + nr + n;
  std::vector<uint32_t> row_of_cell(cells);
  for (uint32_t r = 0; r < cells; r++) row_of_cell[r] = r;   // SoA rows in record order: t U V challenge responses[] values[]
  struct { uint32_t n, nr; uint8_t kinds[AFX_MAX_ATTRIBUTES]; } jd;
  memset(&jd, 0, sizeof jd);
  jd.n = n; jd.nr = nr; memcpy(jd.kinds, at.kinds, AFX_MAX_ATTRIBUTES);
  const PlanKey jkey = plan_key("VIW", &jd, sizeof jd, mode_flags(ctx));
  return host_pipe(ctx, count, [&](Stager& st, size_t first, size_t sn) -> int {
    st.layout_tag = 1;
    const size_t dn = st.dev_items(sn);
    const size_t o_rec = st.add_rows(blob + off, 1, (size_t)cells * 32, count, first, sn, dn), o_map = st.add((const uint8_t*)row_of_cell.data(), 4 * (size_t)cells),
                 o_soa = st.reserve(dn * cells * 32), o_st = st.add(nullptr, dn);
    st.plan_fetch(status, o_st, 1, 1, count, first, sn, dn);
    int rc2 = st.upload();
    if (rc2) return rc2;
    if (!st.app) {
      // the transposition reads what the upload brings: under a Session it waits for the session's one upload
      hipStream_t strm = st.stream();
      const uint8_t* rec_d = st.dev(o_rec);
      uint8_t* soa_d = st.dev(o_soa);
      const uint32_t* map_d = (const uint32_t*)st.dev(o_map);
      const uint32_t cells_ = cells, dn_ = (uint32_t)dn;
      auto transpose = [=]() -> int { AFX_HIP(afxk_aos_to_soa(strm, rec_d, soa_d, map_d, cells_, dn_)); return AFX_OK; };
      if (st.ses) st.ses->pre.push_back(transpose);
      else if ((rc2 = transpose())) return rc2;
    }
    auto rowp = [&](uint32_t r) { return (uint8_t*)st.dev(o_soa) + (size_t)r * dn * 32; };
    afx_attributes_soa as = at;
    as.values = rowp(4 + nr);
    const afx_issuance_soa iss = { rowp(0), rowp(1), rowp(2), rowp(3), rowp(4) };
    if ((rc2 = afx_verify_issuances_dev(ctx, &as, &iss, nr, dn, st.dev(o_st)))) return rc2;
    return st.fetch_all();
  }, jkey);
} catch (...) { return afx::exception_rc(); }
extern "C" int afx_ctx_issuer_parameters(afx_ctx* c, uint8_t out[64]) try {
  if (!c || !out) { set_error("null argument"); return AFX_E_BAD_ARGS; }
  CtxLock lock(c);
  memcpy(out, c->gen_enc[c->id_CW()].data(), 32);
  memcpy(out + 32, c->gen_enc[c->id_I()].data(), 32);
  return AFX_OK;
} catch (...) { return afx::exception_rc(); }
