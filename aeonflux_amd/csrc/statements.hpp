// Helpers shared by the statement translation units (statements.cpp: context + verification,
// statements_prove.cpp: issuance and presentation provers).
#pragma once
#include <string.h>
#include <algorithm>
#include <functional>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>
#include "engine.hpp"

namespace afx {
static constexpr size_t BLOB_CAP = size_t(4) << 20;
// Items per pass (afx_ctx_set_chunk_items).  Bounds the workspace: window tables + points in flight are ~25 KB (issue,
// n = 16) to ~70 KB (C3 verification) per item, so the default 2^19 needs 13-37 GB of the 288 GB.  Measured on 2^20-item
// batches: issue 6.21 M/s at 2^17, 6.45 at 2^18, 6.76 at 2^19, 6.84 at 2^20; C3 verification 2.42 / 2.45 / 2.45 / 2.44.
static constexpr uint32_t CHUNK_DEFAULT = 1u << 19;
}
using namespace afx;

// context construction shared with afx_issuer_keygen (issuer_params may be null there)
int afx_ctx_create_impl(afx_ctx** out, int device, const uint8_t* sp, size_t splen, const uint8_t* key, size_t klen,
                        const uint8_t* key_scalars_only, const uint8_t* issuer_params);

// ------------------------------------------------------------------------------------------------
// running a statement: size it, grow the workspace, assemble for real, launch
// ------------------------------------------------------------------------------------------------
using BuildFn = std::function<void(Assembler&, size_t /*chunk offset*/, uint32_t /*chunk count*/)>;

// The bytes that determine a plan's SIZE (statement, shape, mode flags - never the data pointers), as the cache key itself:
// no hash, so two requests share an entry only when these bytes are equal.  Callers pass a CANONICAL shape (unused array
// tails zeroed: canonical_shape), because the shape of afx_verify_presentations_wire comes from the caller's blob.
using PlanKey = std::string;   // empty: not cached
inline PlanKey plan_key(const char* statement, const void* shape, size_t shape_len, uint64_t flags) {
  PlanKey k(statement);
  k.push_back('\0');
  k.append((const char*)shape, shape_len);
  k.append((const char*)&flags, sizeof flags);
  return k;
}
inline afx_shape canonical_shape(const afx_shape& sh) {
  afx_shape c;
  memset(&c, 0, sizeof c);
  c.n_attributes = sh.n_attributes; c.n_responses = sh.n_responses; c.n_hidden_scalars = sh.n_hidden_scalars; c.n_enc_proofs = sh.n_enc_proofs;
  for (uint32_t i = 0; i < sh.n_attributes && i < AFX_MAX_ATTRIBUTES; i++) c.kinds[i] = sh.kinds[i];
  for (uint32_t i = 0; i < sh.n_hidden_scalars && i < AFX_MAX_ATTRIBUTES; i++) c.hidden_scalar_indices[i] = sh.hidden_scalar_indices[i];
  for (uint32_t i = 0; i < sh.n_enc_proofs && i < AFX_MAX_ATTRIBUTES; i++) c.enc_indices[i] = sh.enc_indices[i];
  return c;
}
// mode flags that change a plan's size, for plan_key (every statement passes them all: a flag that does not matter to a
// statement only costs it a second cache entry)
inline uint64_t mode_flags(const afx_ctx* c) {
  return (c->strict ? 1u : 0u) | (c->fixed_key_schedule ? 2u : 0u) | (c->secret_independent ? 8u : 0u) | ((uint64_t)c->small_batch_items << 8);
}

// key non-empty: the plan's workspace size is remembered per (key, pass size), so that repeated calls of one statement
// on one shape assemble their plan once per pass instead of twice (the dry sizing run is skipped).  A remembered size that
// turns out too small for the assembled plan is dropped and the pass is sized again.
inline int run_chunked(afx_ctx* c, size_t count, const BuildFn& build, const PlanKey& key = PlanKey()) {
  AFX_HIP(hipSetDevice(c->device));
  // all chunks of one call run on one lane; calls alternate lanes only when the caller switched pipelining on
  const int lane = c->force_lane >= 0 ? c->force_lane : (c->pipelining ? (int)(c->lane_next++ & 1u) : 0);
  uint32_t chunk = c->chunk_items ? c->chunk_items : CHUNK_DEFAULT;
  for (size_t off = 0; off < count;) {
    const uint32_t cc = (uint32_t)std::min<size_t>(chunk, count - off);
    try {
      size_t ws_bytes = 0;
      const std::pair<PlanKey, uint32_t> ck(key, cc);
      auto hit = !key.empty() ? c->plan_sizes.find(ck) : c->plan_sizes.end();
      const bool cached = hit != c->plan_sizes.end();
      if (cached) {
        ws_bytes = hit->second;
      } else {
        Assembler sizing(c, cc, true, lane);
        build(sizing, off, cc);
        if (!sizing.plan_error.empty()) { set_error(sizing.plan_error); return AFX_E_BAD_ARGS; }
        if (sizing.blob_bytes() > BLOB_CAP) { set_error("plan blob exceeds its fixed capacity"); return AFX_E_BAD_ARGS; }
        ws_bytes = sizing.total_ws_bytes();
        if (!key.empty() && c->plan_sizes.size() < 4096) c->plan_sizes[ck] = ws_bytes;
      }
      int rc = c->lane[lane].ws.ensure(ws_bytes);
      if (rc) {
        // the device cannot hold this pass's workspace: take smaller passes (an engine on a shared or smaller GPU still works)
        if (chunk <= 4096 || cc <= 4096) return rc;
        (void)hipGetLastError();
        chunk >>= 1;
        continue;
      }
      Assembler as(c, cc, false, lane);
      build(as, off, cc);
      if (cached && (as.total_ws_bytes() > c->lane[lane].ws.cap || as.blob_bytes() > BLOB_CAP)) {
        c->plan_sizes.erase(ck);   // the remembered size does not fit this plan: size the pass again
        continue;
      }
      if ((rc = as.run())) return rc;
    } catch (const std::exception& e) {
      set_error(std::string("plan assembly: ") + e.what());
      return AFX_E_BAD_ARGS;
    }
    off += cc;
  }
  return AFX_OK;
}

struct JobSets {
  std::vector<afx_sccheck_job> sccheck;
  std::vector<afx_decode_job> decode;
  std::vector<afx_scalarop_job> scalarop, scalarop2;   // scalarop2 runs after scalarop (products of its results)
  std::vector<afx_pointop_job> pointop;
  std::vector<afx_negenc_job> negenc;
  std::vector<afx_msm_job> msm1, msm2;
  std::vector<afx_hash_program> hash;
};

inline afx_msm_term mk_term(const uint8_t* scalar, uint32_t stride, const int32_t* var, int32_t fixed, bool neg) {
  afx_msm_term t;
  memset(&t, 0, sizeof t);
  t.scalar = scalar; t.scalar_stride = stride; t.var = var; t.fixed_idx = fixed; t.negate = neg ? 1u : 0u;
  return t;
}
inline void set_terms(afx_msm_job& j, const std::vector<afx_msm_term>& terms) {
  if (terms.size() > AFX_MSM_MAX_TERMS) throw std::length_error("too many terms in one multiscalar job");
  j.n_terms = (uint32_t)terms.size();
  j.n_var = 0;
  j.chain_to = -1;
  uint32_t k = 0;
  for (const afx_msm_term& t : terms) if (t.fixed_idx < 0) { j.term[k++] = t; j.n_var++; }
  for (const afx_msm_term& t : terms) if (t.fixed_idx >= 0) j.term[k++] = t;
}


inline void emit(Assembler& as, JobSets& js, uint8_t* status_dev, uint8_t fail_code) {
  if (!as.fail_all) {
    as.sccheck(js.sccheck);
    as.decode(js.decode);
    as.scalarop(js.scalarop);
    as.scalarop(js.scalarop2);
    as.pointop(js.pointop);
    as.negenc(js.negenc);
    as.msm(js.msm1);
    as.msm(js.msm2);
    as.hash(js.hash);
  }
  as.finish(status_dev, fail_code);
}

// Host-pointer front ends: the call's arrays staged into HBM on one of the context's two lanes.  Inputs are copied on the
// lane's stream; results come back through the lane's pinned buffer (fetch*/drain), so that a front end can keep one slice
// of a batch computing on one lane while it stages the next slice on the other (statements.cpp, HostPipe).
struct Stager {
  afx_ctx* c;
  int ln;
  int prev_force;
  size_t bytes = 0, pin_bytes = 0;
  struct Copy { size_t off; const uint8_t* src; size_t len; };
  struct Out { uint8_t* dst; size_t pin_off, len; };
  std::vector<Copy> copies, blanks;
  std::vector<Out> outs;
  // the *_dev calls made while this object lives run on its lane
  explicit Stager(afx_ctx* ctx, int lane = 0) : c(ctx), ln(lane), prev_force(ctx->force_lane) { c->force_lane = lane; }
  ~Stager() { c->force_lane = prev_force; }
  Stager(const Stager&) = delete;
  Stager& operator=(const Stager&) = delete;
  hipStream_t stream() const { return c->lane[ln].stream; }
  // reserve `len` bytes, to be filled from host `src` (or left for output when src == nullptr); returns offset
  size_t add(const uint8_t* src, size_t len) {
    const size_t off = (bytes + 255) & ~size_t(255);
    bytes = off + len;
    if (src) copies.push_back({ off, src, len });
    else if (len) blanks.push_back({ off, nullptr, len });
    return off;
  }
  // scratch that the call's own kernels fill before anything reads it (not zeroed)
  size_t reserve(size_t len) {
    const size_t off = (bytes + 255) & ~size_t(255);
    bytes = off + len;
    return off;
  }
  // items [first, first + n) of a [rows][total][elem] host array -> a contiguous [rows][n][elem] device array
  size_t add_rows(const uint8_t* src, size_t rows, size_t elem, size_t total, size_t first, size_t n) {
    const size_t off = (bytes + 255) & ~size_t(255);
    bytes = off + rows * n * elem;
    if (src)
      for (size_t r = 0; r < rows; r++) copies.push_back({ off + r * n * elem, src + (r * total + first) * elem, n * elem });
    else if (rows != 0 && n != 0 && elem != 0) blanks.push_back({ off, nullptr, rows * n * elem });   // rows nobody fills start from zero, like add(nullptr, ..)
    return off;
  }
  // Calls of few items are many short rows (75 arrays for a C3 presentation batch): each row as its own copy from pageable
  // memory costs more than the kernels gain from the latency plan.  Up to PACK_LIMIT bytes the staging area's image is put
  // together in a pinned buffer (inputs copied, result areas zeroed) and sent in ONE transfer.
  static constexpr size_t PACK_LIMIT = size_t(4) << 20;
  int upload() {
    afx_ctx::Lane& L = c->lane[ln];
    int rc = L.staging.ensure(bytes + 256);
    if (rc) return rc;
    if (bytes <= PACK_LIMIT && copies.size() > 2) {
      // the event first: a buffer is only published together with the event that guards its reuse
      if (!L.pin_in_done) AFX_HIP(hipEventCreateWithFlags(&L.pin_in_done, hipEventDisableTiming));
      if (bytes > L.pin_in_cap) {
        if (L.pin_in) { AFX_HIP(hipEventSynchronize(L.pin_in_done)); memset(L.pin_in, 0, L.pin_in_cap); (void)hipHostFree(L.pin_in); L.pin_in = nullptr; L.pin_in_cap = 0; }
        const size_t want = std::min(PACK_LIMIT, std::max<size_t>(size_t(1) << 18, (bytes + 65535) & ~size_t(65535)));
        void* fresh = nullptr;
        AFX_HIP(hipHostMalloc(&fresh, want, hipHostMallocDefault));
        L.pin_in = fresh;
        L.pin_in_cap = want;
      } else {
        AFX_HIP(hipEventSynchronize(L.pin_in_done));   // the previous call's transfer out of this buffer
      }
      uint8_t* img = (uint8_t*)L.pin_in;
      // The image starts from zero: what no copy covers - result areas, reserve() scratch, the 256-byte padding between rows -
      // would otherwise carry an EARLIER call's bytes (staged keys, seeds) into this call's staging area (at most 4 MB: ~50 us)
      memset(img, 0, bytes);
      for (const Copy& k : copies) memcpy(img + k.off, k.src, k.len);
      AFX_HIP(hipMemcpyAsync(L.staging.p, img, bytes, hipMemcpyHostToDevice, L.stream));
      AFX_HIP(hipEventRecord(L.pin_in_done, L.stream));
      return AFX_OK;
    }
    for (const Copy& k : copies) AFX_HIP(hipMemcpyAsync((uint8_t*)L.staging.p + k.off, k.src, k.len, hipMemcpyHostToDevice, L.stream));
    // result areas start from zero: the staging buffer is reused from call to call, and what a call does not write (the
    // outputs of a failed item, the hidden rows of attr_values) must not hand an earlier call's bytes to this caller
    for (const Copy& k : blanks) AFX_HIP(hipMemsetAsync((uint8_t*)L.staging.p + k.off, 0, k.len, L.stream));
    return AFX_OK;
  }
  uint8_t* dev(size_t off) const { return (uint8_t*)c->lane[ln].staging.p + off; }
  // results: the device array [rows][n][elem] at `off` goes to items [first, first + n) of the host array [rows][total][elem].
  // plan_fetch() declares them (before reserve_pin); fetch_all() enqueues the copies into the pinned buffer after the
  // kernels; drain() waits for the lane and scatters them to the caller's arrays.
  void plan_fetch(uint8_t* dst, size_t off, size_t rows, size_t elem, size_t total, size_t first, size_t n) {
    if (!dst || !rows || !n) return;
    pend_.push_back({ off, pin_bytes, rows * n * elem });
    for (size_t r = 0; r < rows; r++) outs.push_back({ dst + (r * total + first) * elem, pin_bytes + r * n * elem, n * elem });
    pin_bytes += (rows * n * elem + 63) & ~size_t(63);
  }
  int fetch_all() {
    afx_ctx::Lane& L = c->lane[ln];
    if (pin_bytes > L.pin_cap) {
      if (L.pin) { (void)hipHostFree(L.pin); L.pin = nullptr; L.pin_cap = 0; }
      const size_t want = (pin_bytes + (size_t(1) << 20) - 1) & ~((size_t(1) << 20) - 1);
      AFX_HIP(hipHostMalloc(&L.pin, want, hipHostMallocDefault));
      L.pin_cap = want;
    }
    for (const Pend& p : pend_) AFX_HIP(hipMemcpyAsync((uint8_t*)L.pin + p.pin_off, dev(p.off), p.len, hipMemcpyDeviceToHost, L.stream));
    return AFX_OK;
  }
  int drain() {
    afx_ctx::Lane& L = c->lane[ln];
    AFX_HIP(hipStreamSynchronize(L.stream));
    for (const Out& o : outs) memcpy(o.dst, (const uint8_t*)L.pin + o.pin_off, o.len);
    outs.clear();
    pend_.clear();
    return AFX_OK;
  }

 private:
  struct Pend { size_t off, pin_off, len; };
  std::vector<Pend> pend_;
};

// Items per slice of a host-pointer call: slices alternate between the two lanes, so the host-to-device copy of one
// slice overlaps the kernels of the previous one.  2^17 items keep a pass within 1-2 % of the large-pass rate.
static constexpr size_t HOST_SLICE_DEFAULT = size_t(1) << 17;
inline size_t host_slice_items(const afx_ctx* c) {
  if (c->trace) return ~size_t(0);   // the challenge trace is indexed by the item's position in ONE *_dev call
  const size_t chunk = c->chunk_items ? c->chunk_items : CHUNK_DEFAULT;
  return std::min(chunk, HOST_SLICE_DEFAULT);
}
// Runs `slice(lane, first, n)` over [0, count) in slices on alternating lanes; `slice` stages, launches and calls
// fetch_all() on the Stager it is given; the pipe drains a lane before that lane is used again, and both at the end.
inline int host_pipe(afx_ctx* c, size_t count, const std::function<int(Stager&, size_t, size_t)>& slice) {
  const int entry_force = c->force_lane;
  const size_t per = host_slice_items(c);
  std::unique_ptr<Stager> st[2];
  int rc = AFX_OK;
  size_t i = 0;
  try {
    for (size_t off = 0; off < count && !rc; i++) {
      const size_t n = std::min(per, count - off);
      const int lane = (int)(i & 1);
      if (st[lane]) { rc = st[lane]->drain(); st[lane].reset(); }
      if (rc) break;
      st[lane].reset(new Stager(c, lane));
      c->force_lane = lane;
      rc = slice(*st[lane], off, n);
      off += n;
    }
  } catch (...) {
    // copies from and to the caller's arrays may be in flight: wait for them before the exception goes on to the entry
    // point's handler (which turns it into a return code)
    for (auto& L : c->lane) if (L.stream) (void)hipStreamSynchronize(L.stream);
    st[0].reset(); st[1].reset();
    c->force_lane = entry_force;
    throw;
  }
  for (int k = 0; k < 2; k++) {
    const int lane = (int)((i + k) & 1);   // oldest first
    if (st[lane]) { const int r2 = st[lane]->drain(); if (!rc) rc = r2; }
  }
  // a failed slice may leave work in flight on the other lane: wait before the Stagers (and the caller's arrays) go away
  if (rc) for (auto& L : c->lane) if (L.stream) (void)hipStreamSynchronize(L.stream);
  st[0].reset(); st[1].reset();
  c->force_lane = entry_force;
  return rc;
}
