// How a statement's plan gets to the device: assembled (or taken from the context's cache of position-independent plans),
// placed and launched - alone, or together with the plans of other small calls collected by a Session - and how host-pointer
// front ends stage their arrays (Stager, host_pipe).  Declarations and the design notes: statements.hpp, engine.hpp (afx::Plan).
#include <stdlib.h>
#include "statements.hpp"

// ------------------------------------------------------------------------------------------------
// run_chunked
// ------------------------------------------------------------------------------------------------
namespace {
// assemble the plan of one pass; with plan_selfcheck, twice against different provisional bases: both copies, relocated to the
// same place, must be equal byte for byte (a pointer field Plan::relocate does not know would differ)
int assemble(afx_ctx* c, const BuildFn& build, size_t off, uint32_t cc, const Stager* st, Plan& plan) {
  uint8_t* in_base = st && st->uploaded ? st->in_base() : nullptr;
  uint8_t* out_base = st && st->uploaded ? st->out_base() : nullptr;
  const size_t in_bytes = st && st->uploaded ? st->in_bytes : 0, out_bytes = st && st->uploaded ? st->out_bytes : 0;
  {
    Assembler as(c, cc, 0);
    build(as, off, cc);
    const int rc = as.finish_plan(plan, in_base, in_bytes, out_base, out_bytes);
    if (rc) return rc;
  }
  if (c->plan_selfcheck) {
    Plan twin;
    {
      Assembler as(c, cc, 1);
      build(as, off, cc);
      const int rc = as.finish_plan(twin, in_base, in_bytes, out_base, out_bytes);
      if (rc) return rc;
    }
    Plan a = plan;
    uint8_t* where_blob = (uint8_t*)(uintptr_t)0x4000f00000000000ull;
    uint8_t* where_ws = (uint8_t*)(uintptr_t)0x4000f80000000000ull;
    a.relocate(where_blob, where_ws, a.in_base, a.out_base);
    twin.relocate(where_blob, where_ws, twin.in_base, twin.out_base);
    std::string why;
    if (!a.same_as(twin, &why)) { set_error("plan self-check: " + why); return AFX_E_BAD_ARGS; }
  }
  return AFX_OK;
}
}  // namespace

int run_chunked(afx_ctx* c, size_t count, const BuildFn& build, const PlanKey& key) {
  AFX_HIP(hipSetDevice(c->device));
  Stager* st = c->cur_stager;
  afx::Session* ses = (c->session && !c->session->paused) ? c->session : nullptr;
  // all chunks of one call run on one lane; calls alternate lanes only when the caller switched pipelining on
  const int lane = ses ? ses->lane : (c->force_lane >= 0 ? c->force_lane : (c->pipelining ? (int)(c->lane_next++ & 1u) : 0));
  uint32_t chunk = c->chunk_items ? c->chunk_items : CHUNK_DEFAULT;
  for (size_t off = 0; off < count;) {
    const uint32_t cc = (uint32_t)std::min<size_t>(chunk, count - off);
    try {
      // a small host-pointer call (one pass, arrays staged by a Stager): its plan is kept and reused
      const bool reusable = !key.empty() && st && st->uploaded && cc == count && c->small_batch_items && cc <= c->small_batch_items && !c->trace;
      std::unique_ptr<Plan> plan(new Plan());
      const std::pair<PlanKey, uint32_t> ck(key, cc);
      auto hit = reusable ? c->plan_cache.find(ck) : c->plan_cache.end();
      if (hit != c->plan_cache.end() && hit->second->in_bytes == st->in_bytes && hit->second->out_bytes == st->out_bytes) {
        *plan = *hit->second;
        plan->relocate(plan->blob_base, plan->ws_base, st->in_base(), st->out_base());
        if (c->plan_selfcheck) {   // a reused plan must equal a fresh one
          Plan fresh;
          int rc = assemble(c, build, off, cc, st, fresh);
          if (rc) return rc;
          std::string why;
          if (!plan->same_as(fresh, &why)) { set_error("plan self-check (reuse): " + why); return AFX_E_BAD_ARGS; }
        }
      } else {
        int rc = assemble(c, build, off, cc, st, *plan);
        if (rc) return rc;
        if (reusable && c->plan_cache.size() < PLAN_CACHE_ENTRIES && c->plan_cache_bytes + plan->blob.size() <= PLAN_CACHE_BYTES) {
          if (hit != c->plan_cache.end()) { c->plan_cache_bytes -= hit->second->blob.size(); c->plan_cache.erase(hit); }
          c->plan_cache[ck] = std::make_shared<Plan>(*plan);
          c->plan_cache_bytes += plan->blob.size();
        }
      }
      if (ses) {
        ses->plans.push_back(std::move(plan));
      } else {
        // the device cannot hold this pass's workspace: take smaller passes (an engine on a shared or smaller GPU still works)
        int rc = c->lane[lane].ws.ensure(plan->ws_bytes);
        if (rc) {
          if (chunk <= 4096 || cc <= 4096) return rc;
          (void)hipGetLastError();
          chunk >>= 1;
          continue;
        }
        Plan* one = plan.get();
        if ((rc = run_plans(c, lane, &one, 1))) return rc;
      }
    } catch (const std::exception& e) {
      set_error(std::string("plan assembly: ") + e.what());
      return AFX_E_BAD_ARGS;
    }
    off += cc;
  }
  return AFX_OK;
}

// ------------------------------------------------------------------------------------------------
// Stager
// ------------------------------------------------------------------------------------------------
// AFX_PACK_LIMIT_MB (measurement aid): up to how many bytes a call's rows are gathered into one pinned image
size_t Stager::pack_limit() {
  static const size_t lim = [] { const char* e = getenv("AFX_PACK_LIMIT_MB"); return e ? (size_t)strtoull(e, nullptr, 10) << 20 : PACK_LIMIT; }();
  return lim;
}
static int ensure_pinned(void*& buf, size_t& cap, size_t bytes, size_t granule) {
  if (bytes <= cap) return AFX_OK;
  if (buf) { memset(buf, 0, cap); (void)hipHostFree(buf); buf = nullptr; cap = 0; }
  const size_t want = (bytes + granule - 1) & ~(granule - 1);
  void* p = nullptr;
  AFX_HIP(hipHostMalloc(&p, want, hipHostMallocDefault));
  buf = p;
  cap = want;
  return AFX_OK;
}

int Stager::upload() {
  afx_ctx::Lane& L = c->lane[ln];
  if (ses) {
    // this call's regions inside the session's images; a session that cannot take them is flushed first (and, empty, grown)
    size_t ia = (ses->in_used + 255) & ~size_t(255), oa = (ses->out_used + 255) & ~size_t(255);
    if (ia + in_bytes + 256 > L.staging.cap || oa + out_bytes + 256 > L.staging_out.cap || ia + in_bytes > L.pin_in_cap || oa + out_bytes > L.pin_cap) {
      int rc = ses->flush();
      if (rc) return rc;
      if ((rc = ses->ensure_images(in_bytes + 256, out_bytes + 256))) return rc;
      ia = oa = 0;
    }
    in_at = ia; out_at = oa;
    ses->in_used = ia + in_bytes;
    ses->out_used = oa + out_bytes;
    uint8_t* img = (uint8_t*)L.pin_in + in_at;
    memset(img, 0, in_bytes);
    for (const Copy& k : copies) memcpy(img + k.off, k.src, k.len);
    uploaded = true;
    return AFX_OK;
  }
  int rc = L.staging.ensure(in_bytes + 256);
  if (rc) return rc;
  if ((rc = L.staging_out.ensure(out_bytes + 256))) return rc;
  uploaded = true;
  // result areas start from zero: the buffers are reused from call to call, and what a call does not write (the outputs of a
  // failed item, the hidden rows of attr_values) must not hand an earlier call's bytes to this caller
  if (out_bytes) AFX_HIP(hipMemsetAsync(L.staging_out.p, 0, out_bytes, L.stream));
  if (in_bytes <= pack_limit() && copies.size() > 2) {
    // the event first: a buffer is only published together with the event that guards its reuse
    if (!L.pin_in_done) AFX_HIP(hipEventCreateWithFlags(&L.pin_in_done, hipEventDisableTiming));
    AFX_HIP(hipEventSynchronize(L.pin_in_done));   // the previous call's transfer out of this buffer
    if ((rc = ensure_pinned(L.pin_in, L.pin_in_cap, std::max<size_t>(in_bytes, size_t(1) << 18), 65536))) return rc;
    uint8_t* img = (uint8_t*)L.pin_in;
    // The image starts from zero: what no copy covers - reserve() scratch, the padding between and behind rows - would otherwise
    // carry an EARLIER call's bytes (staged keys, seeds) into this call's staging area (at most 4 MB: ~50 us)
    memset(img, 0, in_bytes);
    for (const Copy& k : copies) memcpy(img + k.off, k.src, k.len);
    AFX_HIP(hipMemcpyAsync(L.staging.p, img, in_bytes, hipMemcpyHostToDevice, L.stream));
    AFX_HIP(hipEventRecord(L.pin_in_done, L.stream));
    return AFX_OK;
  }
  for (const Copy& k : copies) AFX_HIP(hipMemcpyAsync((uint8_t*)L.staging.p + k.off, k.src, k.len, hipMemcpyHostToDevice, L.stream));
  for (const Copy& k : zeros) AFX_HIP(hipMemsetAsync((uint8_t*)L.staging.p + k.off, 0, k.len, L.stream));
  return AFX_OK;
}

int Stager::fetch_all() {
  afx_ctx::Lane& L = c->lane[ln];
  if (ses) {
    // declared, not fetched: the session's flush brings its whole output image back and scatters
    for (const Out& o : outs) ses->outs.push_back({ o.dst, out_at + o.pin_off, o.len });
    outs.clear();
    pend_.clear();
    return AFX_OK;
  }
  fetched_ = true;
  // a small call's output region in ONE copy (its rows are scattered over it); large calls row block by row block
  whole_ = out_bytes <= pack_limit();
  if (whole_) {
    int rc = ensure_pinned(L.pin, L.pin_cap, std::max<size_t>(out_bytes, 1), size_t(1) << 20);
    if (rc) return rc;
    if (!pend_.empty()) AFX_HIP(hipMemcpyAsync(L.pin, L.staging_out.p, out_bytes, hipMemcpyDeviceToHost, L.stream));
    return AFX_OK;
  }
  size_t need = 0;
  for (const Pend& p : pend_) need += (p.len + 63) & ~size_t(63);
  int rc = ensure_pinned(L.pin, L.pin_cap, need, size_t(1) << 20);
  if (rc) return rc;
  // pack the blocks one behind the other in the pinned buffer; every row's pin offset moves with its block
  size_t at = 0, oi = 0;
  for (const Pend& p : pend_) {
    AFX_HIP(hipMemcpyAsync((uint8_t*)L.pin + at, (const uint8_t*)L.staging_out.p + p.off, p.len, hipMemcpyDeviceToHost, L.stream));
    for (; oi < outs.size() && outs[oi].pin_off >= p.off && outs[oi].pin_off < p.off + p.len; oi++) outs[oi].pin_off = at + (outs[oi].pin_off - p.off);
    at += (p.len + 63) & ~size_t(63);
  }
  return AFX_OK;
}

int Stager::drain() {
  if (ses) return AFX_OK;
  afx_ctx::Lane& L = c->lane[ln];
  AFX_HIP(hipStreamSynchronize(L.stream));
  if (fetched_)   // (a slice that failed before fetch_all() declared results it never produced)
    for (const Out& o : outs) memcpy(o.dst, (const uint8_t*)L.pin + o.pin_off, o.len);
  outs.clear();
  pend_.clear();
  return AFX_OK;
}

// ------------------------------------------------------------------------------------------------
// host_pipe
// ------------------------------------------------------------------------------------------------
int host_pipe(afx_ctx* c, size_t count, const std::function<int(Stager&, size_t, size_t)>& slice) {
  if (c->session && !c->session->paused) {
    // collected: one slice, staged into the session's images; nothing is waited for here
    Stager st(c, c->session->lane, c->session);
    return slice(st, 0, count);
  }
  const int entry_force = c->force_lane;
  Stager* const entry_stager = c->cur_stager;
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
    st[1].reset(); st[0].reset();
    c->force_lane = entry_force;
    c->cur_stager = entry_stager;
    throw;
  }
  for (int k = 0; k < 2; k++) {
    const int lane = (int)((i + k) & 1);   // oldest first
    if (st[lane]) { const int r2 = st[lane]->drain(); if (!rc) rc = r2; }
  }
  // a failed slice may leave work in flight on the other lane: wait before the Stagers (and the caller's arrays) go away
  if (rc) for (auto& L : c->lane) if (L.stream) (void)hipStreamSynchronize(L.stream);
  // (destroyed in reverse order of construction is not guaranteed here: each restores what it saw, the entry values are put back below)
  st[1].reset(); st[0].reset();
  c->force_lane = entry_force;
  c->cur_stager = entry_stager;
  return rc;
}

// ------------------------------------------------------------------------------------------------
// Session
// ------------------------------------------------------------------------------------------------
int afx::Session::ensure_images(size_t in_bytes, size_t out_bytes) {
  afx_ctx::Lane& L = c->lane[lane];
  // start generous: a session of a few dozen small calls never has to flush early
  in_bytes = std::max(in_bytes, size_t(8) << 20);
  out_bytes = std::max(out_bytes, size_t(8) << 20);
  int rc = L.staging.ensure(in_bytes);
  if (rc) return rc;
  if ((rc = L.staging_out.ensure(out_bytes))) return rc;
  if (L.pin_in_done) AFX_HIP(hipEventSynchronize(L.pin_in_done));
  if ((rc = ensure_pinned(L.pin_in, L.pin_in_cap, in_bytes, 65536))) return rc;
  return ensure_pinned(L.pin, L.pin_cap, out_bytes, size_t(1) << 20);
}

int afx::Session::flush() {
  if (empty()) return AFX_OK;
  afx_ctx::Lane& L = c->lane[lane];
  hipStream_t s = L.stream;
  AFX_HIP(hipSetDevice(c->device));
  int rc = AFX_OK;
  if (in_used) AFX_HIP(hipMemcpyAsync(L.staging.p, L.pin_in, in_used, hipMemcpyHostToDevice, s));
  if (out_used) AFX_HIP(hipMemsetAsync(L.staging_out.p, 0, out_used, s));
  for (auto& f : pre)
    if ((rc = f())) break;
  if (!rc && !plans.empty()) {
    std::vector<Plan*> ps;
    for (auto& p : plans) ps.push_back(p.get());
    rc = run_plans(c, lane, ps.data(), ps.size());
  }
  if (!rc && out_used && !outs.empty()) AFX_HIP(hipMemcpyAsync(L.pin, L.staging_out.p, out_used, hipMemcpyDeviceToHost, s));
  const hipError_t e = hipStreamSynchronize(s);
  if (!rc && e != hipSuccess) { set_error(std::string("hipStreamSynchronize: ") + hipGetErrorString(e)); rc = AFX_E_HIP; }
  if (!rc)
    for (const Out& o : outs) memcpy(o.dst, (const uint8_t*)L.pin + o.pin_off, o.len);
  drop();
  return rc;
}
