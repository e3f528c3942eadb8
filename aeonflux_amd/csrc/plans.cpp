// How a statement's plan gets to the device: assembled (or taken from the context's cache of position-independent plans),
// placed and launched - alone, or together with the plans of other small calls collected by a Session - and how host-pointer
// front ends stage their arrays (Stager, host_pipe).  Declarations and the design notes: statements.hpp, engine.hpp (afx::Plan).
#include <ctype.h>
#include <stdio.h>
#include <stdlib.h>
#include "statements.hpp"

// ------------------------------------------------------------------------------------------------
// where a device's host threads run; the copy pool of large host-pointer calls
// ------------------------------------------------------------------------------------------------
afx::NodeCpus afx::cpus_of_device(int device) {
  NodeCpus out;
  char bdf[64] = { 0 };
  if (hipDeviceGetPCIBusId(bdf, (int)sizeof bdf, device) != hipSuccess || !bdf[0]) return out;
  for (char* p = bdf; *p; p++) *p = (char)tolower((unsigned char)*p);
  const char* root_env = getenv("AFX_SYSFS_ROOT");
  const std::string root = root_env ? root_env : "";
  int node = -1;
  {
    FILE* f = fopen((root + "/sys/bus/pci/devices/" + bdf + "/numa_node").c_str(), "r");
    if (!f) return out;
    if (fscanf(f, "%d", &node) != 1) node = -1;
    fclose(f);
  }
  if (node < 0) return out;
  FILE* f = fopen((root + "/sys/devices/system/node/node" + std::to_string(node) + "/cpulist").c_str(), "r");
  if (!f) return out;
  char list[4096] = { 0 };
  const size_t got = fread(list, 1, sizeof list - 1, f);
  fclose(f);
  list[got] = 0;
  cpu_set_t allowed, want;
  CPU_ZERO(&want);
  if (sched_getaffinity(0, sizeof allowed, &allowed) != 0) return out;
  for (char* tok = strtok(list, ",\n"); tok; tok = strtok(nullptr, ",\n")) {
    int a = 0, b = 0;
    const int k = sscanf(tok, "%d-%d", &a, &b);
    if (k < 1) continue;
    if (k == 1) b = a;
    for (int c = a; c <= b && c < CPU_SETSIZE; c++)
      if (c >= 0 && CPU_ISSET(c, &allowed)) CPU_SET(c, &want);
  }
  if (CPU_COUNT(&want) == 0) return out;
  out.set = want;
  out.valid = true;
  return out;
}

afx::CopyPool::CopyPool(int device, uint32_t threads) : node_(cpus_of_device(device)) {
  // no more threads than the node (or, with the topology hidden, the process) has CPUs for
  cpu_set_t allowed;
  uint32_t cpus = 1;
  if (node_.valid) cpus = (uint32_t)CPU_COUNT(&node_.set);
  else if (sched_getaffinity(0, sizeof allowed, &allowed) == 0) cpus = (uint32_t)CPU_COUNT(&allowed);
  threads = std::max<uint32_t>(1, std::min(threads, cpus));
  for (uint32_t i = 1; i < threads; i++) {
    try { workers_.emplace_back([this] { loop(); }); } catch (const std::system_error&) { break; }   // (fewer threads copy, nothing fails)
  }
}
afx::CopyPool::~CopyPool() {
  { std::lock_guard<std::mutex> g(mu_); stop_ = true; }
  cv_work_.notify_all();
  for (std::thread& t : workers_) t.join();
}
void afx::CopyPool::cut(std::vector<Piece>& out, uint8_t* dst, const uint8_t* src, size_t len) {
  static constexpr size_t PIECE = size_t(1) << 20;
  for (size_t o = 0; o < len; o += PIECE) out.push_back({ dst + o, src + o, std::min(PIECE, len - o) });
}
void afx::CopyPool::work(Job& j) {
  const size_t n = j.pieces.size();
  for (;;) {
    const size_t i = j.next.fetch_add(1, std::memory_order_relaxed);
    if (i >= n) return;
    const Piece& p = j.pieces[i];
    memcpy(p.dst, p.src, p.len);
    if (j.left.fetch_sub(1, std::memory_order_acq_rel) == 1) {
      std::lock_guard<std::mutex> g(mu_);
      cv_done_.notify_all();
    }
  }
}
void afx::CopyPool::loop() {
  PinScope pin(node_, false);
  uint64_t seen = 0;
  std::unique_lock<std::mutex> lk(mu_);
  for (;;) {
    cv_work_.wait(lk, [&] { return stop_ || gen_ != seen; });
    if (stop_) return;
    seen = gen_;
    std::shared_ptr<Job> j = cur_;   // (a worker that wakes after the job is over finds none, or one with nothing left to take)
    if (!j) continue;
    lk.unlock();
    work(*j);
    lk.lock();
  }
}
void afx::CopyPool::run(std::vector<Piece> pieces) {
  if (pieces.empty()) return;
  std::lock_guard<std::mutex> one(run_mu_);
  std::shared_ptr<Job> j = std::make_shared<Job>();
  j->pieces = std::move(pieces);
  j->left.store(j->pieces.size());
  if (!workers_.empty()) {
    { std::lock_guard<std::mutex> g(mu_); cur_ = j; gen_++; }
    cv_work_.notify_all();
  }
  {
    // the caller copies too - from the device's node for the duration, like a group member's thread (its mask comes back)
    PinScope pin(node_, true);
    work(*j);
  }
  std::unique_lock<std::mutex> lk(mu_);
  cv_done_.wait(lk, [&] { return j->left.load(std::memory_order_acquire) == 0; });
  cur_.reset();
}

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
  // the call took free item slots of a pass another call of this statement, shape and mode has already left with the session
  // (afx::Session::Slots): that pass's plan covers its items too
  if (st && st->app) return AFX_OK;
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
      // (the staging layout is part of the key: the column-array and the serialized front ends of one statement stage differently)
      const std::pair<PlanKey, uint32_t> ck(reusable ? key + (char)('0' + st->layout_tag) : key, cc);
      auto hit = reusable ? c->plan_cache.find(ck) : c->plan_cache.end();
      if (hit != c->plan_cache.end() && hit->second.plan->in_bytes == st->in_bytes && hit->second.plan->out_bytes == st->out_bytes) {
        *plan = *hit->second.plan;
        hit->second.last_use = ++c->plan_cache_tick;
        c->plan_cache_hits++;
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
        if (reusable && plan->blob.size() <= PLAN_CACHE_BYTES) {
          c->plan_cache_misses++;
          if (hit != c->plan_cache.end()) { c->plan_cache_bytes -= hit->second.plan->blob.size(); c->plan_cache.erase(hit); }
          // room: the least recently used entries go (a linear scan of at most 512 entries, on a path that has just spent 0.3 ms assembling)
          while (!c->plan_cache.empty() && (c->plan_cache.size() >= PLAN_CACHE_ENTRIES || c->plan_cache_bytes + plan->blob.size() > PLAN_CACHE_BYTES)) {
            auto lru = c->plan_cache.begin();
            for (auto it = c->plan_cache.begin(); it != c->plan_cache.end(); ++it)
              if (it->second.last_use < lru->second.last_use) lru = it;
            c->plan_cache_bytes -= lru->second.plan->blob.size();
            c->plan_cache.erase(lru);   // (~Plan wipes the blob)
            c->plan_cache_evictions++;
          }
          c->plan_cache[ck] = afx_ctx::CachedPlan{ std::make_shared<Plan>(*plan), ++c->plan_cache_tick };
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
size_t Stager::pack_limit() { return PACK_LIMIT; }
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
  if (ses && app) {
    // item slots of an existing group: nothing is written before the call is known to fit them
    if (mismatch || op_i != app->ops.size()) return afx::AFX_RETRY_NOAPPEND;
    in_at = app->in_at; out_at = app->out_at;
    in_bytes = app->in_bytes; out_bytes = app->out_bytes;
    uint8_t* img = (uint8_t*)L.pin_in + in_at;
    for (const Copy& k : copies)
      if (k.constant && memcmp(img + k.off, k.src, k.len) != 0) return afx::AFX_RETRY_NOAPPEND;
    for (const Copy& k : copies)
      if (!k.constant) memcpy(img + k.off, k.src, k.len);
    uploaded = true;
    return AFX_OK;
  }
  if (ses) {
    // this call's regions inside the session's images; a session that cannot take them is flushed first (and, empty, grown)
    size_t ia = (ses->in_used + 255) & ~size_t(255), oa = (ses->out_used + 255) & ~size_t(255);
    if (ia + in_bytes + 256 > L.staging.cap || oa + out_bytes + 256 > L.staging_out.cap || ia + in_bytes > L.pin_in_cap || oa + out_bytes > L.pin_cap) {
      // (a shared session has other callers' rows in it: they are launched by its leader, this call goes to the next session)
      if (ses->shared && !ses->empty()) return afx::AFX_RETRY_FULL;
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
  if (c->host_copy_threads && in_bytes > pack_limit() && !copies.empty()) {
    // A large slice: its rows are gathered into the lane's pinned image by the context's copy pool (threads on the device's
    // NUMA node, the caller's among them) and go to HBM as one transfer per contiguous run of the staging area - instead of one
    // runtime copy per row out of pageable memory on the caller's thread, wherever that runs (afx::CopyPool, engine.hpp).
    // The image is the runs packed one behind the other: what lies between them (kernel scratch: reserve()) is not sent.
    struct Run { size_t off, len, img; };
    std::vector<Run> runs;
    size_t img_bytes = 0;
    for (const Copy& k : copies) {   // (in staging order: add() / add_rows() only ever append)
      if (!runs.empty() && k.off >= runs.back().off && k.off <= runs.back().off + runs.back().len + 4096) {
        runs.back().len = std::max(runs.back().len, k.off + k.len - runs.back().off);
      } else {
        runs.push_back({ k.off, k.len, 0 });
      }
    }
    for (Run& r : runs) { r.img = img_bytes; img_bytes += (r.len + 255) & ~size_t(255); }
    if (!c->copy_pool) {
      try { c->copy_pool.reset(new afx::CopyPool(c->device, c->host_copy_threads)); }
      catch (const std::exception& e) { set_error(std::string("copy pool: ") + e.what()); return AFX_E_NO_MEMORY; }
    }
    if (!L.pin_in_done) AFX_HIP(hipEventCreateWithFlags(&L.pin_in_done, hipEventDisableTiming));
    AFX_HIP(hipEventSynchronize(L.pin_in_done));   // the previous transfer out of this image
    if ((rc = ensure_pinned(L.pin_in, L.pin_in_cap, img_bytes, size_t(1) << 21))) return rc;
    uint8_t* img = (uint8_t*)L.pin_in;
    std::vector<afx::CopyPool::Piece> pieces;
    size_t ri = 0, covered = runs[0].off;
    for (const Copy& k : copies) {
      while (!(k.off >= runs[ri].off && k.off + k.len <= runs[ri].off + runs[ri].len)) covered = runs[++ri].off;
      // the gaps a run bridges (alignment padding, at most 4 KB each) start from zero: the image is reused from call to call
      if (k.off > covered) memset(img + runs[ri].img + (covered - runs[ri].off), 0, k.off - covered);
      covered = std::max(covered, k.off + k.len);
      afx::CopyPool::cut(pieces, img + runs[ri].img + (k.off - runs[ri].off), k.src, k.len);
    }
    c->copy_pool->run(std::move(pieces));
    for (const Run& r : runs) AFX_HIP(hipMemcpyAsync((uint8_t*)L.staging.p + r.off, img + r.img, r.len, hipMemcpyHostToDevice, L.stream));
    AFX_HIP(hipEventRecord(L.pin_in_done, L.stream));
    for (const Copy& k : zeros) AFX_HIP(hipMemsetAsync((uint8_t*)L.staging.p + k.off, 0, k.len, L.stream));
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
    // (all or none: a call that fails here must leave no pointer into its caller's arrays behind)
    if (ses->outs.capacity() < ses->outs.size() + outs.size()) ses->outs.reserve(std::max(ses->outs.size() + outs.size(), 2 * ses->outs.capacity()));
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
  if (fetched_) {   // (a slice that failed before fetch_all() declared results it never produced)
    size_t total = 0;
    for (const Out& o : outs) total += o.len;
    if (c->copy_pool && c->host_copy_threads && total >= (size_t(8) << 20)) {
      // a large slice's results (an issuance is 800 bytes, a presentation 907 and more): scattered by the copy pool
      std::vector<afx::CopyPool::Piece> pieces;
      for (const Out& o : outs) afx::CopyPool::cut(pieces, o.dst, (const uint8_t*)L.pin + o.pin_off, o.len);
      c->copy_pool->run(std::move(pieces));
    } else {
      for (const Out& o : outs) memcpy(o.dst, (const uint8_t*)L.pin + o.pin_off, o.len);
    }
  }
  outs.clear();
  pend_.clear();
  return AFX_OK;
}

// ------------------------------------------------------------------------------------------------
// host_pipe
// ------------------------------------------------------------------------------------------------
namespace {
using SliceFn = std::function<int(Stager&, size_t, size_t)>;

// The calls of one thread that were left with collecting sessions instead of being waited for one by one: the small groups of a mixed
// request (mixed.cpp).  `pending`: the sessions that carry them, each once.
thread_local afx::Deferred* tl_deferred = nullptr;

// The leader's part: launch S when it may go, wait for the device without the context, wake the callers S carried.  Entered with
// c->mu held once by S's leader, S collecting; returns the flush's code with the lock held again.
int lead(afx_ctx* c, const std::shared_ptr<afx::Session>& S) {
  afx_ctx::Coalesce& co = c->co;
  using clock = std::chrono::steady_clock;
  S->leader_defers = false;   // whoever is in here launches S: nobody takes it over from now on (coalesced_call's hand-over)
  while (S->state != afx::Session::DONE) {
    const bool go = S->full || S->hurry || co.inflight < co.max_inflight || clock::now() >= S->deadline;
    if (!go) { co.n_waited_flushes++; CtxLock::wait_until(c, S->deadline); continue; }
    S->state = afx::Session::LAUNCHING;
    co.open.reset();
    co.inflight++;
    co.cv.notify_all();   // (callers that found the session full wait for the next one to open)
    uint64_t waves = 0;
    for (const auto& p : S->plans) waves += (p->count + 63) / 64;
    co.last_waves = (uint32_t)waves;
    co.last_plans = (uint32_t)S->plans.size();
    try {
      for (const auto& kv : S->key_items) co.demand[kv.first] = kv.second;
      if (co.demand.size() > 4096) co.demand.clear();   // (keys come from callers' shapes)
    } catch (...) { }
    co.n_max_calls = std::max<uint64_t>(co.n_max_calls, S->calls);
    const clock::time_point launch_t0 = clock::now();
    // (whatever happens in here, the session is completed and its callers are woken: an exception becomes the flush's return code)
    int rc;
    try { rc = S->launch(); } catch (...) { rc = afx::exception_rc(); }
    co.launch_ns += (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(clock::now() - launch_t0).count();
    S->state = afx::Session::INFLIGHT;
    std::string err;
    {
      // the device wait and the scatter of the results run without the context: the next session collects meanwhile
      const int d = c->lock_depth;
      c->lock_depth = 0;
      c->mu.unlock();
      try { rc = S->complete(rc); } catch (...) { rc = afx::exception_rc(); }
      try { if (rc) err = afx_last_error(); } catch (...) { }
      S->finish(rc, err);          // the joined callers go home now; the context's bookkeeping follows
      c->mu.lock();
      c->lock_depth = d;
    }
    S->state = afx::Session::DONE;
    co.inflight--;
    co.lane_busy[S->lane] = false;
    co.cv.notify_all();            // leaders waiting for a launch slot, callers waiting for a lane, callers that need the context alone
  }
  if (S->rc) set_error(S->err);
  return S->rc;
}

// One small host-pointer call on a context other threads are calling too (afx_ctx::co has the protocol).  Entered with c->mu
// held once; returns with it held once - or, for a caller that only had to wait for its session, given up (CtxLock::release).
int coalesced_call(afx_ctx* c, size_t count, const SliceFn& slice, const PlanKey& jkey) {
  afx_ctx::Coalesce& co = c->co;
  using clock = std::chrono::steady_clock;
  const std::thread::id me = std::this_thread::get_id();
  std::shared_ptr<afx::Session> S;
  bool no_append = false;
  // A deferring caller (a mixed request: it comes here once per group and does not wait in between) that still LEADS a session it left
  // rows with: that session goes now if its time has come - it is full, an exclusive caller waits for it, or its deadline has passed.
  // (The header's promise - a collection is launched at the latest max_wait_us after it opened - would otherwise hold only once the
  // request has staged its last group, 0.3 ms of plan assembly per new shape later.)
  if (tl_deferred) {
    std::vector<std::shared_ptr<afx::Session>> due;
    for (const auto& P : tl_deferred->pending)
      if (P->leader == me && P->state == afx::Session::COLLECTING && (P->full || P->hurry || clock::now() >= P->deadline)) due.push_back(P);
    for (const auto& P : due) { const int rc = lead(c, P); tl_deferred->forget(P); if (rc) return rc; }
  }
  for (;;) {
    // ---- a session to stage into: the one that collects, or a new one on a free lane
    while (!co.open) {
      int lane = -1;
      if (!co.exclusive_waiters)
        for (int k = 0; k < afx_ctx::AFX_LANES && k <= co.max_inflight && lane < 0; k++)   // (one lane more than sessions may compute: the one that collects)
          if (!co.lane_busy[k]) lane = k;
      if (lane < 0) { CtxLock::wait(c); continue; }
      AFX_HIP(hipSetDevice(c->device));
      std::shared_ptr<afx::Session> n(new afx::Session(c, true));
      n->lane = lane;
      const int rc = n->ensure_images(0, 0);
      if (rc) return rc;
      n->leader = me;
      n->leader_defers = tl_deferred != nullptr;
      n->deadline = clock::now() + std::chrono::microseconds(co.max_wait_us);
      // the launches this session's plans will share, guessed from the last one's (afx_ctx::merge_class: how long the chains of a
      // latency plan are; results do not depend on it)
      n->mclass = co.last_plans > 1 ? afx_ctx::merge_class_of(co.last_waves) : 0;
      co.open = n;
      co.lane_busy[lane] = true;
      co.n_sessions++;
    }
    S = co.open;
    if (S->full) {
      if (S->leader == me) { const int rc = lead(c, S); if (tl_deferred) tl_deferred->forget(S); if (rc) return rc; continue; }   // (only a deferring caller leads a session it is not waiting in)
      CtxLock::wait(c);   // its leader is about to launch it
      continue;
    }
    // ---- stage this call's rows (and, unless they went into another call's free item slots, its plan)
    int rc;
    bool appended = false;
    const clock::time_point stage_t0 = clock::now();
    {
      struct Staging {   // the context collects into S for exactly this scope
        afx_ctx* c;
        Staging(afx_ctx* ctx, afx::Session* s) : c(ctx) { c->session = s; c->merge_class = s->mclass; }
        ~Staging() { c->session = nullptr; c->merge_class = 0; }
      } staging(c, S.get());
      Stager st(c, S->lane, S.get());
      if (!jkey.empty() && !no_append)
        for (afx::Session::Slots& g : S->slots)
          if (g.key == jkey && g.dn - g.used >= count) { st.app = &g; st.slot = g.used; break; }
      if (!st.app && !jkey.empty()) {
        auto d = co.demand.find(jkey);
        if (d != co.demand.end()) st.slots_hint = d->second;
      }
      // (nothing may leave this function by exception: a session whose leader is gone would never be launched)
      try { rc = slice(st, 0, count); } catch (...) { rc = afx::exception_rc(); }
      if (!rc) {
        // (from here on the call's rows and result declarations are with the session: nothing below may fail the call - a caller
        // that got an error would take its arrays away from under the flush)
        if (st.app) { st.app->used += (uint32_t)count; appended = true; }
        else if (!jkey.empty() && st.last_dn && st.uploaded) {
          try { S->slots.push_back({ jkey, st.ops, st.in_at, st.out_at, st.in_bytes, st.out_bytes, st.last_dn, (uint32_t)count }); }
          catch (...) { }   // out of memory: the group simply takes no joiners
        }
      }
    }
    co.staging_ns += (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(clock::now() - stage_t0).count();
    if (rc == afx::AFX_RETRY_NOAPPEND && !no_append) { no_append = true; continue; }
    if (rc == afx::AFX_RETRY_FULL) {
      S->full = true;
      co.cv.notify_all();
      if (S->leader == me) { const int rc2 = lead(c, S); if (tl_deferred) tl_deferred->forget(S); if (rc2) return rc2; continue; }
      while (co.open == S) CtxLock::wait(c);
      continue;
    }
    if (rc) {
      if (rc == afx::AFX_RETRY_NOAPPEND) { set_error("internal: a call does not reproduce its own staging layout"); rc = AFX_E_BAD_ARGS; }
      // nothing of this call is in the session (a plan is left with it last); a session its opener could not use is given up
      if (S->leader == me && S->calls == 0) {
        S->drop();
        S->state = afx::Session::DONE;
        S->finish(rc, std::string());
        co.open.reset();
        co.lane_busy[S->lane] = false;
        co.cv.notify_all();
      }
      return rc;
    }
    S->calls++;
    S->items += count;
    if (!jkey.empty()) { try { S->key_items[jkey] += (uint32_t)count; } catch (...) { } }   // (a hint for the next session's item slots)
    co.n_calls++; co.n_items += count; co.n_appended += appended;
    if (S->items >= co.max_items) { S->full = true; co.cv.notify_all(); }
    break;
  }
  // ---- a caller that defers (a mixed request staging its groups one after the other): the session is noted and the call returns;
  // afx::drain_deferred waits for - or launches - what it left behind.  A session it leads and that is full goes now.
  if (tl_deferred) {
    tl_deferred->note(S);
    if (S->leader == me && (S->full || S->hurry || clock::now() >= S->deadline)) { const int rc = lead(c, S); tl_deferred->forget(S); return rc; }
    return AFX_OK;
  }
  // ---- an ordinary caller that joined a session a deferring caller opened takes it over: it is going to wait here anyway, and the
  // opener is busy assembling its request's next group
  if (S->leader != me && S->leader_defers && S->state == afx::Session::COLLECTING) { S->leader = me; S->leader_defers = false; }
  // ---- everybody but the leader: sleep on the session's own condition, with the context given up for good - a completion wakes
  // the callers it answers, and they go home without queueing on the context's lock again
  if (S->leader != me) {
    CtxLock* mine = CtxLock::outermost();
    if (mine && mine->c == c && c->lock_depth == 1) mine->release();
    else { c->lock_depth = 0; c->mu.unlock(); }   // (not reached: host_pipe only comes here at depth 1 under a CtxLock)
    std::unique_lock<std::mutex> lk(S->done_mu);
    S->done_cv.wait(lk, [&] { return S->done; });
    if (S->rc) set_error(S->err);
    return S->rc;
  }
  return lead(c, S);
}
}  // namespace

afx::DeferScope::DeferScope(afx::Deferred* d) : prev(tl_deferred) { tl_deferred = d; }
afx::DeferScope::~DeferScope() { tl_deferred = (afx::Deferred*)prev; }

// Entered with c->mu held once.  The sessions this thread's deferred calls were left with: the one it leads is launched, the others
// are waited for (with the context released meanwhile: their leaders need it).  Returns the first failure.
int afx::drain_deferred(afx_ctx* c, afx::Deferred& d) {
  const std::thread::id me = std::this_thread::get_id();
  int rc = AFX_OK;
  std::string err;
  std::vector<std::shared_ptr<afx::Session>> pending;
  pending.swap(d.pending);
  for (const auto& S : pending) {
    int r;
    if (S->leader == me && S->state == afx::Session::COLLECTING) r = lead(c, S);
    else {
      const int depth = c->lock_depth;
      c->lock_depth = 0;
      c->mu.unlock();
      {
        std::unique_lock<std::mutex> lk(S->done_mu);
        S->done_cv.wait(lk, [&] { return S->done; });
        r = S->rc;
        if (r) set_error(S->err);
      }
      c->mu.lock();
      c->lock_depth = depth;
    }
    if (r && !rc) { rc = r; err = afx_last_error(); }
  }
  if (rc) set_error(err);
  return rc;
}

int host_pipe(afx_ctx* c, size_t count, const SliceFn& slice, const PlanKey& join_key) {
  if (c->session && !c->session->paused) {
    // collected: one slice, staged into the session's images; nothing is waited for here
    Stager st(c, c->session->lane, c->session);
    return slice(st, 0, count);
  }
  if (c->lock_depth == 1) {
    // the outermost call on this context: a small one joins the other threads' small calls, anything else needs the context to itself
    const afx_ctx::Coalesce& co = c->co;
    if (co.enabled && co.max_items && count && count <= co.max_call_items && c->small_batch_items && count <= c->small_batch_items && !c->trace &&
        !c->pipelining && !c->cur_stager)
      return coalesced_call(c, count, slice, join_key);
    // (a group of a mixed request too large to be collected: what the request's small groups left with the sessions goes first -
    // quiesce would otherwise wait for a session only this thread can launch)
    if (tl_deferred && !tl_deferred->pending.empty()) { const int rc = afx::drain_deferred(c, *tl_deferred); if (rc) return rc; }
    CtxLock::quiesce(c);
  }
  const int entry_force = c->force_lane;
  Stager* const entry_stager = c->cur_stager;
  // A short first slice, then whole passes: the first slice's copy is the one nothing overlaps, and slices the size of a pass
  // compute at the large-pass rate (statements.hpp host_slice_items; profiles/r06_host_slices.txt)
  const size_t per = host_slice_items(c), first = host_first_slice_items(c);
  std::unique_ptr<Stager> st[2];
  int rc = AFX_OK;
  size_t i = 0;
  try {
    for (size_t off = 0; off < count && !rc; i++) {
      const size_t n = std::min(i == 0 ? std::min(first, per) : per, count - off);
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

// A set of plans whose workspaces (or blobs) do not fit the device side by side runs in halves, one behind the other on the lane's
// stream - before every caller the session carried is sent home with an engine fault for ONE large neighbour's sake (run_plans
// makes room for its blob and workspace before it enqueues anything: afx::device_alloc_failed tells that failure from the others)
static int run_plans_fitting(afx_ctx* c, int lane, Plan** ps, size_t n) {
  int rc = run_plans(c, lane, ps, n);
  if (!rc || n <= 1 || !afx::device_alloc_failed()) return rc;
  (void)hipGetLastError();
  const size_t h = n / 2;
  if ((rc = run_plans_fitting(c, lane, ps, h))) return rc;
  return run_plans_fitting(c, lane, ps + h, n - h);
}

int afx::Session::launch() {
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
    rc = run_plans_fitting(c, lane, ps.data(), ps.size());
  }
  if (!rc && out_used && !outs.empty()) AFX_HIP(hipMemcpyAsync(L.pin, L.staging_out.p, out_used, hipMemcpyDeviceToHost, s));
  return rc;
}

int afx::Session::complete(int rc) {
  if (empty()) return rc;
  afx_ctx::Lane& L = c->lane[lane];
  (void)hipSetDevice(c->device);
  // whatever was enqueued before a failure is waited for too: the callers' arrays and the images are reused after this
  const hipError_t e = hipStreamSynchronize(L.stream);
  if (!rc && e != hipSuccess) { set_error(std::string("hipStreamSynchronize: ") + hipGetErrorString(e)); rc = AFX_E_HIP; }
  if (!rc)
    for (const Out& o : outs) memcpy(o.dst, (const uint8_t*)L.pin + o.pin_off, o.len);
  drop();
  return rc;
}
