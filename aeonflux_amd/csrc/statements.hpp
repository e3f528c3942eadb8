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
static constexpr size_t BLOB_CAP = size_t(4) << 20;   // initial size of a lane's plan-blob buffers (they grow: engine.cpp run_plans)
// Items per pass (afx_ctx_set_chunk_items).  Bounds the workspace: window tables + points in flight are ~25 KB (issue,
// n = 16) to ~70 KB (C3 verification) per item, so the default 2^19 needs 13-37 GB of the 288 GB.  Measured on 2^20-item
// batches: issue 6.21 M/s at 2^17, 6.45 at 2^18, 6.76 at 2^19, 6.84 at 2^20; C3 verification 2.42 / 2.45 / 2.45 / 2.44.
static constexpr uint32_t CHUNK_DEFAULT = 1u << 19;
// assembled plans kept for reuse (afx_ctx.plan_cache): entries / bytes
static constexpr size_t PLAN_CACHE_ENTRIES = 512, PLAN_CACHE_BYTES = size_t(64) << 20;
}
using namespace afx;
struct Stager;

// pins the calling thread to the CPUs of member `index`'s NUMA node for the life of the returned object (group.cpp; no-op where the
// topology is not exposed); `caller_thread`: restore the thread's mask afterwards
struct GroupPin { void* impl = nullptr; GroupPin(afx_group* g, uint32_t index, bool caller_thread); ~GroupPin(); GroupPin(const GroupPin&) = delete; };
// member 0's afx_ctx_set_small_batch_items, read under its lock (which calls of a group are "small": group.cpp run_members, mixed.cpp)
uint32_t afx_group_small_batch_items(afx_group* g);
// context construction shared with afx_issuer_keygen (issuer_params may be null there)
int afx_ctx_create_impl(afx_ctx** out, int device, const uint8_t* sp, size_t splen, const uint8_t* key, size_t klen,
                        const uint8_t* key_scalars_only, const uint8_t* issuer_params);

// ------------------------------------------------------------------------------------------------
// running a statement: assemble its plan (or take it from the cache), place it, launch
// ------------------------------------------------------------------------------------------------
using BuildFn = std::function<void(Assembler&, size_t /*chunk offset*/, uint32_t /*chunk count*/)>;

// The bytes that determine a plan apart from its addresses (statement, shape, mode flags - never the data pointers), as the cache
// key itself: no hash, so two requests share an entry only when these bytes are equal.  Callers pass a CANONICAL shape (unused
// array tails zeroed: canonical_shape), because the shape of afx_verify_presentations_wire comes from the caller's blob.
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
// mode flags that change a plan, for plan_key (every statement passes them all: a flag that does not matter to a statement only
// costs it a second cache entry)
inline uint64_t mode_flags(const afx_ctx* c) {
  // (bits 5-7: the chain width a collecting session asks of the latency plan; outside a session it follows from the count, which is in the key)
  return (c->strict ? 1u : 0u) | (c->fixed_key_schedule ? 2u : 0u) | ((uint64_t)(c->secret_mode & 3) << 3) | ((uint64_t)(c->merge_class & 7) << 5) |
         ((uint64_t)c->small_batch_items << 8) | ((uint64_t)(c->variants & 0x3f) << 40);   // (small_batch_items <= 2^16: bits 8-24)
}

namespace afx {
// internal return codes of the staging path (never leave the library)
static constexpr int AFX_RETRY_FULL = -1000;       // the collecting session cannot take this call: wait for the next one
static constexpr int AFX_RETRY_NOAPPEND = -1001;   // the call does not fit the item slots it was offered: stage it as a group of its own

// Several small host-pointer calls collected into ONE set of kernel launches (mixed.cpp: the shape groups of a mixed request;
// plans.cpp coalesced_call: calls of several host threads on one context).
// While a session is open on a context, every host-pointer front end stages its arrays into the session's image instead of
// sending them, every *_dev call leaves its plan with the session instead of launching it, and results are declared instead of
// fetched; flush() sends the image in one copy, runs all the plans merged launch by launch (engine.cpp run_plans), brings every
// result back in one copy and scatters them to the callers' arrays.  The calls must be independent of each other.
struct Session {
  afx_ctx* c;
  int lane = 0;
  bool paused = false;
  std::vector<std::unique_ptr<Plan>> plans;
  size_t in_used = 0, out_used = 0;                 // bump pointers into the lane's staging / staging_out buffers (and their pinned images)
  struct Out { uint8_t* dst; size_t pin_off, len; };
  std::vector<Out> outs;
  std::vector<std::function<int()>> pre;            // launches that run after the upload and before the plans (k_aos_to_soa of a serialized batch)
  // `shared`: a session of the context's coalescer (afx_ctx::co) - it is the context's `session` only while one call stages into it
  explicit Session(afx_ctx* ctx, bool shared_ = false) : c(ctx), shared(shared_) { if (!shared) c->session = this; }
  ~Session() { if (!shared && c->session == this) c->session = nullptr; }   // (a shared session's last owner may be a caller that no longer holds the context)
  Session(const Session&) = delete;
  Session& operator=(const Session&) = delete;
  bool empty() const { return plans.empty() && outs.empty() && pre.empty() && in_used == 0 && out_used == 0; }
  void drop() { plans.clear(); outs.clear(); pre.clear(); slots.clear(); in_used = out_used = 0; }
  int ensure_images(size_t in_bytes, size_t out_bytes);   // device staging + pinned images of at least these sizes (only while empty)
  // flush() = launch() + complete(): everything is enqueued on the lane's stream by launch() (the context's host state is used:
  // under afx_ctx::mu), complete() waits for the stream and scatters the results (no context state: the coalescer runs it unlocked)
  int launch();
  int complete(int launch_rc);
  int flush() { return complete(launch()); }

  // ---- item slots: same-shape calls of a shared session in ONE pass -------------------------------------------------------
  // A group created by the first call of a (statement, shape, mode) - the join key - is laid out for `dn` items (at least a
  // wave's 64) of which the call uses the first few; a later call of the same key copies its rows into the next free slots of
  // the SAME arrays and declares its results there: no second plan, no second set of rows in the launches.  `ops` is the
  // layout the creating call's Stager recorded; a joining call must reproduce it operation by operation (else it is staged as a
  // group of its own).
  struct Op { uint8_t kind; size_t rows, elem, dn, len, off; };   // kind: 0 constant bytes, 1 output bytes, 2 scratch, 3 input rows, 4 output rows
  struct Slots { PlanKey key; std::vector<Op> ops; size_t in_at, out_at, in_bytes, out_bytes; uint32_t dn, used; };
  std::vector<Slots> slots;
  // ---- coalescer state (under afx_ctx::mu) ---------------------------------------------------------------------------------
  bool shared = false;
  enum State { COLLECTING, LAUNCHING, INFLIGHT, DONE } state = COLLECTING;
  int rc = 0;                       // of the flush: every joined call returns it
  std::string err;
  // what the joined callers sleep on - the session's own, not the context's: a completion wakes the callers it answers and nobody
  // else, and they return without touching the context again (`done`, and rc / err before it, under `done_mu`)
  std::mutex done_mu;
  std::condition_variable done_cv;
  bool done = false;
  void finish(int rc_, std::string err_) {
    { std::lock_guard<std::mutex> g(done_mu); rc = rc_; err.swap(err_); done = true; }
    done_cv.notify_all();
  }
  std::thread::id leader;           // the caller that opened the session launches it ...
  bool leader_defers = false;       // ... unless it is a deferring one (a mixed request staging its groups one after the other, afx::DeferScope):
                                    // the first ordinary caller that joins takes the session over, so that nobody's single call waits
                                    // for a request's remaining groups to be assembled (plans.cpp coalesced_call)
  bool hurry = false, full = false; // launch now: an exclusive caller waits / the session has all it can take
  std::chrono::steady_clock::time_point deadline;
  uint32_t mclass = 0;              // afx_ctx::merge_class while calls stage into this session
  uint32_t calls = 0;
  uint64_t items = 0;
  std::map<std::string, uint32_t> key_items;   // items per join key (next session's slot counts)
};
}  // namespace afx

namespace afx {
// Calls of one thread that were LEFT with the coalescer's sessions instead of being waited for one by one (the small groups of a mixed
// request): while a DeferScope is alive on the thread, a collected call returns as soon as its rows are staged - its results exist once
// drain_deferred has returned.  plans.cpp.
struct Deferred {
  std::vector<std::shared_ptr<Session>> pending;
  void note(const std::shared_ptr<Session>& s) { for (const auto& p : pending) if (p == s) return; pending.push_back(s); }
  void forget(const std::shared_ptr<Session>& s) { for (size_t i = 0; i < pending.size(); i++) if (pending[i] == s) { pending.erase(pending.begin() + i); return; } }
};
struct DeferScope { void* prev; explicit DeferScope(Deferred* d); ~DeferScope(); DeferScope(const DeferScope&) = delete; DeferScope& operator=(const DeferScope&) = delete; };
int drain_deferred(afx_ctx* c, Deferred& d);   // with c->mu held once (a CtxLock of the joining kind)
}  // namespace afx

// Every entry point that touches a context's host state holds this.  `joiner`: a small host-pointer front end, which may join the
// coalescer's collecting session (host_pipe decides); everything else needs the context to itself and first waits until no session
// collects or is in flight (quiesce).  Re-entrant on the owning thread (the host-pointer front ends call the *_dev forms).
struct CtxLock {
  afx_ctx* c;
  CtxLock* outer_prev = nullptr;
  bool is_outermost = false;
  // the calling thread's outermost lock (per thread): a collected call that only has to wait for its session gives the context up
  // early through it (release) instead of taking the lock again just to drop it
  static CtxLock*& outermost() { static thread_local CtxLock* p = nullptr; return p; }
  explicit CtxLock(afx_ctx* ctx, bool joiner = false) : c(ctx) {
    if (!c) return;
    // (Trying the lock for a while before sleeping on it - the hold is a few microseconds, a futex hand-over several times that - was
    // measured and dropped: +13 % at 16 threads on 16 cores, -60 ... -90 % from 128 threads on, where the spinners take the cores the
    // holder needs: profiles/r05_ab_lock_spin.txt.)
    c->mu.lock();
    if (++c->lock_depth == 1) {
      is_outermost = true;
      outer_prev = outermost();
      outermost() = this;
      if (!joiner) quiesce(c);
    }
  }
  ~CtxLock() {
    if (is_outermost) outermost() = outer_prev;
    if (c) { --c->lock_depth; c->mu.unlock(); }
  }
  // drops the lock now; the destructor then has nothing to do (only the outermost lock of the thread, at depth 1)
  void release() { if (c) { --c->lock_depth; c->mu.unlock(); c = nullptr; } }
  CtxLock(const CtxLock&) = delete;
  CtxLock& operator=(const CtxLock&) = delete;
  // waits on the coalescer's condition with `mu` released (only ever at depth 1: the recursive mutex is released by ONE unlock)
  static void wait(afx_ctx* c) { const int d = c->lock_depth; c->lock_depth = 0; c->co.cv.wait(c->mu); c->lock_depth = d; }
  // (the bound is converted to the system clock: waits on it are pthread_cond_timedwait, which the thread sanitizer of this
  // toolchain understands; it only bounds how long a collection lingers)
  static void wait_until(afx_ctx* c, std::chrono::steady_clock::time_point t) {
    const auto left = t - std::chrono::steady_clock::now();
    const int d = c->lock_depth;
    c->lock_depth = 0;
    if (left > std::chrono::steady_clock::duration::zero()) c->co.cv.wait_until(c->mu, std::chrono::system_clock::now() + std::chrono::duration_cast<std::chrono::system_clock::duration>(left));
    c->lock_depth = d;
  }
  static void quiesce(afx_ctx* c) {
    afx_ctx::Coalesce& co = c->co;
    if (!co.open && !co.inflight) return;
    co.exclusive_waiters++;
    while (co.open || co.inflight) {
      if (co.open) co.open->hurry = true;
      co.cv.notify_all();
      wait(c);
    }
    co.exclusive_waiters--;
    co.cv.notify_all();
  }
};

// key non-empty and the call is a small host-pointer one (a Stager is staging it): the assembled plan is kept, position-independent,
// and reused by later calls of the same statement, shape, mode and (padded) size - those only copy it and move its pointers.
int run_chunked(afx_ctx* c, size_t count, const BuildFn& build, const PlanKey& key = PlanKey());

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

// Host-pointer front ends: the call's arrays staged into HBM on one of the context's two lanes.  Inputs (and kernel scratch) and
// outputs live in two regions (Lane::staging, Lane::staging_out), offsets of the second carry the OUT bit.  Inputs are copied on
// the lane's stream; results come back through the lane's pinned buffer (fetch_all / drain), so that a front end can keep one
// slice of a batch computing on one lane while it stages the next slice on the other (host_pipe).  Under a Session (several small
// calls in one set of launches) the same calls only fill the session's images and declare their results.
struct Stager {
  afx_ctx* c;
  int ln;
  int prev_force;
  Stager* prev_stager;
  afx::Session* ses;                 // non-null: this call is being collected
  size_t in_bytes = 0, out_bytes = 0, pin_bytes = 0;
  size_t in_at = 0, out_at = 0;      // session mode: where this call's regions start inside the lane's buffers
  bool uploaded = false;
  static constexpr size_t OUT = size_t(1) << 62;
  struct Copy { size_t off; const uint8_t* src; size_t len; bool constant; };
  struct Out { uint8_t* dst; size_t pin_off, len; };
  std::vector<Copy> copies, zeros;   // zeros: row tails of padded passes (only the unpacked upload needs them spelled out)
  std::vector<Out> outs;
  // ---- item slots of a shared session (afx::Session::Slots) ----
  using Op = afx::Session::Op;
  std::vector<Op> ops;               // this call's layout, operation by operation (what a later call of the same key must reproduce)
  afx::Session::Slots* app = nullptr;   // non-null: this call takes items [slot, slot + n) of that group's arrays
  size_t slot = 0, op_i = 0;
  bool mismatch = false;             // the call's operations are not the group's
  uint32_t slots_hint = 0;           // a new group in a shared session: lay it out for at least this many items
  uint32_t last_dn = 0;              // what dev_items() answered (0: the front end does not pad - its group cannot be joined)
  uint32_t layout_tag = 0;           // which front end staged the arrays (part of the plan cache key: 0 column arrays, 1 serialized records)
  // the *_dev calls made while this object lives run on its lane and know its staged ranges (run_chunked: plan reuse)
  // session: the Session collecting this call (host_pipe passes the context's), or null: the call stages and launches by itself
  explicit Stager(afx_ctx* ctx, int lane = 0, afx::Session* session = nullptr)
      : c(ctx), ln(session ? session->lane : lane), prev_force(ctx->force_lane), prev_stager(ctx->cur_stager), ses(session) {
    c->force_lane = ln;
    c->cur_stager = this;
  }
  ~Stager() { c->force_lane = prev_force; c->cur_stager = prev_stager; }
  Stager(const Stager&) = delete;
  Stager& operator=(const Stager&) = delete;
  hipStream_t stream() const { return c->lane[ln].stream; }
  // Items a pass of `n` host items runs with on the device.  Small calls are padded up to a power of two (at least 16): the
  // plan of a padded size serves every call of that statement and shape up to it (afx_ctx.plan_cache), the extra lanes work on
  // zeros and their results are never fetched - in the latency regime the device has lanes to spare.  Larger calls: n.
  // In a shared session: the item slots of the group the call joins, or (a new group) at least a wave's 64 and what the last
  // session carried of this key - the slots later callers fill.
  uint32_t dev_items(size_t n) {
    if (app) return last_dn = app->dn;
    if (!c->small_batch_items || n > c->small_batch_items || c->trace || n == 0) return last_dn = (uint32_t)n;
    uint32_t b = 16;
    // (the hint - what the last session carried of this key - rounded up to a power of two and capped: passes stay padded to powers
    // of two, one kept plan per size, and a lone call after a burst is not laid out for thousands of items)
    if (ses && ses->shared) { b = 64; while (b < slots_hint && b < 512) b <<= 1; }
    while (b < n) b <<= 1;
    return last_dn = (std::min<uint32_t>(b, c->small_batch_items) < n ? (uint32_t)n : std::min<uint32_t>(b, c->small_batch_items));
  }
  // the recorded operation a joining call is at: must equal what the call is doing now
  size_t joined(uint8_t kind, size_t rows, size_t elem, size_t dn, size_t len) {
    if (op_i >= app->ops.size()) { mismatch = true; return 0; }
    const Op& o = app->ops[op_i++];
    if (o.kind != kind || o.rows != rows || o.elem != elem || o.dn != dn || o.len != len) { mismatch = true; return 0; }
    return o.off;
  }
  // reserve `len` bytes: filled from host `src` (input region; the same bytes for every call of the join key: a cell map), or an
  // output area when src == nullptr (output region, zeroed)
  size_t add(const uint8_t* src, size_t len) {
    if (app) {
      const size_t off = joined(src ? 0 : 1, 0, 0, 0, len);
      if (src && !mismatch) copies.push_back({ off, src, len, true });
      return off;
    }
    if (!src) { const size_t off = (out_bytes + 255) & ~size_t(255); out_bytes = off + len; ops.push_back({ 1, 0, 0, 0, len, OUT | off }); return OUT | off; }
    const size_t off = (in_bytes + 255) & ~size_t(255);
    in_bytes = off + len;
    copies.push_back({ off, src, len, true });
    ops.push_back({ 0, 0, 0, 0, len, off });
    return off;
  }
  // scratch that the call's own kernels fill before anything reads it
  size_t reserve(size_t len) {
    if (app) return joined(2, 0, 0, 0, len);
    const size_t off = (in_bytes + 255) & ~size_t(255);
    in_bytes = off + len;
    ops.push_back({ 2, 0, 0, 0, len, off });
    return off;
  }
  // items [first, first + n) of a [rows][total][elem] host array -> a [rows][dn][elem] device array (dn >= n: dev_items; the
  // rows' tails stay zero); src == nullptr: an output array of that extent
  size_t add_rows(const uint8_t* src, size_t rows, size_t elem, size_t total, size_t first, size_t n, size_t dn = 0) {
    if (dn < n) dn = n;
    if (app) {
      const size_t off = joined(src ? 3 : 4, rows, elem, dn, 0);
      if (src && !mismatch)
        for (size_t r = 0; r < rows; r++) copies.push_back({ off + (r * dn + slot) * elem, src + (r * total + first) * elem, n * elem, false });
      return off;
    }
    if (!src) {
      const size_t off = (out_bytes + 255) & ~size_t(255);
      out_bytes = off + rows * dn * elem;
      ops.push_back({ 4, rows, elem, dn, 0, OUT | off });
      return OUT | off;
    }
    const size_t off = (in_bytes + 255) & ~size_t(255);
    in_bytes = off + rows * dn * elem;
    for (size_t r = 0; r < rows; r++) {
      copies.push_back({ off + r * dn * elem, src + (r * total + first) * elem, n * elem, false });
      if (dn > n) zeros.push_back({ off + (r * dn + n) * elem, nullptr, (dn - n) * elem, false });   // the padding lanes read zeros
    }
    ops.push_back({ 3, rows, elem, dn, 0, off });
    return off;
  }
  uint8_t* in_base() const { return (uint8_t*)c->lane[ln].staging.p + in_at; }
  uint8_t* out_base() const { return (uint8_t*)c->lane[ln].staging_out.p + out_at; }
  uint8_t* dev(size_t off) const { return (off & OUT) ? out_base() + (off & ~OUT) : in_base() + off; }
  // Calls of few items are many short rows (75 arrays for a C3 presentation batch): each row as its own copy from pageable
  // memory costs more than the kernels gain from the latency plan.  Up to PACK_LIMIT bytes the input region's image is put
  // together in a pinned buffer and sent in ONE transfer.  16 MB by measurement (tools/midsize_host_calls.py, C3 shape, ms per
  // call at a limit of 4 / 16 / 64 MB: 2^11 items 2.73 / 2.04 / 2.03, 2^12 4.19 / 3.51 / 3.54, 2^13 and beyond no gain).
  static constexpr size_t PACK_LIMIT = size_t(16) << 20;
  static size_t pack_limit();
  int upload();
  // results: the device array [rows][dn][elem] at `off` goes to items [first, first + n) of the host array [rows][total][elem].
  // plan_fetch() declares them; fetch_all() enqueues the copies into the pinned buffer after the kernels; drain() waits for the
  // lane and scatters them to the caller's arrays.
  void plan_fetch(uint8_t* dst, size_t off, size_t rows, size_t elem, size_t total, size_t first, size_t n, size_t dn = 0) {
    if (!dst || !rows || !n) return;
    if (dn < n) dn = n;
    const size_t o = off & ~OUT;   // outputs only
    for (size_t r = 0; r < rows; r++) outs.push_back({ dst + (r * total + first) * elem, o + (r * dn + (app ? slot : 0)) * elem, n * elem });
    if (!app) pend_.push_back({ o, rows * dn * elem });
  }
  int fetch_all();
  int drain();

 private:
  struct Pend { size_t off, len; };
  std::vector<Pend> pend_;
  bool whole_ = false;   // fetch_all copied the whole output region: pin offsets are output offsets
  bool fetched_ = false; // fetch_all ran: the pinned buffer holds (or will hold, once the lane is drained) this call's results
};

// Items per slice of a host-pointer call: slices alternate between the two lanes, so the host-to-device copy of one
// slice overlaps the kernels of the previous one.  The FIRST slice's copy overlaps nothing, so it is short (2^16 items: 159 MB of
// C3 rows, ~4 ms; its kernels - 23 ms - then cover the next slice's 25-50 ms of copy only in part, but the slice after computes
// meanwhile); every later slice is a whole pass (afx_ctx_set_chunk_items; 2^19), which computes at the large-pass rate.  Measured
// on C3, 2^20 presentations in pageable memory, against the device-resident rate of the same process (profiles/r06_host_slices.txt):
// slices of 2^17 (rounds 3-5) 96.2-96.8 %; 2^16 then 2^19: 97.9-98.4 %; 2^14, 2^16, 2^18, 2^19: 97.8-98.9 %.
static constexpr size_t HOST_FIRST_SLICE = size_t(1) << 16;
inline size_t host_slice_items(const afx_ctx* c) {
  if (c->trace) return ~size_t(0);   // the challenge trace is indexed by the item's position in ONE *_dev call
  return c->chunk_items ? c->chunk_items : CHUNK_DEFAULT;
}
inline size_t host_first_slice_items(const afx_ctx* c) { return c->trace ? ~size_t(0) : HOST_FIRST_SLICE; }
// Runs `slice(stager, first, n)` over [0, count) in slices on alternating lanes; `slice` stages, launches and calls
// fetch_all() on the Stager it is given; the pipe drains a lane before that lane is used again, and both at the end.
// Under a Session the call is one slice whose work is left with the session.
// `join_key` (optional): what makes two calls of this front end the same pass apart from their items - statement, shape, mode, which
// optional arrays are present.  With it, a small call may run in the coalescer's shared session (afx_ctx::co) and share a pass
// with other threads' calls of the same key; the front end must then stage per-item data with add_rows() / dev_items() only.
int host_pipe(afx_ctx* c, size_t count, const std::function<int(Stager&, size_t, size_t)>& slice, const PlanKey& join_key = PlanKey());
