// Host side of the engine: context (generators, tables, key on the device), workspace, the plan
// "assembler" that turns a statement into kernel launches, and the Schnorr constraint-system builder that
// mirrors zkp's toolbox API (allocate_scalar / allocate_point / constrain / verify_compact / prove_compact)
// so the statement code in statements.cpp reads like /root/reference/src/nizk/*.rs.
#pragma once
#include <hip/hip_runtime_api.h>
#include <sched.h>
#include <array>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <map>
#include <memory>
#include <set>
#include <mutex>
#include <string>
#include <thread>
#include <vector>
#include "../../include/aeonflux_gpu.h"
#include "kernels.h"
#include "plan.h"
#include "strobe_sim.hpp"

namespace afx {

void set_error(const std::string& s);
const char* last_error();
// No exception leaves the library: every int-returning entry point is a function-try-block whose handler calls this
// (inside a catch clause) to turn the exception in flight into a return code and an error string.
int exception_rc() noexcept;
#define AFX_HIP(call)                                                                                  \
  do {                                                                                                 \
    hipError_t e__ = (call);                                                                           \
    if (e__ != hipSuccess) {                                                                           \
      afx::set_error(std::string(#call) + ": " + hipGetErrorString(e__));                             \
      return AFX_E_HIP;                                                                                \
    }                                                                                                  \
  } while (0)

// true while the calling thread's most recent DevBuf::ensure failed because the device had no room (cleared by the next ensure):
// what tells "this set of plans does not fit" from any other HIP failure (plans.cpp run_plans_fitting)
bool device_alloc_failed();
struct DevBuf {
  void* p = nullptr;
  size_t cap = 0;
  bool sensitive = true;     // holds key-derived or prover-secret data at some point: zeroed before it is freed, also when it
                             // is only being replaced by a larger (or, after an out-of-memory back-off, smaller) buffer
  int ensure(size_t n);      // grow-only; returns AFX_OK / AFX_E_HIP
  void release(bool wipe);
};

using Enc = std::array<uint8_t, 32>;

}  // namespace afx

namespace afx { struct Plan; struct Session; }
struct Stager;

namespace afx {
// The CPUs of the NUMA node a device hangs off, intersected with what the process may use (plans.cpp; nothing where the topology
// is not exposed: containers, the host simulation).  /sys/bus/pci/devices/<bdf>/numa_node -> /sys/devices/system/node/node<N>/cpulist;
// AFX_SYSFS_ROOT (tests) stands in for "/".
struct NodeCpus {
  bool valid = false;
  cpu_set_t set;
};
NodeCpus cpus_of_device(int device);
// pins the calling thread for the scope's life (a thread the library started simply ends; a caller's own thread gets its mask back)
struct PinScope {
  bool restore = false;
  cpu_set_t saved;
  PinScope(const NodeCpus& n, bool caller_thread) {
    if (!n.valid) return;
    if (caller_thread) restore = sched_getaffinity(0, sizeof saved, &saved) == 0;
    (void)sched_setaffinity(0, sizeof n.set, &n.set);
  }
  ~PinScope() { if (restore) (void)sched_setaffinity(0, sizeof saved, &saved); }
  PinScope(const PinScope&) = delete;
  PinScope& operator=(const PinScope&) = delete;
};
// Host copies of a LARGE host-pointer call (plans.cpp Stager::upload / drain): the caller's pageable rows are gathered into the
// lane's pinned image by a few threads that run on the device's NUMA node, and the image goes to HBM in one transfer per
// contiguous run.  What this replaces: 75 hipMemcpyAsync calls per slice out of pageable memory, which the runtime carries out on
// the CALLING thread wherever the caller's scheduler put it - on the far socket of a two-socket host the slice's copy then takes
// longer than the previous slice's kernels and the device waits (the driver's round-5 box: 90.7 % of the device-resident rate
// against 96-97 % elsewhere).  One job at a time (large calls have the context to themselves); the caller's thread copies too.
struct CopyPool {
  struct Piece { uint8_t* dst; const uint8_t* src; size_t len; };
  struct Job {
    std::vector<Piece> pieces;
    std::atomic<size_t> next{ 0 }, left{ 0 };
  };
  explicit CopyPool(int device, uint32_t threads);   // threads - 1 workers (the caller is the first)
  ~CopyPool();
  CopyPool(const CopyPool&) = delete;
  CopyPool& operator=(const CopyPool&) = delete;
  void run(std::vector<Piece> pieces);               // returns when every piece has been copied
  uint32_t threads() const { return (uint32_t)workers_.size() + 1; }
  static void cut(std::vector<Piece>& out, uint8_t* dst, const uint8_t* src, size_t len);   // appends [src, src + len) in pieces of at most 1 MB
 private:
  void work(Job& j);
  void loop();
  NodeCpus node_;
  std::mutex mu_, run_mu_;
  std::condition_variable cv_work_, cv_done_;
  std::shared_ptr<Job> cur_;
  uint64_t gen_ = 0;
  bool stop_ = false;
  std::vector<std::thread> workers_;
};
}  // namespace afx

struct afx_ctx {
  std::recursive_mutex mu;   // the context's host state (workspace, staging, plan ring and cache) has one user at a time; see `co` below
  std::mutex settings_mu;    // the afx_ctx_set_* values are written under `mu` AND this; a group reads its members' settings under this alone (group.cpp)
  int lock_depth = 0;        // how many CtxLocks (statements.hpp) the thread that owns `mu` holds; 0 while it waits on co.cv
  // Concurrent small calls.  Issuer::verify takes `&self`, has no interior state and is called one presentation at a time from as
  // many threads as the server has (/root/reference/src/issuer.rs:141-147; Issuer::issue :111-124, AnonymousCredential::show
  // src/credential.rs:37-46 likewise).  A mutex around the whole call would give such a server the rate of ONE call however many
  // threads it has.  Instead a small host-pointer call that finds the context busy JOINS the session that is collecting (statements.hpp
  // afx::Session; plans.cpp coalesced_call): it holds `mu` only while it stages its rows - into free item slots of a same-shape
  // call's arrays when there are any, so that 64 callers of one shape become ONE pass of 64 items - and then sleeps on `cv` until
  // the flush that carries its rows has completed.  One leader per session (the caller that opened it) launches it: at once while
  // fewer than `max_inflight` sessions compute, otherwise when one of those completes, it fills up, or `max_wait_us` have passed;
  // the device wait itself runs without `mu`, so the next session collects (on another lane) while these compute.
  struct Coalesce {
    bool enabled = true;
    uint32_t max_wait_us = 2000;      // a collecting session waits at most this long for the launches ahead of it
    uint32_t max_items = 4096;        // items per session: a session that reaches them is launched whatever is in flight
    uint32_t max_call_items = 512;    // larger calls fill the device by themselves: they run alone
    std::condition_variable_any cv;   // waits on `mu`: session completed / lane free / exclusive caller done
    std::shared_ptr<afx::Session> open;   // the session that collects, or null
    int inflight = 0;                 // sessions launched and not yet completed
    bool lane_busy[5] = { false, false, false, false, false };   // a session (collecting or in flight) owns the lane's staging images (AFX_LANES entries)
    int max_inflight = 2;             // sessions computing at once (small passes leave most of the device idle); two measured best: profiles/r05_ab_sessions_in_flight.txt
    int exclusive_waiters = 0;        // callers that need the whole context (large batches, setters): no new session opens meanwhile
    std::map<std::string, uint32_t> demand;   // by join key: the items the last session carried (the item slots the next one starts with)
    uint32_t last_waves = 0, last_plans = 0;  // width of the last session launched: the merge class the next one assembles for
    uint64_t n_sessions = 0, n_calls = 0, n_items = 0, n_appended = 0, n_max_calls = 0, n_waited_flushes = 0;   // afx_ctx_get_coalescing_stats
    uint64_t staging_ns = 0, launch_ns = 0;   // how long `mu` was held staging calls / launching sessions
  } co;
  int device = 0;
  hipStream_t stream = nullptr;
  uint32_t n = 0, g = 0;
  bool has_key = false;
  // generator ids: index into the SystemParameters point list, then I, C_W, W
  uint32_t ngen = 0;
  uint32_t id_G() const { return 0; }
  uint32_t id_Gw() const { return 1; }
  uint32_t id_Gwp() const { return 2; }
  uint32_t id_Gx0() const { return 3; }
  uint32_t id_Gx1() const { return 4; }
  uint32_t id_Gy(uint32_t i) const { return 5 + i; }
  uint32_t id_Gm(uint32_t i) const { return 5 + g + i; }
  uint32_t id_GV() const { return 5 + g + n; }
  uint32_t id_Ga() const { return 6 + g + n; }
  uint32_t id_Ga0() const { return 7 + g + n; }
  uint32_t id_Ga1() const { return 8 + g + n; }
  uint32_t id_I() const { return 9 + g + n; }
  uint32_t id_CW() const { return 10 + g + n; }
  uint32_t id_W() const { return 11 + g + n; }
  std::vector<afx::Enc> gen_enc, gen_neg_enc;   // host copies of compress(G), compress(-G)
  // device residents
  afx::DevBuf d_gen_enc, d_pos_tables, d_gen_ext, d_key, d_consts;
  afx::DevBuf d_sec_tables;   // 6-bit positional tables of the generators (AFX_SEC_*) for secret scalars, built at context creation
  // key scalar slots in d_key ([k][32]): w, w', x0, x1, y[0..n), then constants one, zero
  const uint8_t* key_w() const { return (const uint8_t*)d_key.p; }
  const uint8_t* key_wp() const { return (const uint8_t*)d_key.p + 32; }
  const uint8_t* key_x0() const { return (const uint8_t*)d_key.p + 64; }
  const uint8_t* key_x1() const { return (const uint8_t*)d_key.p + 96; }
  const uint8_t* key_y(uint32_t i) const { return (const uint8_t*)d_key.p + 128 + 32 * i; }
  const uint8_t* const_one() const { return (const uint8_t*)d_consts.p; }
  unsigned long long* clock_probe() const { return (unsigned long long*)((uint8_t*)d_consts.p + 1024); }   // k_msm's clock probe (2 counters)
  std::vector<afx::Enc> host_key;               // w, w', x0, x1, y...; wiped on destroy
  const int32_t* gen_ext(uint32_t id) const { return (const int32_t*)d_gen_ext.p + (size_t)id * AFX_VAR_DWORDS; }
  // Execution lanes.  A lane = one HIP stream + its own workspace + a two-deep ring of (pinned host, device) plan
  // blobs (`blob_event[i]` marks the end of the copy that last used slot i, so a call can be assembled while the
  // previous one still runs).  Lane 0's stream is `stream`.  With pipelining off (default) every call runs on lane 0,
  // strictly ordered; with afx_ctx_set_pipelining(ctx, 1) successive *_dev calls alternate between the two lanes so
  // one call's launch tail and small kernels overlap the next call's work (the caller guarantees independence).
  struct Lane {
    hipStream_t stream = nullptr;
    afx::DevBuf ws;
    afx::DevBuf blob_dev[2];
    void* blob_host[2] = { nullptr, nullptr };
    size_t blob_host_cap[2] = { 0, 0 };
    hipEvent_t blob_event[2] = { nullptr, nullptr };
    int blob_next = 0;
    hipEvent_t msm_done = nullptr;   // end of this lane's latest k_msm launch
    bool msm_recorded = false;
    afx::DevBuf staging;             // host-pointer front ends: the call's inputs (and kernel scratch) in HBM (grow-only; zeroed on destroy)
    afx::DevBuf staging_out;         // ... and its outputs, a region of their own: one copy brings a small call's results back
    void* pin = nullptr;             // pinned bounce buffer for results (a device-to-host copy into pageable memory would
    size_t pin_cap = 0;              // block the host until the kernels end and serialise the two lanes)
    void* pin_in = nullptr;          // pinned image of a SMALL call's whole staging area: its many short input rows are gathered
    size_t pin_in_cap = 0;           // here on the host and go to HBM in one copy (statements.hpp Stager::upload); wiped on destroy
    hipEvent_t pin_in_done = nullptr;   // end of the copy that last read pin_in
  } lane[5];   // AFX_LANES: large calls alternate between lanes 0 and 1 (host_pipe, pipelining); the coalescer's sessions take whichever is free
  static constexpr int AFX_LANES = 5;
  bool pipelining = false;
  bool strict = false;   // afx_ctx_set_strict
  bool fixed_key_schedule = false;   // afx_ctx_set_fixed_key_schedule: no NAF for the issuer key's scalars
  // afx_ctx_set_secret_independent_addressing: where no memory address may depend on a secret scalar's digits.
  //   2 (default) the prover-side plans - issue, show, the symmetric-key helpers: what the crate's users get from dalek's constant-time
  //     arithmetic (src/amacs.rs:267-270, src/nizk/presentation.rs:162-184, zkp's Prover);  1: those and the issuer key's terms of
  //     Issuer::verify;  0: nowhere (the fastest tables; a device of the engine's own)
  int secret_mode = 2;
  bool secure_plan(bool prover_plan) const { return prover_plan ? secret_mode != 0 : secret_mode == 1; }
  uint32_t chunk_items = 0;   // afx_ctx_set_chunk_items; 0 = default
  uint32_t small_batch_items = 4096;   // afx_ctx_set_small_batch_items: passes of at most this many items take the latency plan
  // While a session collects the groups of a mixed request (mixed.cpp run_groups): how many 64-lane waves ONE grid row of the merged
  // launches will be (every collected group's items / 64, summed), as a class: 0 no session, k > 0 up to 12 << k waves.  The
  // latency plan cuts a stage's jobs into fewer, longer chains when one chain per term would queue on the device (Assembler::msm)
  uint32_t merge_class = 0;
  static uint32_t merge_class_of(uint64_t waves) { uint32_t k = 1; while (k < 7 && waves > (12ull << k)) k++; return k; }
  // waves per grid row of a pass of `count` items, alone or among the passes merged with it
  uint32_t row_waves(uint32_t count) const { return merge_class ? (9u << merge_class) : (count + 63) / 64; }   // (the class's middle)
  // grid rows of a walk kernel (k_compress2x, k_table_affine: one inversion per row) for `njobs` jobs of a small pass: a row per job
  // while those rows leave the device idle (a call waits for the longest walk: one inversion and one job instead of one and four),
  // 8 rows otherwise, 1 for large passes
  uint32_t walk_rows(uint32_t count, size_t njobs, bool small) const {
    if (!small) return 1u;
    return (uint64_t)row_waves(count) * njobs <= 2ull * 4 * n_cu ? (uint32_t)std::max<size_t>(njobs, 1) : 8u;
  }
  afx_plan_stats last_stats = {};   // per-item operation counts of the most recent plan
  std::map<std::string, std::array<uint64_t, 25>> folded_states;   // STROBE state after a transcript's all-constant leading blocks, by those blocks' bytes
                                                                   // (SchnorrBuilder::make_program); may derive from the key: wiped on destroy
  // assembled plans of small host-pointer calls, by (plan key bytes, padded pass size): position-independent (afx::Plan), placed by
  // relocation at every reuse (statements.hpp run_chunked).  May hold key material (prover plans): wiped on destroy.
  // Least recently used entries make room for new ones (shapes come from callers - a serialized batch names its own - so a stream of
  // unusual shapes must not pin the cache: the shapes a server really sees come back after a call or two).
  struct CachedPlan { std::shared_ptr<afx::Plan> plan; uint64_t last_use = 0; };
  std::map<std::pair<std::string, uint32_t>, CachedPlan> plan_cache;
  size_t plan_cache_bytes = 0;
  uint64_t plan_cache_tick = 0, plan_cache_hits = 0, plan_cache_misses = 0, plan_cache_evictions = 0;   // afx_ctx_get_plan_cache_stats
  uint32_t variants = 0;         // afx_ctx_set_plan_variants (AFX_VARIANT_*; tests): forced choices among equivalent plans / kernels; part of every plan key
  bool plan_selfcheck = false;   // AFX_VARIANT_SELFCHECK, or AFX_PLAN_SELFCHECK=1 at context creation (tests): every plan is assembled twice against different
                                 // provisional bases and both relocated copies must be byte-identical - a pointer field the relocation
                                 // does not know shows up as a difference
  afx::Session* session = nullptr;   // set while several small calls are being collected into one set of launches (statements.hpp)
  uint32_t n_cu = 256;   // compute units of the device (k_msm keeps 2 blocks resident on each)
  // parity aid (afx_ctx_set_challenge_trace): device array [trace_rows][trace_count][32] receiving every recomputed challenge
  uint8_t* trace = nullptr;
  size_t trace_rows = 0, trace_count = 0;
  afx::DevBuf trace_buf;
  unsigned lane_next = 0;
  int force_lane = -1;   // >= 0: every *_dev call runs on this lane (host-pointer front ends pick the lane they staged on)
  Stager* cur_stager = nullptr;   // the host-pointer front end whose *_dev call is running (its staged ranges: plan reuse), or null
  // afx_ctx_set_host_copy_threads: how many host threads gather a large host-pointer call's rows into the lane's pinned image
  // (afx::CopyPool, made at the first such call); 0: the runtime's own copies out of pageable memory on the caller's thread
  uint32_t host_copy_threads = 0;
  std::unique_ptr<afx::CopyPool> copy_pool;
  // optional per-launch HIP-event timing on `stream` (bench.py's roofline figure)
  bool timing = false;
  struct TimedLaunch { int kind; hipEvent_t start, stop; };
  std::vector<TimedLaunch> timed;          // recorded, not yet read back
  std::vector<hipEvent_t> event_pool;      // recycled events
  double kind_ms[24] = { 0 };
  uint64_t kind_launches[24] = { 0 };
};

namespace afx {

// L_MSM_WINDOW keeps the slot the single k_msm kernel had (timing names: statements.cpp KIND_NAMES)
enum LaunchKind { L_FILL_BAD, L_DECODE, L_SCCHECK, L_POINTOP, L_SCALAROP, L_MSM_WINDOW, L_HASH, L_FROM_UNIFORM, L_REDUCE_WIDE, L_COPY, L_FINISH,
                  L_MSM_FIXED, L_MSM_NAF, L_MSM_TABLES, L_COMPRESS, L_POINTSUM, L_NEGENC, L_TABLE_AFFINE, L_POWERS, L_KINDS };
// the kernels whose grid rows WALK a range of the launch's jobs (afx_walk_row) instead of taking one job each (afx_row)
inline bool walks(LaunchKind k) { return k == L_COMPRESS || k == L_NEGENC || k == L_TABLE_AFFINE; }

struct Launch {
  LaunchKind kind;
  size_t jobs_off = 0;      // offset of the job array in the plan's blob
  uint32_t njobs = 0;
  size_t rows_off = 0;      // L_COMPRESS, L_NEGENC, L_TABLE_AFFINE (walks()): the afx_walk_row array (grid rows; each walks a range of the jobs)
  uint32_t nrows = 0;
  // L_COPY: direct arguments (device-to-device)
  const uint8_t* in = nullptr;
  uint8_t* out = nullptr;
  size_t bytes = 0;
  int odd = 0;              // L_MSM_TABLES: kind of table (plan.h afx_table_job): 0 multiples 1..8, 1 odd multiples (NAF terms), 2 narrow;
                            // L_POINTSUM: 1 = some job sums sixteen parts or more (kernels.hip afxk_pointsum);
                            // L_DECODE: 1 = the launch holds Elligator jobs (afx_decode_job.elligator)
  int encodes = 1;          // L_MSM_*: some job of the launch encodes its result inside the kernel (kernels.hip k_msm<KIND, ENC, SEC>)
  int secret = 0;           // L_MSM_*: some term of the launch has a secret scalar under secret-independent addressing
};

// What an Assembler leaves behind: one statement over one pass of `count` items as a POSITION-INDEPENDENT unit.  Every device
// pointer inside - in the job arrays, in their side tables, in the pass - points into one of four ranges: the plan's own blob,
// its workspace, and the staged inputs / outputs of the host-pointer call it serves (anything else is context-resident: generator
// tables, the key, or the caller's own device arrays).  The plan is assembled against PROVISIONAL bases (non-canonical addresses
// that are never dereferenced) and moved to real ones by relocate(), which knows every pointer field by type (engine.cpp).  That
// is what lets a plan be (1) assembled once and reused by later calls of the same statement, shape and size - the host side of a
// small call was 0.3 ms of its 1.2 ms - and (2) placed next to the plans of OTHER small calls in one blob, workspace and set of
// kernel launches (run_plans): presentations of 64 shapes in one Issuer::verify stream cost a dozen launches, not 64 dozens.
struct Plan {
  std::vector<uint8_t> blob;
  std::vector<Launch> launches;
  std::vector<std::pair<size_t, uint32_t>> ptr_tables, term_tables;   // side tables that hold device pointers: (blob offset, entries)
  uint8_t* blob_base = nullptr;                       // what blob[0]'s device address is assumed to be
  uint8_t* ws_base = nullptr;  size_t ws_bytes = 0;   // workspace: failure words, variables, encodings, window tables, recoded scalars
  uint8_t* in_base = nullptr;  size_t in_bytes = 0;   // staged inputs of the call (Stager), or none
  uint8_t* out_base = nullptr; size_t out_bytes = 0;  // staged outputs
  size_t pass_off = 0;                                // the plan's afx_pass inside blob
  uint32_t count = 0;
  bool small = false;                                 // a latency plan: its hash launches may run k_hash_coop
  afx_plan_stats stats = {};
  Plan() = default;
  Plan(const Plan&) = default;
  ~Plan();                                            // plans can hold key material (NAF digits, witnesses in transcript constants): wiped
  void relocate(uint8_t* new_blob, uint8_t* new_ws, uint8_t* new_in, uint8_t* new_out);
  bool same_as(const Plan& o, std::string* why) const;   // byte comparison after relocation (plan_selfcheck)
};
// Launches the plans' kernels on the lane's stream (asynchronous): ONE plan runs its launches as they are; several are merged
// launch by launch (same kernel = one launch over all their rows, plan.h afx_pass).  Places the plans in the lane's blob and
// workspace (growing them if need be) and relocates them there.
int run_plans(afx_ctx* c, int lane, Plan* const* plans, size_t n);

// Builds one call's kernel launch list over a pass of `count` items.
class Assembler {
 public:
  // variant: which set of provisional bases the plan is assembled against (plan_selfcheck assembles every plan under two)
  Assembler(afx_ctx* ctx, uint32_t count, int variant = 0);
  ~Assembler();   // the plan can hold key material (prover witnesses in transcript constants, NAF digits of the key): wiped
  Assembler(const Assembler&) = delete;
  Assembler& operator=(const Assembler&) = delete;
  afx_ctx* ctx;
  uint32_t count;
  bool fail_all = false;      // statement-level failure for every item (reference would panic / reject all)
  bool secret_scalars = false;   // a prover-side plan (issue, show, the symmetric-key helpers): every scalar of its multiscalar jobs
                                 // but the constant 1 is a secret (blindings, witnesses, the key, nonces) - Assembler::msm marks the
                                 // terms when the context runs with secret-independent addressing
  std::string plan_error;     // a request the plan cannot serve (reported as AFX_E_BAD_ARGS, nothing is launched)
  afx_plan_stats stats = {};  // per-item operation counts of this plan

  // workspace (addresses relative to the plan's provisional workspace base)
  int32_t* new_var();         // extended point, SoA [36][count]
  uint8_t* new_enc();         // [count][32]
  uint8_t* new_wide();        // [count][64]
  uint64_t* new_state();      // [25][count]
  size_t blob_bytes() const { return blob_.size(); }

  // launches, executed in the order added
  void decode(const std::vector<afx_decode_job>& jobs);
  void sccheck(const std::vector<afx_sccheck_job>& jobs);
  void pointop(const std::vector<afx_pointop_job>& jobs);
  void negenc(const std::vector<afx_negenc_job>& jobs);   // encodings of the negations of decoded points, one inversion per item
  void scalarop(const std::vector<afx_scalarop_job>& jobs);
  void msm(std::vector<afx_msm_job> jobs);   // assigns digit/table slots; small batches: one chain per term (msm_split)
  // one more point for the NEXT msm() call's k_compress2x launch: out_enc = encoding of +-2 * var, where var is (or will be, by
  // that call's chains) the half some job left (afx_msm_job.leave_half) - the "-E1" of a proof of encryption being created
  void compress_also(const int32_t* var, uint8_t* out_enc, bool negate, uint32_t reject_identity);
  void hash(const std::vector<afx_hash_program>& progs);
  void from_uniform(const uint8_t* wide, uint8_t* out_enc, int32_t* out_var);
  void reduce_wide(const uint8_t* wide, uint8_t* out);
  void copy(uint8_t* dst, const uint8_t* src, size_t bytes);   // device-to-device
  void finish(uint8_t* status_dev, uint8_t fail_code);

  // blob: plan data copied to the device in one transfer; returns the (provisional) DEVICE address of the copy
  template <class T>
  const T* put(const T* src, size_t n) {
    const size_t off = blob_alloc(sizeof(T) * n, alignof(T) < 8 ? 8 : alignof(T));
    memcpy(blob_.data() + off, src, sizeof(T) * n);
    return reinterpret_cast<const T*>(blob_base_ + off);
  }
  size_t blob_alloc(size_t bytes, size_t align);
  // a table of device pointers: listed, so that Plan::relocate moves its entries
  template <class T>
  T* const* put_ptrs(T* const* src, size_t n) {
    T* const* dev = put(src, n);
    if (n) ptr_tables_.push_back({ (size_t)((const uint8_t*)dev - blob_base_), (uint32_t)n });
    return dev;
  }

  uint32_t* bad() const { return bad_; }
  // this pass takes the latency plan (afx_ctx_set_small_batch_items): one chain per term, statements avoid chains that wait for chains
  bool small() const { return ctx->small_batch_items != 0 && count <= ctx->small_batch_items; }
  // The plan, relocatable (window-table and digit workspace added, the pass written).  in/out: the staged ranges of the host-pointer
  // call this plan serves (its pointers into them are moved when the plan is reused under another staging layout), or null.
  int finish_plan(Plan& out, uint8_t* in_base, size_t in_bytes, uint8_t* out_base, size_t out_bytes);

  size_t max_digit_slots = 0, max_table_slots = 0;
  std::vector<Launch> launches;

 private:
  uint8_t* ws_alloc(size_t bytes);
  void msm_list(std::vector<afx_msm_job> jobs, bool no_naf, std::vector<afx_compress_job>& cjobs);
  void msm_split(std::vector<afx_msm_job> jobs, std::vector<afx_compress_job>& cjobs, bool no_naf, uint32_t var_per_part = 1, bool segments = false);
 public:
  uint32_t segments() const;                                         // 1: off
  bool secure() const;                                               // this pass runs the secret-independent plan
 private:
  // SEGMENTS (small prover passes under secret-independent addressing; engine.cpp Assembler::msm): a secret scalar on a per-item base P
  // runs as `segments()` short chains, segment k over 2^(k * bits/segments) * P, instead of one chain over all its windows.  The
  // points come from ONE k_powers chain per base, made the first time a stage multiplies by the base, and their two-entry tables
  // are built once and kept for the pass: a later stage (the proof's commitments, on the same bases) waits for a quarter of a chain.
  bool segment_bases(const std::vector<afx_msm_job>& jobs);          // makes the powers of the jobs' new bases; false: nothing to segment
  std::map<const int32_t*, std::vector<int32_t*>> powers_;           // base -> its powers 1 .. segments() - 1
  std::map<const int32_t*, uint32_t> kept_tables_;                   // narrow table slots that stay for the whole pass (segmenting passes: all of them)
  bool segmenting_ = false;                                          // this pass keeps its narrow tables (set by the first segmented stage)
  void compress(const std::vector<afx_compress_job>& cjobs, uint32_t groups);
  template <class T>
  void add_jobs(LaunchKind k, const std::vector<T>& jobs);
  void flush_encodings();
  void flush_maps();
  std::vector<afx_decode_job> pending_maps_;         // Elligator maps queued by from_uniform (small passes): they ride in the next decode launch
  std::vector<afx_pointop_job> pending_map_sums_;    // ... and the additions of the pairs
  std::vector<afx_reduce_job> pending_reductions_;   // blindings a small pass's transcripts squeeze out as 64 bytes: reduced by one launch behind the hash
  void add_walk_rows(Launch& l, uint32_t per_row);   // the afx_walk_row array of a k_compress2x / k_negenc / k_table_affine launch (+ its prefix scratch)
  std::set<const int32_t*> half_bases_;            // variables that hold HALF their point (producers with leave_half)
  std::vector<afx_compress_job> pending_cjobs_;    // compress_also()
  std::vector<const int32_t*> pending_half_vars_;  // ... and the variables they read: checked to hold halves when the queue is consumed
  std::vector<std::pair<size_t, uint32_t>> ptr_tables_, term_tables_;   // (blob offset, entries) of the pointer tables / afx_msm_term tables written so far
  std::vector<uint8_t> blob_;
  uint8_t* blob_base_ = nullptr;   // provisional device address of blob_[0]
  uint8_t* ws_base_ = nullptr;     // provisional device address of the workspace
  size_t ws_off_ = 0;
  uint32_t* bad_ = nullptr;
  friend class SchnorrBuilder;
};

// a compiled transcript (its all-constant leading blocks folded into the initial state, once per distinct prefix and context) as a
// device hash program over the given [count][32] field arrays; the caller sets outs / challenge / trace
afx_hash_program make_hash_program(Assembler& as, const StrobeSim& sim, const std::vector<const uint8_t*>& fields);

// A point variable of a constraint system: a batch constant (generator id, maybe negated) or a per-item
// variable (extended coordinates in the workspace + the [count][32] array holding its encoding).
struct PointVar {
  bool is_const = true;
  uint32_t gen = 0;
  bool neg = false;
  const int32_t* var = nullptr;
  bool var_negated = false;   // `var` holds the NEGATION of this point (its own encoding is still enc_dev): a term on it is the
                              // negated term on `var`, and shares var's window table (-E1 beside E1 in a proof of encryption)
  const uint8_t* enc_dev = nullptr;
  // optional fixed-base form of a variable point: P = alt_scalar[item] * G_alt_gen (e.g. M_i = m_i * G_m_i,
  // src/amacs.rs:234-235).  A term s*P then runs as the fixed-base term (s*alt_scalar)*G, one scalar product away.
  bool has_alt = false;
  uint32_t alt_gen = 0;
  const uint8_t* alt_scalar = nullptr;
  // optional: the point as a sum of parts, P = sum +-coef_k * base_k (coef == nullptr: coefficient 1).  With it, a verifier's term
  // -c * P can run as the independent terms -+(c * coef_k) * base_k, none of which waits for P itself to be computed (small
  // passes: Z of Issuer::verify and the constraint Z = z*I, statements.cpp)
  struct Part { const uint8_t* coef; uint32_t coef_stride; bool neg; const int32_t* var; int32_t fixed; };
  std::vector<Part> parts;
  static PointVar Const(uint32_t gen, bool neg = false) { PointVar p; p.is_const = true; p.gen = gen; p.neg = neg; return p; }
  static PointVar Var(const int32_t* var, const uint8_t* enc) { PointVar p; p.is_const = false; p.var = var; p.enc_dev = enc; return p; }
  static PointVar NegOf(const int32_t* var, const uint8_t* enc_of_negation) { PointVar p = Var(var, enc_of_negation); p.var_negated = true; return p; }
};
// A scalar variable: verifier side = the response array; prover side = the witness (per item, or a batch
// constant such as the issuer key when stride == 0; `host` holds its bytes then).
struct ScalarVar {
  const uint8_t* dev = nullptr;
  uint32_t stride = 32;
  Enc host{};   // only for stride == 0 (needed as constant bytes in the prover's rng rekeying)
};

// zkp::toolbox::{prover::Prover, verifier::Verifier} + SchnorrCS, batch form.
class SchnorrBuilder {
 public:
  SchnorrBuilder(Assembler& as, const char* transcript_label, const char* proof_label);
  int allocate_scalar(const char* label, const ScalarVar& v);
  int allocate_point(const char* label, const PointVar& p);
  void constrain(int lhs, const std::vector<std::pair<int, int>>& terms);
  // Verifier::verify_compact over the batch: commitments, transcript, challenge comparison
  // `pre_ops` receives the scalar products that fixed-base forms of variable points need (run them before msm_out)
  // trace_row: which row of the context's challenge trace (afx_ctx_set_challenge_trace) this proof reports to
  // `expand_ops` (optional): left-hand sides that carry `parts` have their -c * LHS term expanded over them; the products
  // c * coef_k go here (run them before msm_out, after whatever computes the coefficients)
  void verify_compact(const uint8_t* challenge_dev, uint32_t trace_row, size_t total, size_t off, std::vector<afx_msm_job>& msm_out, std::vector<afx_hash_program>& hash_out,
                      std::vector<afx_scalarop_job>* pre_ops = nullptr, std::vector<afx_scalarop_job>* expand_ops = nullptr);
  // Prover::prove_compact over the batch.  Fills: rng hash program (blindings), commitment msm jobs,
  // challenge hash program, response scalar ops.  rng_seed_dev: [count][32].
  void prove_compact(const uint8_t* rng_seed_dev, uint8_t* challenge_out, uint8_t* responses_out /* [nsc][count][32] */,
                     size_t response_row_stride, std::vector<afx_hash_program>& rng_hash, std::vector<afx_msm_job>& msm_out,
                     std::vector<afx_hash_program>& chal_hash, std::vector<afx_scalarop_job>& resp_ops,
                     std::vector<afx_scalarop_job>* pre_ops = nullptr);
  size_t num_scalars() const { return scalars_.size(); }

 private:
  int field_of(const uint8_t* dev);
  afx_msm_term term_for(const uint8_t* scalar, uint32_t stride, const PointVar& p, bool negate, std::vector<afx_scalarop_job>* pre_ops);
  afx_hash_program make_program(const StrobeSim& sim);
  Assembler& as_;
  StrobeSim sim_;
  std::vector<ScalarVar> scalars_;
  std::vector<PointVar> points_;
  std::vector<std::string> point_labels_;
  std::vector<std::pair<int, std::vector<std::pair<int, int>>>> constraints_;
  std::vector<const uint8_t*> fields_;
};

}  // namespace afx
