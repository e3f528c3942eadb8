// Several devices behind one call: a group holds one context per device, each with its own copy of the
// parameters, key and generator tables.  A batch call splits [0, count) into contiguous ranges, one per member, runs
// the members' range calls (afx_*_range: staging slices on two streams, results through pinned buffers) on one host thread
// per member, and every member writes its part of the caller's arrays: no data moves between devices and there is no
// collective.  This is what one `&self` method of the reference (Issuer::verify, /root/reference/src/issuer.rs:141-147;
// Issuer::issue, :111-124) becomes when the issuer owns a node of GPUs.
#include <ctype.h>
#include <sched.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <atomic>
#include <memory>
#include <mutex>
#include <string>
#include <system_error>
#include <thread>
#include <vector>
#include "statements.hpp"

// Host threads follow their devices: a member's thread runs on the CPUs of the NUMA node its GPU hangs off (afx::cpus_of_device,
// plans.cpp), so that the pageable-to-pinned staging copies and the pinned buffers themselves (first touched by that thread) stay
// on the node the PCIe root of the device belongs to.  Eight members pulling ~6 GB/s each out of the caller's arrays through ONE
// node's memory controllers was the review's concern; nothing is pinned where the topology is not exposed (containers, the host
// simulation).
using afx::NodeCpus;
using afx::PinScope;
using afx::cpus_of_device;

struct afx_group {
  std::vector<afx_ctx*> members;
  std::vector<NodeCpus> node_cpus;   // per member: where its host thread runs
  std::atomic<uint32_t> next_small{ 0 };   // small calls go to one member each, in turn (run_members)
  ~afx_group() { for (afx_ctx* m : members) afx_ctx_destroy(m); }   // wipes every member's copy of the key
};

extern "C" int afx_group_create(afx_group** out, const int* devices, uint32_t n_devices, const uint8_t* sysparams, size_t sysparams_len,
                                const uint8_t* amacs_key, size_t amacs_key_len, const uint8_t issuer_params[64]) try {
  if (!out || !devices || n_devices == 0 || n_devices > 64) { set_error("bad device list"); return AFX_E_BAD_ARGS; }
  *out = nullptr;
  std::unique_ptr<afx_group> g(new afx_group());
  g->members.reserve(n_devices);
  for (uint32_t i = 0; i < n_devices; i++) {
    afx_ctx* c = nullptr;
    g->node_cpus.push_back(cpus_of_device(devices[i]));
    // (the context's pinned plan buffers are first touched here: on the device's node too)
    PinScope pin(g->node_cpus.back(), true);
    const int rc = afx_ctx_create(&c, devices[i], sysparams, sysparams_len, amacs_key, amacs_key_len, issuer_params);
    if (rc) {
      const std::string why = afx_last_error();
      set_error("member " + std::to_string(i) + " (device " + std::to_string(devices[i]) + "): " + why);
      return rc;
    }
    g->members.push_back(c);
  }
  *out = g.release();
  return AFX_OK;
} catch (...) { return afx::exception_rc(); }
GroupPin::GroupPin(afx_group* g, uint32_t index, bool caller_thread) { if (g && index < g->node_cpus.size()) impl = new PinScope(g->node_cpus[index], caller_thread); }
GroupPin::~GroupPin() { delete (PinScope*)impl; }
extern "C" void afx_group_destroy(afx_group* g) { delete g; }
extern "C" uint32_t afx_group_size(const afx_group* g) { return g ? (uint32_t)g->members.size() : 0; }
extern "C" afx_ctx* afx_group_member(afx_group* g, uint32_t i) { return (g && i < g->members.size()) ? g->members[i] : nullptr; }

// contiguous split of [0, count) over m members: the first count % m members take one item more
extern "C" void afx_shard_bounds(size_t count, uint32_t members, uint32_t index, size_t* first, size_t* n) {
  if (!first || !n) return;
  if (members == 0 || index >= members) { *first = 0; *n = 0; return; }
  const size_t base = count / members, extra = count % members;
  *first = (size_t)index * base + (index < extra ? index : extra);
  *n = base + (index < extra ? 1 : 0);
}

// a member's settings, read under its settings lock (afx_ctx_set_* on a member - afx_group_member hands the contexts out - writes them under it)
struct MemberSettings {
  uint32_t small_batch_items, chunk_items;
  bool strict, fixed_key_schedule, timing, trace;
  int secret_mode;
  bool same_as(const MemberSettings& o) const {
    return small_batch_items == o.small_batch_items && chunk_items == o.chunk_items && strict == o.strict && fixed_key_schedule == o.fixed_key_schedule &&
           timing == o.timing && secret_mode == o.secret_mode;
  }
};
static MemberSettings settings_of(afx_ctx* c) {
  std::lock_guard<std::mutex> lock(c->settings_mu);   // (not the context's own lock: a member may be busy with a long call)
  return { c->small_batch_items, c->chunk_items, c->strict, c->fixed_key_schedule, c->timing, c->trace != nullptr, c->secret_mode };
}
uint32_t afx_group_small_batch_items(afx_group* g) { return g && !g->members.empty() ? settings_of(g->members[0]).small_batch_items : 0; }

// one host thread per member; the first failure (lowest member index) is reported, with its message
template <class F>
static int run_members(afx_group* g, size_t count, F&& call) {
  const uint32_t m = (uint32_t)g->members.size();
  // A call small enough for the latency plan gains nothing from being cut into even smaller pieces (its duration is that of one
  // chain either way) and would pay a host thread per member: it goes to ONE member, the next in turn, so that small calls
  // arriving from several host threads spread over the group's devices (where each member collects the calls it is dealt:
  // afx_ctx_set_coalescing).
  const MemberSettings s0 = settings_of(g->members[0]);
  if (m > 1 && count != 0 && count <= s0.small_batch_items) {
    // ... in turn only while the members are interchangeable: settings are per member (afx_group_member), and a call must not
    // see strict mode, secret-independent addressing, kernel timing or a challenge trace on every m-th call only.  Members that
    // differ (a test set one of them up on purpose): member 0, whose threshold routed the call here, takes every small call.
    bool alike = !s0.trace;   // (a challenge trace is read back from ONE member's buffer)
    for (uint32_t k = 1; k < m && alike; k++) {
      const MemberSettings sk = settings_of(g->members[k]);
      alike = s0.same_as(sk) && !sk.trace;
    }
    const uint32_t i = alike ? g->next_small.fetch_add(1, std::memory_order_relaxed) % m : 0;
    PinScope pin(g->node_cpus[i], true);
    const int rc = call(g->members[i], (size_t)0, count);
    if (rc) { const std::string why = afx_last_error(); set_error("member " + std::to_string(i) + ": " + why); }
    return rc;
  }
  std::vector<int> rcs(m, AFX_OK);
  std::vector<std::string> errs(m);
  std::vector<std::thread> threads;
  auto body = [&](uint32_t i) {
    size_t first = 0, n = 0;
    afx_shard_bounds(count, m, i, &first, &n);
    if (n == 0) return;
    PinScope pin(g->node_cpus[i], i == 0);   // member 0 runs on the caller's thread, which gets its mask back
    rcs[i] = call(g->members[i], first, n);
    if (rcs[i]) errs[i] = afx_last_error();   // the error string is per thread
  };
  // a member whose thread cannot be started runs on this one; the members' calls themselves do not throw (C entry points)
  threads.reserve(m);
  for (uint32_t i = 1; i < m; i++) {
    try { threads.emplace_back(body, i); } catch (const std::system_error&) { body(i); }
  }
  body(0);
  for (std::thread& t : threads) t.join();
  for (uint32_t i = 0; i < m; i++)
    if (rcs[i]) { set_error("member " + std::to_string(i) + ": " + errs[i]); return rcs[i]; }
  return AFX_OK;
}

extern "C" int afx_group_verify_presentations(afx_group* g, const afx_shape* shape, const afx_presentation_soa* batch, size_t count, uint8_t* status) try {
  if (!g || g->members.empty() || !shape || !batch || !status) { set_error("null argument"); return AFX_E_BAD_ARGS; }
  return run_members(g, count, [&](afx_ctx* c, size_t first, size_t n) { return afx_verify_presentations_range(c, shape, batch, count, first, n, status); });
} catch (...) { return afx::exception_rc(); }
// a serialized batch over the group's devices: the records are contiguous, so every member takes a byte range of the caller's blob
extern "C" int afx_group_verify_presentations_wire(afx_group* g, const uint8_t* blob, size_t len, uint8_t* status, size_t status_cap, size_t* count_out) try {
  if (!g || g->members.empty() || !status || !count_out) { set_error("null argument"); return AFX_E_BAD_ARGS; }
  afx_shape sh;
  size_t count = 0, off = 0;
  int rc = afx_wire_parse(blob, len, &sh, &count, &off);
  if (rc) return rc;
  *count_out = count;
  if (status_cap < count) { set_error("status buffer too small"); return AFX_E_BAD_ARGS; }
  return run_members(g, count, [&](afx_ctx* c, size_t first, size_t n) {
    size_t seen = 0;
    return afx_verify_presentations_wire_range(c, blob, len, first, n, status, status_cap, &seen);
  });
} catch (...) { return afx::exception_rc(); }
extern "C" int afx_group_issue(afx_group* g, const afx_attributes_soa* requests, const afx_issue_randomness* rnd, size_t count,
                               const afx_issuance_soa* out, uint8_t* status) try {
  if (!g || g->members.empty() || !requests || !rnd || !out || !status) { set_error("null argument"); return AFX_E_BAD_ARGS; }
  return run_members(g, count, [&](afx_ctx* c, size_t first, size_t n) { return afx_issue_range(c, requests, rnd, count, first, n, out, status); });
} catch (...) { return afx::exception_rc(); }
extern "C" int afx_group_verify_issuances(afx_group* g, const afx_attributes_soa* attrs, const afx_issuance_soa* issuances, uint32_t n_responses,
                                          size_t count, uint8_t* status) try {
  if (!g || g->members.empty() || !attrs || !issuances || !status) { set_error("null argument"); return AFX_E_BAD_ARGS; }
  return run_members(g, count, [&](afx_ctx* c, size_t first, size_t n) { return afx_verify_issuances_range(c, attrs, issuances, n_responses, count, first, n, status); });
} catch (...) { return afx::exception_rc(); }
// the shape is a function of the credentials' layout alone: member 0 reports it (an empty range still does), the other
// members write theirs to a local
extern "C" int afx_group_show(afx_group* g, const afx_credentials_soa* creds, const afx_keypairs_soa* keypairs, const afx_show_randomness* rnd,
                              size_t count, const afx_presentation_out* out, afx_shape* shape_out, uint8_t* status) try {
  if (!g || g->members.empty() || !creds || !rnd || !out || !shape_out || !status) { set_error("null argument"); return AFX_E_BAD_ARGS; }
  int rc = afx_show_range(g->members[0], creds, keypairs, rnd, count, 0, 0, out, shape_out, status);
  if (rc) return rc;
  return run_members(g, count, [&](afx_ctx* c, size_t first, size_t n) {
    afx_shape local;
    return afx_show_range(c, creds, keypairs, rnd, count, first, n, out, &local, status);
  });
} catch (...) { return afx::exception_rc(); }
