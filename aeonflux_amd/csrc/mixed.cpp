// Presentations of different shapes behind one call (include/aeonflux_gpu.h, "presentations of DIFFERENT shapes").
// Issuer::verify (/root/reference/src/issuer.rs:141-147) accepts any presentation; its shape is per presentation
// (src/nizk/presentation.rs:118-127, :293-309).  The GPU batch is shape-uniform, so a mixed stream is grouped here, on the
// host, by the used part of the shape; each group is one ordinary batch call of this library and the status bytes go back to
// the caller's order.  Only bytes move on the host (record copies when a shape's sections are not adjacent).
#include <map>
#include <memory>
#include <string>
#include <system_error>
#include <thread>
#include <vector>
#include "statements.hpp"

namespace {

std::string shape_key(const afx_shape& sh) {
  const afx_shape c = canonical_shape(sh);
  return std::string((const char*)&c, sizeof c);
}
uint32_t rd32(const uint8_t* b) { return (uint32_t)b[0] | ((uint32_t)b[1] << 8) | ((uint32_t)b[2] << 16) | ((uint32_t)b[3] << 24); }

// does a request of several groups on this context leave its small groups with the collector's sessions (plans.cpp)?
bool joins_the_collector(afx_ctx* ctx) {
  CtxLock probe(ctx, true);
  return ctx->lock_depth == 1 && ctx->co.enabled && ctx->co.max_items && ctx->small_batch_items && !ctx->trace && !ctx->pipelining && !ctx->session;
}

// positions given: each < status_len and used once over all groups; not given: contiguous after the groups before
template <class G>
int check_positions(const G* groups, size_t n_groups, size_t status_len) {
  std::vector<uint8_t> used(status_len, 0);
  size_t next = 0;
  for (size_t g = 0; g < n_groups; g++) {
    const G& grp = groups[g];
    for (size_t i = 0; i < grp.count; i++) {
      const uint64_t p = grp.positions ? grp.positions[i] : (uint64_t)(next + i);
      if (p >= status_len) { set_error("group " + std::to_string(g) + ": position outside the status array"); return AFX_E_BAD_ARGS; }
      if (used[p]) { set_error("group " + std::to_string(g) + ": a status position is used twice"); return AFX_E_BAD_ARGS; }
      used[p] = 1;
    }
    next += grp.count;
  }
  return AFX_OK;
}

// One ordinary batch call per group (`run_one(group, its status bytes)`), status bytes back to the caller's order.  With a
// context (`ctx` non-null) the groups small enough for the latency plan are COLLECTED (afx::Session): their arrays go to the
// device in one copy, their plans run merged - one launch per kernel over all the groups' rows - and their results come back in
// one copy, so a request of many layouts costs about one small call instead of one per layout.  A large group runs by itself, in
// between (the session is flushed first: results of one item never wait on another request's batch longer than they must).
template <class G, class RunOne>
int run_groups(afx_ctx* ctx, G* groups, size_t n_groups, uint8_t* status, size_t status_len, RunOne&& run_one) {
  if ((!groups && n_groups) || (!status && status_len)) { set_error("null argument"); return AFX_E_BAD_ARGS; }
  int rc = check_positions(groups, n_groups, status_len);
  if (rc) return rc;
  // With the context's collector on (afx_ctx_set_coalescing, the default) the request's small groups JOIN the session that is
  // collecting - other threads' calls and requests share its launches - instead of taking the context for a session of the request's
  // own: every group's call returns once its rows are staged (afx::DeferScope) and the request waits for all of them at the end.
  if (ctx && n_groups > 1 && joins_the_collector(ctx)) {
    afx::Deferred deferred;
    std::vector<std::vector<uint8_t>> tmpj(n_groups);
    size_t nextj = 0;
    {
      afx::DeferScope scope(&deferred);
      // (an exception must not pass the drain below: other threads' calls may sit in a session only this thread launches)
      try {
        for (size_t g = 0; g < n_groups && !rc; g++) {
          G& grp = groups[g];
          if (grp.count) {
            uint8_t* st = status + nextj;
            if (grp.positions) { tmpj[g].assign(grp.count, AFX_ST_VERIFICATION_FAILURE); st = tmpj[g].data(); }
            rc = run_one(grp, st);
            if (rc) set_error("group " + std::to_string(g) + ": " + afx_last_error());
          }
          nextj += grp.count;
        }
      } catch (...) { rc = afx::exception_rc(); }
      // what the groups left with the sessions: launched (the session this thread leads) or waited for - also after a failure, since
      // the staged rows point into this request's arrays
      CtxLock lk(ctx, true);
      const int rc2 = afx::drain_deferred(ctx, deferred);
      if (!rc) rc = rc2;
    }
    if (rc) return rc;
    for (size_t g = 0; g < n_groups; g++)
      if (groups[g].positions)
        for (size_t i = 0; i < groups[g].count; i++) status[groups[g].positions[i]] = tmpj[g][i];
    return AFX_OK;
  }
  std::unique_ptr<CtxLock> lock;
  std::unique_ptr<afx::Session> ses;
  if (ctx && n_groups > 1) lock.reset(new CtxLock(ctx));   // the session owns the context until its last flush
  if (lock && ctx->small_batch_items && !ctx->trace && !ctx->session) {
    ses.reset(new afx::Session(ctx));
    if ((rc = ses->ensure_images(0, 0))) return rc;
    // how wide the merged launches will be, in 64-lane waves per grid row: the latency plans of the collected groups cut a stage's
    // jobs into fewer, longer chains when one chain per term would queue on the device (Assembler::msm, afx_ctx::merge_class)
    uint64_t width = 0;
    for (size_t g = 0; g < n_groups; g++)
      if (groups[g].count && groups[g].count <= ctx->small_batch_items) width += (groups[g].count + 63) / 64;
    ctx->merge_class = afx_ctx::merge_class_of(width);
  }
  struct WidthReset { afx_ctx* c; ~WidthReset() { if (c) c->merge_class = 0; } } width_reset = { ses ? ctx : nullptr };
  // statuses of groups with positions land in a buffer of the group's own first (the session fills it at its flush)
  std::vector<std::vector<uint8_t>> tmp(n_groups);
  size_t next = 0;
  for (size_t g = 0; g < n_groups && !rc; g++) {
    G& grp = groups[g];
    if (grp.count) {
      uint8_t* st = status + next;
      if (grp.positions) { tmp[g].assign(grp.count, AFX_ST_VERIFICATION_FAILURE); st = tmp[g].data(); }
      const bool collect = ses && grp.count <= ctx->small_batch_items;
      if (ses && !collect) {
        if ((rc = ses->flush())) break;
        ses->paused = true;
      }
      rc = run_one(grp, st);
      if (ses) ses->paused = false;
      if (rc) set_error("group " + std::to_string(g) + ": " + afx_last_error());
    }
    next += grp.count;
  }
  if (ses) {
    if (rc) ses->drop();
    else rc = ses->flush();
    ses.reset();
  }
  if (rc) return rc;
  for (size_t g = 0; g < n_groups; g++)
    if (groups[g].positions)
      for (size_t i = 0; i < groups[g].count; i++) status[groups[g].positions[i]] = tmp[g][i];
  return AFX_OK;
}

// The same request on a group of devices: the groups small enough to be collected are dealt out to the members in turn and every
// member runs its share as ONE collected request on a host thread of its own (`mixed_on_ctx`: the single-context entry point);
// a large group is split over all the members by the group's ordinary batch call (`run_one`), before the small ones start.
// Statuses land where they belong because every share carries explicit positions.
template <class G, class MixedOnCtx, class RunOne>
int run_groups_on_devices(afx_group* group, G* groups, size_t n_groups, uint8_t* status, size_t status_len, MixedOnCtx&& mixed_on_ctx, RunOne&& run_one) {
  if ((!groups && n_groups) || (!status && status_len)) { set_error("null argument"); return AFX_E_BAD_ARGS; }
  int rc = check_positions(groups, n_groups, status_len);
  if (rc) return rc;
  const uint32_t m = afx_group_size(group);
  if (m == 0) { set_error("empty group"); return AFX_E_BAD_ARGS; }
  const uint32_t small = afx_group_small_batch_items(group);   // (the rule afx_group_* calls route by: group.cpp run_members)
  std::vector<std::vector<G>> share(m);
  std::vector<std::vector<uint64_t>> made;   // positions of groups that came without: contiguous after the groups before them
  made.reserve(n_groups);
  std::vector<uint8_t> tmp;
  size_t next = 0;
  uint32_t turn = 0;
  for (size_t g = 0; g < n_groups; g++) {
    G grp = groups[g];
    if (grp.count) {
      if (small && grp.count <= small) {
        if (!grp.positions) {
          made.emplace_back(grp.count);
          for (size_t i = 0; i < grp.count; i++) made.back()[i] = next + i;
          grp.positions = made.back().data();
        }
        share[turn++ % m].push_back(grp);
      } else if (!grp.positions) {
        if ((rc = run_one(groups[g], status + next))) { set_error("group " + std::to_string(g) + ": " + afx_last_error()); return rc; }
      } else {
        tmp.assign(grp.count, AFX_ST_VERIFICATION_FAILURE);
        if ((rc = run_one(groups[g], tmp.data()))) { set_error("group " + std::to_string(g) + ": " + afx_last_error()); return rc; }
        for (size_t i = 0; i < grp.count; i++) status[grp.positions[i]] = tmp[i];
      }
    }
    next += grp.count;
  }
  std::vector<int> rcs(m, AFX_OK);
  std::vector<std::string> errs(m);
  auto body = [&](uint32_t k) {
    if (share[k].empty()) return;
    GroupPin pin(group, k, k == 0);   // the member's thread on its device's NUMA node (member 0: the caller's thread, restored)
    rcs[k] = mixed_on_ctx(afx_group_member(group, k), share[k].data(), share[k].size());
    if (rcs[k]) errs[k] = afx_last_error();   // the error string is per thread
  };
  std::vector<std::thread> threads;
  for (uint32_t k = 1; k < m; k++) {
    if (share[k].empty()) continue;
    try { threads.emplace_back(body, k); } catch (const std::system_error&) { body(k); }
  }
  body(0);
  for (std::thread& t : threads) t.join();
  // shape_out and the like were written into the shares' copies of the group structs: hand them back
  {
    std::vector<size_t> at(m, 0);
    uint32_t t2 = 0;
    for (size_t g = 0; g < n_groups; g++) {
      if (!groups[g].count || !(small && groups[g].count <= small)) continue;
      const uint32_t k = t2++ % m;
      const uint64_t* keep = groups[g].positions;
      groups[g] = share[k][at[k]++];
      groups[g].positions = keep;
    }
  }
  for (uint32_t k = 0; k < m; k++)
    if (rcs[k]) { set_error("member " + std::to_string(k) + ": " + errs[k]); return rcs[k]; }
  return AFX_OK;
}

}  // namespace

extern "C" int afx_verify_presentations_mixed(afx_ctx* ctx, const afx_presentation_group* groups, size_t n_groups, uint8_t* status,
                                              size_t status_len) try {
  if (!ctx) { set_error("null argument"); return AFX_E_BAD_ARGS; }
  return run_groups(ctx, groups, n_groups, status, status_len, [&](const afx_presentation_group& G, uint8_t* st) {
    return afx_verify_presentations(ctx, &G.shape, &G.batch, G.count, st);
  });
} catch (...) { return afx::exception_rc(); }

extern "C" int afx_group_verify_presentations_mixed(afx_group* group, const afx_presentation_group* groups, size_t n_groups, uint8_t* status, size_t status_len) try {
  if (!group) { set_error("null argument"); return AFX_E_BAD_ARGS; }
  std::vector<afx_presentation_group> mine(groups, groups + (groups ? n_groups : 0));   // the shares take copies; what the calls write into them (shape_out) comes back
  const int rc = run_groups_on_devices(group, mine.data(), n_groups, status, status_len,
    [&](afx_ctx* c, afx_presentation_group* sub, size_t n) { return afx_verify_presentations_mixed(c, sub, n, status, status_len); },
    [&](afx_presentation_group& G, uint8_t* st) { return afx_group_verify_presentations(group, &G.shape, &G.batch, G.count, st); });

  return rc;
} catch (...) { return afx::exception_rc(); }

// Issuer::issue over requests of several attribute layouts (/root/reference/src/issuer.rs:111-124; kinds per attribute: src/amacs.rs:168-179)
extern "C" int afx_issue_mixed(afx_ctx* ctx, const afx_issue_group* groups, size_t n_groups, uint8_t* status, size_t status_len) try {
  if (!ctx) { set_error("null argument"); return AFX_E_BAD_ARGS; }
  return run_groups(ctx, groups, n_groups, status, status_len, [&](const afx_issue_group& G, uint8_t* st) {
    return afx_issue(ctx, &G.requests, &G.rnd, G.count, &G.out, st);
  });
} catch (...) { return afx::exception_rc(); }
extern "C" int afx_group_issue_mixed(afx_group* group, const afx_issue_group* groups, size_t n_groups, uint8_t* status, size_t status_len) try {
  if (!group) { set_error("null argument"); return AFX_E_BAD_ARGS; }
  std::vector<afx_issue_group> mine(groups, groups + (groups ? n_groups : 0));   // the shares take copies; what the calls write into them (shape_out) comes back
  const int rc = run_groups_on_devices(group, mine.data(), n_groups, status, status_len,
    [&](afx_ctx* c, afx_issue_group* sub, size_t n) { return afx_issue_mixed(c, sub, n, status, status_len); },
    [&](afx_issue_group& G, uint8_t* st) { return afx_group_issue(group, &G.requests, &G.rnd, G.count, &G.out, st); });

  return rc;
} catch (...) { return afx::exception_rc(); }

// CredentialIssuance::verify over issuances of several layouts (src/issuer.rs:48-57)
extern "C" int afx_verify_issuances_mixed(afx_ctx* ctx, const afx_issuance_group* groups, size_t n_groups, uint8_t* status, size_t status_len) try {
  if (!ctx) { set_error("null argument"); return AFX_E_BAD_ARGS; }
  return run_groups(ctx, groups, n_groups, status, status_len, [&](const afx_issuance_group& G, uint8_t* st) {
    return afx_verify_issuances(ctx, &G.attrs, &G.issuances, G.n_responses, G.count, st);
  });
} catch (...) { return afx::exception_rc(); }
extern "C" int afx_group_verify_issuances_mixed(afx_group* group, const afx_issuance_group* groups, size_t n_groups, uint8_t* status, size_t status_len) try {
  if (!group) { set_error("null argument"); return AFX_E_BAD_ARGS; }
  std::vector<afx_issuance_group> mine(groups, groups + (groups ? n_groups : 0));   // the shares take copies; what the calls write into them (shape_out) comes back
  const int rc = run_groups_on_devices(group, mine.data(), n_groups, status, status_len,
    [&](afx_ctx* c, afx_issuance_group* sub, size_t n) { return afx_verify_issuances_mixed(c, sub, n, status, status_len); },
    [&](afx_issuance_group& G, uint8_t* st) { return afx_group_verify_issuances(group, &G.attrs, &G.issuances, G.n_responses, G.count, st); });

  return rc;
} catch (...) { return afx::exception_rc(); }

// AnonymousCredential::show over credentials of several layouts (src/credential.rs:37-46); every group reports its own shape
extern "C" int afx_show_mixed(afx_ctx* ctx, afx_show_group* groups, size_t n_groups, uint8_t* status, size_t status_len) try {
  if (!ctx) { set_error("null argument"); return AFX_E_BAD_ARGS; }
  return run_groups(ctx, groups, n_groups, status, status_len, [&](afx_show_group& G, uint8_t* st) {
    return afx_show(ctx, &G.creds, G.keypairs, &G.rnd, G.count, &G.out, &G.shape_out, st);
  });
} catch (...) { return afx::exception_rc(); }
extern "C" int afx_group_show_mixed(afx_group* group, afx_show_group* groups, size_t n_groups, uint8_t* status, size_t status_len) try {
  if (!group) { set_error("null argument"); return AFX_E_BAD_ARGS; }
  std::vector<afx_show_group> mine(groups, groups + (groups ? n_groups : 0));   // the shares take copies; what the calls write into them (shape_out) comes back
  const int rc = run_groups_on_devices(group, mine.data(), n_groups, status, status_len,
    [&](afx_ctx* c, afx_show_group* sub, size_t n) { return afx_show_mixed(c, sub, n, status, status_len); },
    [&](afx_show_group& G, uint8_t* st) { return afx_group_show(group, &G.creds, G.keypairs, &G.rnd, G.count, &G.out, &G.shape_out, st); });
  if (!rc) for (size_t g = 0; g < n_groups; g++) groups[g].shape_out = mine[g].shape_out;
  return rc;
} catch (...) { return afx::exception_rc(); }

extern "C" int afx_wire_section_bytes(const uint8_t* blob, size_t len, size_t* section_len_out) try {
  if (!blob || !section_len_out) { set_error("null argument"); return AFX_E_BAD_ARGS; }
  if (len < 32 || memcmp(blob, "AFXP", 4) != 0 || rd32(blob + 4) != 1) { set_error("not an AFXP v1 section"); return AFX_E_BAD_ARGS; }
  const uint64_t count = rd32(blob + 8), cells = rd32(blob + 12);
  const uint32_t n = rd32(blob + 16), nr = rd32(blob + 20), hs = rd32(blob + 24), ne = rd32(blob + 28);
  if (n > AFX_MAX_ATTRIBUTES || nr > 3 + AFX_MAX_ATTRIBUTES || hs > AFX_MAX_ATTRIBUTES || ne > AFX_MAX_ATTRIBUTES) { set_error("shape field out of range"); return AFX_E_BAD_ARGS; }
  const size_t hdr = (32 + (size_t)n + 2 * (size_t)hs + 2 * (size_t)ne + 31) & ~size_t(31);
  if (cells > 4 + 3 * (uint64_t)AFX_MAX_ATTRIBUTES + 14 * (uint64_t)AFX_MAX_ATTRIBUTES + 3) { set_error("cells_per_record out of range"); return AFX_E_BAD_ARGS; }
  const uint64_t total = (uint64_t)hdr + count * cells * 32;   // < 2^32 * 2^10 * 2^5: no overflow
  if (total > len) { set_error("section runs past the end of the blob"); return AFX_E_BAD_ARGS; }
  *section_len_out = (size_t)total;
  return AFX_OK;
} catch (...) { return afx::exception_rc(); }

// `verify_section(section bytes, length, status, cap, &n)`: one same-shape AFXP batch on a context or on a group of them
// (one context; the group form below deals the shape groups out to its members)
template <class VerifySection>
static int mixed_wire(afx_ctx* ctx, const uint8_t* blob, size_t len, uint8_t* status, size_t status_cap, size_t* count_out, VerifySection&& verify_section) {
  if ((!blob && len) || !count_out || (!status && status_cap)) { set_error("null argument"); return AFX_E_BAD_ARGS; }
  struct Section { size_t off, len, hdr, count, first; };
  struct Group { std::vector<Section> sections; size_t count = 0; };
  std::map<std::string, Group> by_shape;
  std::vector<std::string> order;   // groups in order of first appearance
  size_t total = 0;
  for (size_t off = 0; off < len;) {
    size_t sl = 0;
    int rc = afx_wire_section_bytes(blob + off, len - off, &sl);
    if (rc) { set_error("section at byte " + std::to_string(off) + ": " + afx_last_error()); return rc; }
    afx_shape sh;
    size_t cnt = 0, rec = 0;
    if ((rc = afx_wire_parse(blob + off, sl, &sh, &cnt, &rec))) { set_error("section at byte " + std::to_string(off) + ": " + afx_last_error()); return rc; }
    const std::string key = shape_key(sh);
    auto it = by_shape.find(key);
    if (it == by_shape.end()) { it = by_shape.emplace(key, Group()).first; order.push_back(key); }
    it->second.sections.push_back({ off, sl, rec, cnt, total });
    it->second.count += cnt;
    total += cnt;
    off += sl;
  }
  *count_out = total;
  if (total > status_cap) { set_error("status buffer too small"); return AFX_E_BAD_ARGS; }
  // with a context: the small groups are collected and run as one set of launches (see run_groups)
  std::unique_ptr<CtxLock> lock;
  std::unique_ptr<afx::Session> ses;
  int rc = AFX_OK;
  // (as in run_groups: with the collector on, the stream's shape groups join the collecting session)
  afx::Deferred deferred;
  std::unique_ptr<afx::DeferScope> defer;
  const bool join = ctx && order.size() > 1 && joins_the_collector(ctx);
  if (join) defer.reset(new afx::DeferScope(&deferred));
  if (ctx && order.size() > 1 && !join) lock.reset(new CtxLock(ctx));
  if (lock && ctx->small_batch_items && !ctx->trace && !ctx->session) {
    ses.reset(new afx::Session(ctx));
    if ((rc = ses->ensure_images(0, 0))) return rc;
    uint64_t width = 0;   // as in run_groups
    for (size_t gi = 0; gi < order.size(); gi++) {
      const size_t cnt = by_shape[order[gi]].count;
      if (cnt && cnt <= ctx->small_batch_items) width += (cnt + 63) / 64;
    }
    ctx->merge_class = afx_ctx::merge_class_of(width);
  }
  struct WidthReset { afx_ctx* c; ~WidthReset() { if (c) c->merge_class = 0; } } width_reset = { ses ? ctx : nullptr };
  std::vector<uint8_t> merged;
  std::vector<std::vector<uint8_t>> sts(order.size());   // statuses of the groups whose sections are not adjacent: scattered after the flush
  try {   // (an exception must not pass the drain below: see run_groups)
  for (size_t gi = 0; gi < order.size() && !rc; gi++) {
    const Group& G = by_shape[order[gi]];
    if (G.count == 0) continue;
    size_t got = 0;
    const bool collect = ses && G.count <= ctx->small_batch_items;
    if (ses && !collect) {
      if ((rc = ses->flush())) break;
      ses->paused = true;
    }
    if (G.sections.size() == 1) {   // the section as it lies in the caller's blob; its statuses are contiguous in the stream
      const Section& S = G.sections[0];
      rc = verify_section(blob + S.off, S.len, status + S.first, S.count, &got);
    } else if (G.count > 0xffffffffu) {
      set_error("too many presentations of one shape");
      rc = AFX_E_BAD_ARGS;
    } else {
      // one header (the first section's, with the group's count) and every section's records behind it
      const Section& S0 = G.sections[0];
      size_t bytes = S0.hdr;
      for (const Section& S : G.sections) bytes += S.len - S.hdr;
      merged.resize(bytes);
      memcpy(merged.data(), blob + S0.off, S0.hdr);
      const uint32_t c32 = (uint32_t)G.count;
      for (int b = 0; b < 4; b++) merged[8 + b] = (uint8_t)(c32 >> (8 * b));
      size_t w = S0.hdr;
      for (const Section& S : G.sections) { memcpy(merged.data() + w, blob + S.off + S.hdr, S.len - S.hdr); w += S.len - S.hdr; }
      sts[gi].assign(G.count, AFX_ST_VERIFICATION_FAILURE);
      rc = verify_section(merged.data(), merged.size(), sts[gi].data(), sts[gi].size(), &got);   // (a collected call copies its records at once)
    }
    if (ses) ses->paused = false;
  }
  } catch (...) {
    if (!join) throw;
    rc = afx::exception_rc();
  }
  if (ses) {
    if (rc) ses->drop();
    else rc = ses->flush();
    ses.reset();
  }
  if (join) {
    CtxLock lk(ctx, true);
    const int rc2 = afx::drain_deferred(ctx, deferred);   // (also after a failure: the staged records point into this request's buffers)
    if (!rc) rc = rc2;
    defer.reset();
  }
  if (rc) return rc;
  for (size_t gi = 0; gi < order.size(); gi++) {
    if (sts[gi].empty()) continue;
    size_t r = 0;
    for (const Section& S : by_shape[order[gi]].sections) { memcpy(status + S.first, sts[gi].data() + r, S.count); r += S.count; }
  }
  return AFX_OK;
}
extern "C" int afx_verify_presentations_mixed_wire(afx_ctx* ctx, const uint8_t* blob, size_t len, uint8_t* status, size_t status_cap,
                                                   size_t* count_out) try {
  if (!ctx) { set_error("null argument"); return AFX_E_BAD_ARGS; }
  return mixed_wire(ctx, blob, len, status, status_cap, count_out, [&](const uint8_t* b, size_t l, uint8_t* st, size_t cap, size_t* n) {
    return afx_verify_presentations_wire(ctx, b, l, st, cap, n);
  });
} catch (...) { return afx::exception_rc(); }
// The same stream on a group of devices.  Shape groups small enough for the latency plan are dealt out to the members in turn and
// every member verifies its share - sections of several shapes - as ONE collected request on a host thread of its own (the
// single-context entry point above: one upload, one set of launches, one download per member); a large shape group is split over all
// the members by the group's batch call.  Only bytes move on the host: a member's share is its sections copied back to back.
extern "C" int afx_group_verify_presentations_mixed_wire(afx_group* group, const uint8_t* blob, size_t len, uint8_t* status, size_t status_cap,
                                                         size_t* count_out) try {
  if (!group) { set_error("null argument"); return AFX_E_BAD_ARGS; }
  if ((!blob && len) || !count_out || (!status && status_cap)) { set_error("null argument"); return AFX_E_BAD_ARGS; }
  const uint32_t m = afx_group_size(group);
  if (m == 0) { set_error("empty group"); return AFX_E_BAD_ARGS; }
  struct Section { size_t off, len, hdr, count, first; };
  struct Group { std::vector<Section> sections; size_t count = 0; };
  std::map<std::string, Group> by_shape;
  std::vector<std::string> order;   // groups in order of first appearance
  size_t total = 0;
  for (size_t off = 0; off < len;) {
    size_t sl = 0;
    int rc = afx_wire_section_bytes(blob + off, len - off, &sl);
    if (rc) { set_error("section at byte " + std::to_string(off) + ": " + afx_last_error()); return rc; }
    afx_shape sh;
    size_t cnt = 0, rec = 0;
    if ((rc = afx_wire_parse(blob + off, sl, &sh, &cnt, &rec))) { set_error("section at byte " + std::to_string(off) + ": " + afx_last_error()); return rc; }
    const std::string key = shape_key(sh);
    auto it = by_shape.find(key);
    if (it == by_shape.end()) { it = by_shape.emplace(key, Group()).first; order.push_back(key); }
    it->second.sections.push_back({ off, sl, rec, cnt, total });
    it->second.count += cnt;
    total += cnt;
    off += sl;
  }
  *count_out = total;
  if (total > status_cap) { set_error("status buffer too small"); return AFX_E_BAD_ARGS; }
  const uint32_t small = afx_group_small_batch_items(group);
  // a member's share: its sections back to back, and where each section's statuses belong in the caller's array
  struct Share { std::vector<uint8_t> bytes; std::vector<std::pair<size_t, size_t>> ranges; size_t items = 0; };
  std::vector<Share> share(m);
  uint32_t turn = 0;
  std::vector<uint8_t> merged, st;
  for (const std::string& key : order) {
    const Group& G = by_shape[key];
    if (G.count == 0) continue;
    if (small && G.count <= small) {
      Share& S = share[turn++ % m];
      for (const Section& sec : G.sections) {
        S.bytes.insert(S.bytes.end(), blob + sec.off, blob + sec.off + sec.len);
        S.ranges.push_back({ sec.first, sec.count });
        S.items += sec.count;
      }
      continue;
    }
    // a large group: over all the members, now (one header with the group's count and every section's records behind it)
    if (G.count > 0xffffffffu) { set_error("too many presentations of one shape"); return AFX_E_BAD_ARGS; }
    size_t got = 0;
    int rc;
    if (G.sections.size() == 1) {
      const Section& S0 = G.sections[0];
      rc = afx_group_verify_presentations_wire(group, blob + S0.off, S0.len, status + S0.first, S0.count, &got);
    } else {
      const Section& S0 = G.sections[0];
      size_t bytes = S0.hdr;
      for (const Section& sec : G.sections) bytes += sec.len - sec.hdr;
      merged.resize(bytes);
      memcpy(merged.data(), blob + S0.off, S0.hdr);
      const uint32_t c32 = (uint32_t)G.count;
      for (int b = 0; b < 4; b++) merged[8 + b] = (uint8_t)(c32 >> (8 * b));
      size_t w = S0.hdr;
      for (const Section& sec : G.sections) { memcpy(merged.data() + w, blob + sec.off + sec.hdr, sec.len - sec.hdr); w += sec.len - sec.hdr; }
      st.assign(G.count, AFX_ST_VERIFICATION_FAILURE);
      rc = afx_group_verify_presentations_wire(group, merged.data(), merged.size(), st.data(), st.size(), &got);
      size_t r = 0;
      if (!rc) for (const Section& sec : G.sections) { memcpy(status + sec.first, st.data() + r, sec.count); r += sec.count; }
    }
    if (rc) return rc;
  }
  std::vector<int> rcs(m, AFX_OK);
  std::vector<std::string> errs(m);
  auto body = [&](uint32_t k) {
    Share& S = share[k];
    if (S.bytes.empty()) return;
    GroupPin pin(group, k, k == 0);   // the member's thread on its device's NUMA node (member 0: the caller's thread, restored)
    std::vector<uint8_t> mine(S.items, AFX_ST_VERIFICATION_FAILURE);
    size_t got = 0;
    rcs[k] = afx_verify_presentations_mixed_wire(afx_group_member(group, k), S.bytes.data(), S.bytes.size(), mine.data(), mine.size(), &got);
    if (rcs[k]) { errs[k] = afx_last_error(); return; }   // the error string is per thread
    size_t r = 0;
    for (const auto& rg : S.ranges) { memcpy(status + rg.first, mine.data() + r, rg.second); r += rg.second; }
  };
  std::vector<std::thread> threads;
  for (uint32_t k = 1; k < m; k++) {
    if (share[k].bytes.empty()) continue;
    try { threads.emplace_back(body, k); } catch (const std::system_error&) { body(k); }
  }
  body(0);
  for (std::thread& t : threads) t.join();
  for (uint32_t k = 0; k < m; k++)
    if (rcs[k]) { set_error("member " + std::to_string(k) + ": " + errs[k]); return rcs[k]; }
  return AFX_OK;
} catch (...) { return afx::exception_rc(); }
