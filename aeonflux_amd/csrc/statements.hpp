// Helpers shared by the statement translation units (statements.cpp: context + verification,
// statements_prove.cpp: issuance and presentation provers).
#pragma once
#include <string.h>
#include <functional>
#include <stdexcept>
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

inline int run_chunked(afx_ctx* c, size_t count, const BuildFn& build) {
  AFX_HIP(hipSetDevice(c->device));
  // all chunks of one call run on one lane; calls alternate lanes only when the caller switched pipelining on
  const int lane = (c->pipelining && !c->force_lane0) ? (int)(c->lane_next++ & 1u) : 0;
  uint32_t chunk = c->chunk_items ? c->chunk_items : CHUNK_DEFAULT;
  for (size_t off = 0; off < count;) {
    const uint32_t cc = (uint32_t)std::min<size_t>(chunk, count - off);
    try {
      Assembler sizing(c, cc, true, lane);
      build(sizing, off, cc);
      if (!sizing.plan_error.empty()) { set_error(sizing.plan_error); return AFX_E_BAD_ARGS; }
      int rc = c->lane[lane].ws.ensure(sizing.total_ws_bytes());
      if (rc) {
        // the device cannot hold this pass's workspace: take smaller passes (an engine on a shared or smaller GPU still works)
        if (chunk <= 4096 || cc <= 4096) return rc;
        (void)hipGetLastError();
        chunk >>= 1;
        continue;
      }
      if (sizing.blob_bytes() > BLOB_CAP) { set_error("plan blob exceeds its fixed capacity"); return AFX_E_BAD_ARGS; }
      Assembler as(c, cc, false, lane);
      build(as, off, cc);
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
  std::vector<afx_scalarop_job> scalarop;
  std::vector<afx_pointop_job> pointop;
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
    as.pointop(js.pointop);
    as.msm(js.msm1);
    as.msm(js.msm2);
    as.hash(js.hash);
  }
  as.finish(status_dev, fail_code);
}

struct Stager {
  afx_ctx* c;
  size_t bytes = 0;
  struct Copy { size_t off; const uint8_t* src; size_t len; };
  std::vector<Copy> copies;
  // a staged (host-pointer) call reads its results back on lane 0's stream: keep its device work on lane 0
  explicit Stager(afx_ctx* ctx) : c(ctx) { c->force_lane0++; }
  ~Stager() { c->force_lane0--; }
  Stager(const Stager&) = delete;
  Stager& operator=(const Stager&) = delete;
  // reserve `len` bytes, to be filled from host `src` (or left for output when src == nullptr); returns offset
  size_t add(const uint8_t* src, size_t len) {
    const size_t off = (bytes + 255) & ~size_t(255);
    bytes = off + len;
    if (src) copies.push_back({ off, src, len });
    return off;
  }
  int upload() {
    int rc = c->staging.ensure(bytes + 256);
    if (rc) return rc;
    for (const Copy& k : copies) AFX_HIP(hipMemcpyAsync((uint8_t*)c->staging.p + k.off, k.src, k.len, hipMemcpyHostToDevice, c->stream));
    return AFX_OK;
  }
  uint8_t* dev(size_t off) const { return (uint8_t*)c->staging.p + off; }
};
