// Engine core: device buffers, launch assembler, Schnorr constraint-system builder.
#include "engine.hpp"
#include <stdio.h>
#include <string.h>
#include <algorithm>
#include <map>
#include <tuple>
#include <new>
#include <system_error>
#include <stdexcept>

namespace afx {
namespace { size_t job_size(LaunchKind k); }   // bytes of one job of a launch kind (below, with the relocation)

static thread_local std::string g_error;
void set_error(const std::string& s) { g_error = s; }
int exception_rc() noexcept {
  try {
    throw;
  } catch (const std::bad_alloc&) {
    try { set_error("out of host memory"); } catch (...) {}
    return AFX_E_NO_MEMORY;
  } catch (const std::system_error& e) {
    try { set_error(std::string("system: ") + e.what()); } catch (...) {}
    return AFX_E_NO_MEMORY;
  } catch (const std::exception& e) {
    try { set_error(std::string("internal: ") + e.what()); } catch (...) {}
    return AFX_E_BAD_ARGS;
  } catch (...) {
    return AFX_E_BAD_ARGS;
  }
}
const char* last_error() { return g_error.c_str(); }

static thread_local bool g_alloc_failed = false;
bool device_alloc_failed() { return g_alloc_failed; }
int DevBuf::ensure(size_t n) {
  g_alloc_failed = false;
  if (n <= cap) return AFX_OK;
  if (p) {
    // The buffer may still be in use: the *_dev entry points are asynchronous and the lanes' streams are non-blocking, so
    // neither the null-stream memset below nor a later reuse of the freed range is ordered after their kernels.  Growing a
    // buffer is rare (once per new high-water mark): wait for the whole device first.
    hipError_t e = hipDeviceSynchronize();
    if (e != hipSuccess) { set_error(std::string("hipDeviceSynchronize: ") + hipGetErrorString(e)); return AFX_E_HIP; }
    // workspace / staging hold blindings, y_i*m_i products, staged user keys, key-derived tables: wipe first
    if (sensitive) (void)hipMemset(p, 0, cap);
    e = hipFree(p);
    p = nullptr; cap = 0;
    if (e != hipSuccess) { set_error(std::string("hipFree: ") + hipGetErrorString(e)); return AFX_E_HIP; }
  }
  const size_t want = (n + (size_t(1) << 20) - 1) & ~((size_t(1) << 20) - 1);
  hipError_t e = hipMalloc(&p, want);
  if (e != hipSuccess) { p = nullptr; g_alloc_failed = true; set_error("hipMalloc(" + std::to_string(want) + "): " + hipGetErrorString(e)); return AFX_E_HIP; }
  cap = want;
  return AFX_OK;
}
void DevBuf::release(bool wipe) {
  if (!p) return;
  if (wipe) (void)hipMemset(p, 0, cap);
  (void)hipFree(p);
  p = nullptr; cap = 0;
}

// ------------------------------------------------------------------------------------------------
// Assembler
// ------------------------------------------------------------------------------------------------
// memset that the optimiser may not drop (the byte-by-byte volatile loop this replaces was 43 % of a small call's host time:
// an Assembler wipes its ~150 KB plan blob on destruction)
static void secure_zero(void* p, size_t n) {
  if (!n) return;
  memset(p, 0, n);
  __asm__ __volatile__("" : : "r"(p) : "memory");
}
Assembler::~Assembler() { secure_zero(blob_.data(), blob_.size()); }
// Provisional bases: non-canonical addresses (bit 62 set: never a valid user-space pointer on x86-64, never a device allocation),
// one window per range and per variant, so that a pointer's range is unambiguous and a pointer that escaped relocation faults
// instead of reading somewhere plausible.
static uint8_t* provisional_base(int range, int variant) { return (uint8_t*)(uintptr_t)(0x4000000000000000ull + ((uint64_t)(1 + range + 4 * variant) << 44)); }
Assembler::Assembler(afx_ctx* c, uint32_t cnt, int variant) : ctx(c), count(cnt) {
  blob_.reserve(1 << 16);
  blob_base_ = provisional_base(0, variant);
  ws_base_ = provisional_base(1, variant);
  blob_alloc(sizeof(afx_pass), 16);   // the pass first (finish_plan fills it in): offset 0 of every plan's blob
  bad_ = (uint32_t*)ws_alloc(sizeof(uint32_t) * (size_t)count);
  Launch l;
  l.kind = L_FILL_BAD;   // its one job is written by finish_plan, when fail_all is known
  l.njobs = 1;
  l.jobs_off = blob_alloc(sizeof(afx_fill_job), 16);
  launches.push_back(l);
}
uint8_t* Assembler::ws_alloc(size_t bytes) {
  const size_t off = (ws_off_ + 255) & ~size_t(255);
  ws_off_ = off + bytes;
  return ws_base_ + off;
}
int32_t* Assembler::new_var() { return (int32_t*)ws_alloc(sizeof(int32_t) * AFX_VAR_DWORDS * (size_t)count); }
uint8_t* Assembler::new_enc() { return ws_alloc(32 * (size_t)count); }
uint8_t* Assembler::new_wide() { return ws_alloc(64 * (size_t)count); }
uint64_t* Assembler::new_state() { return (uint64_t*)ws_alloc(sizeof(uint64_t) * 25 * (size_t)count); }
size_t Assembler::blob_alloc(size_t bytes, size_t align) {
  const size_t off = (blob_.size() + align - 1) & ~(align - 1);
  blob_.resize(off + bytes);
  return off;
}
// A job as it goes into the blob: field by field into zeroed storage for the structs that have padding, so that equal plans are
// equal BYTES (plan self-check, and no stack garbage travels to the device)
template <class T> static void put_job(uint8_t* dst, const T& j) { memcpy(dst, &j, sizeof j); }
template <> void put_job(uint8_t* dst, const afx_decode_job& j) {
  afx_decode_job z; memset(&z, 0, sizeof z);
  z.enc = j.enc; z.out = j.out; z.reject_identity = j.reject_identity; z.elligator = j.elligator;
  memcpy(dst, &z, sizeof z);
}
template <> void put_job(uint8_t* dst, const afx_pointop_job& j) {
  afx_pointop_job z; memset(&z, 0, sizeof z);
  z.a = j.a; z.b = j.b; z.b_const = j.b_const; z.sa = j.sa; z.sb = j.sb; z.out = j.out; z.out_enc = j.out_enc; z.reject_identity = j.reject_identity;
  memcpy(dst, &z, sizeof z);
}
template <> void put_job(uint8_t* dst, const afx_scalarop_job& j) {
  afx_scalarop_job z; memset(&z, 0, sizeof z);
  z.a = j.a; z.a_stride = j.a_stride; z.b = j.b; z.b_stride = j.b_stride; z.c = j.c; z.c_stride = j.c_stride; z.negate = j.negate; z.out = j.out;
  memcpy(dst, &z, sizeof z);
}
template <class T>
void Assembler::add_jobs(LaunchKind k, const std::vector<T>& jobs) {
  if (jobs.empty()) return;
  Launch l;
  l.kind = k;
  l.njobs = (uint32_t)jobs.size();
  l.jobs_off = blob_alloc(sizeof(T) * jobs.size(), 16);
  for (size_t i = 0; i < jobs.size(); i++) put_job(blob_.data() + l.jobs_off + sizeof(T) * i, jobs[i]);
  launches.push_back(l);
}
void Assembler::decode(const std::vector<afx_decode_job>& jobs) {
  stats.decodings += jobs.size();
  stats.field_mul += AFX_DECODE_MUL * jobs.size();
  stats.field_sq += AFX_DECODE_SQ * jobs.size();
  stats.chain_mul += AFX_CHAIN_SQRT_MUL * jobs.size();
  stats.chain_sq += AFX_CHAIN_SQRT_SQ * jobs.size();
  if (!pending_maps_.empty()) {
    // the Elligator maps a small pass queued (from_uniform) ride in this launch, a grid row each beside the decodings; their sums follow
    std::vector<afx_decode_job> all(pending_maps_);
    all.insert(all.end(), jobs.begin(), jobs.end());
    pending_maps_.clear();
    add_jobs(L_DECODE, all);
    launches.back().odd = 1;
    flush_maps();
    return;
  }
  add_jobs(L_DECODE, jobs);
}
// what from_uniform queued and no decode() call took along: a launch of its own, then the sums
void Assembler::flush_maps() {
  if (!pending_maps_.empty()) {
    std::vector<afx_decode_job> maps;
    maps.swap(pending_maps_);
    add_jobs(L_DECODE, maps);
    launches.back().odd = 1;
  }
  if (!pending_map_sums_.empty()) {
    std::vector<afx_pointop_job> sums;
    sums.swap(pending_map_sums_);
    pointop(sums);
  }
}
void Assembler::sccheck(const std::vector<afx_sccheck_job>& jobs) { add_jobs(L_SCCHECK, jobs); }
void Assembler::pointop(const std::vector<afx_pointop_job>& jobs) {
  flush_maps();
  for (const afx_pointop_job& j : jobs) {
    stats.var_additions += j.sb != 0;
    stats.encodings += j.out_enc != nullptr;
    stats.field_mul += (j.sb != 0 ? 9 : 0) + (j.out_enc ? AFX_ENCODE_MUL : 0);
    stats.field_sq += j.out_enc ? AFX_ENCODE_SQ : 0;
    stats.chain_mul += j.out_enc ? AFX_CHAIN_SQRT_MUL : 0;
    stats.chain_sq += j.out_enc ? AFX_CHAIN_SQRT_SQ : 0;
  }
  // A small pass on an idle device: the encodings of these points (a square-root chain each) leave this launch, which stands
  // BEFORE the chains, for the k_compress2x launch after them, where they run beside the other encodings (plain jobs, a row
  // each): the transcript is their only reader.  (reject_identity 2, the strict mode's "must be the identity", stays here.)
  if (small() && (uint64_t)ctx->row_waves(count) * jobs.size() <= 2ull * 4 * ctx->n_cu) {
    std::vector<afx_pointop_job> moved(jobs);
    for (afx_pointop_job& j : moved) {
      if (!j.out_enc || j.reject_identity == 2) continue;
      if (!j.out) j.out = new_var();
      afx_compress_job cj = { j.out, j.out_enc, j.reject_identity, AFX_COMPRESS_PLAIN };
      pending_cjobs_.push_back(cj);
      j.out_enc = nullptr; j.reject_identity = 0;
    }
    add_jobs(L_POINTOP, moved);
    return;
  }
  add_jobs(L_POINTOP, jobs);
}
void Assembler::negenc(const std::vector<afx_negenc_job>& jobs) {
  flush_maps();
  if (jobs.empty()) return;
  stats.encodings += jobs.size();
  stats.field_mul += 11 + (2 * 3 + 2 + 17) * jobs.size();   // one inversion; per point: the value to invert twice, prefix products, the encoding's tail
  stats.field_sq += 254 + 3 * jobs.size();
  stats.chain_mul += AFX_CHAIN_INVERT_MUL; stats.chain_sq += AFX_CHAIN_INVERT_SQ;
  add_jobs(L_NEGENC, jobs);
  add_walk_rows(launches.back(), 0);
}
// grid rows of a k_compress2x / k_negenc / k_table_affine launch: per_row jobs each (0 = all in one row), each row with scratch for
// its prefix products (k_table_affine keeps them inside the table entries)
void Assembler::add_walk_rows(Launch& l, uint32_t per_row) {
  if (per_row == 0 || per_row > l.njobs) per_row = l.njobs;
  std::vector<afx_walk_row> rows;
  for (uint32_t first = 0; first < l.njobs; first += per_row) {
    afx_walk_row r;
    memset(&r, 0, sizeof r);
    r.job_off = first * (uint32_t)job_size(l.kind);
    r.n_jobs = std::min(per_row, l.njobs - first);
    r.pass = 0;
    r.prefix_ws = l.kind == L_TABLE_AFFINE ? nullptr : (int32_t*)ws_alloc(sizeof(int32_t) * 9 * (size_t)r.n_jobs * (size_t)count);
    rows.push_back(r);
  }
  l.nrows = (uint32_t)rows.size();
  l.rows_off = blob_alloc(sizeof(afx_walk_row) * rows.size(), 16);
  memcpy(blob_.data() + l.rows_off, rows.data(), sizeof(afx_walk_row) * rows.size());
}
void Assembler::scalarop(const std::vector<afx_scalarop_job>& jobs) { flush_maps(); add_jobs(L_SCALAROP, jobs); }
void Assembler::hash(const std::vector<afx_hash_program>& progs) {
  flush_maps();
  flush_encodings();   // encodings still queued (Assembler::pointop, compress_also, a small pass's stages) are launched before their reader
  for (const afx_hash_program& p : progs) stats.keccak_permutations += p.n_records;
  add_jobs(L_HASH, progs);
  if (!pending_reductions_.empty()) {   // the blindings these transcripts squeezed out unreduced (SchnorrBuilder::prove_compact, small passes)
    std::vector<afx_reduce_job> rj;
    rj.swap(pending_reductions_);
    add_jobs(L_REDUCE_WIDE, rj);
  }
}
// Width-5 NAF of a canonical scalar (little-endian 32 bytes): digits in {0, +-1, +-3, ..., +-15}, at most one
// nonzero digit in any 5 consecutive positions; returns the position of the highest nonzero digit (-1 for zero).
static int naf5(int8_t out[256], const uint8_t s[32]) {
  uint32_t k[9] = { 0 };
  memcpy(k, s, 32);
  memset(out, 0, 256);
  int top = -1;
  for (int pos = 0; pos < 256; pos++) {
    if (k[0] & 1u) {
      int d = (int)(k[0] & 31u);
      if (d >= 16) d -= 32;
      out[pos] = (int8_t)d;
      top = pos;
      // k -= d
      if (d >= 0) {
        uint64_t borrow = (uint64_t)d;
        for (int i = 0; i < 9 && borrow; i++) { const uint64_t t = (uint64_t)k[i] - borrow; k[i] = (uint32_t)t; borrow = (t >> 63) & 1u; }
      } else {
        uint64_t carry = (uint64_t)(-d);
        for (int i = 0; i < 9 && carry; i++) { const uint64_t t = (uint64_t)k[i] + carry; k[i] = (uint32_t)t; carry = t >> 32; }
      }
    }
    for (int i = 0; i < 8; i++) k[i] = (k[i] >> 1) | (k[i + 1] << 31);
    k[8] >>= 1;
  }
  return top;
}

// s / 2 mod l for a canonical scalar (little-endian 32 bytes): (s + l) >> 1 when s is odd
static void host_half(uint8_t out[32], const uint8_t in[32]) {
  static const uint8_t L[32] = { 0xed, 0xd3, 0xf5, 0x5c, 0x1a, 0x63, 0x12, 0x58, 0xd6, 0x9c, 0xf7, 0xa2, 0xde, 0xf9, 0xde, 0x14,
                                 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0x10 };
  uint8_t t[33];
  unsigned carry = 0;
  for (int i = 0; i < 32; i++) {
    const unsigned v = in[i] + ((in[0] & 1) ? L[i] : 0) + carry;
    t[i] = (uint8_t)v;
    carry = v >> 8;
  }
  t[32] = (uint8_t)carry;
  for (int i = 0; i < 32; i++) out[i] = (uint8_t)((t[i] >> 1) | (t[i + 1] << 7));
}

// host copy of a batch-constant scalar that lives in the context's key block (w, w', x0, x1, y...), or null
static const uint8_t* host_scalar_of(const afx_ctx* c, const uint8_t* dev) {
  const uint8_t* base = (const uint8_t*)c->d_key.p;
  if (!base || dev < base || c->host_key.empty()) return nullptr;
  const size_t off = (size_t)(dev - base);
  if (off % 32 || off / 32 >= c->host_key.size()) return nullptr;
  return c->host_key[off / 32].data();
}

// Order of the grid rows of one k_msm launch.  Rows are dispatched in order, each row = ceil(count / 256) blocks of
// equal duration, onto 2 * n_cu resident blocks: when a row is a fraction 1/m of the device the launch behaves like
// list scheduling on m machines, and longest-first alone leaves the last rows unbalanced (C2 at 2^16 items: two
// "machines", makespan 6 % above the mean load).  So: longest-first assignment to m bins, pairwise move/swap
// refinement of the maximum load, then rows sorted by their planned start time.  cost[] in kilo-cycles per wave.
static std::vector<size_t> balanced_row_order(const std::vector<size_t>& heads, const std::vector<uint32_t>& cost, uint32_t m) {
  std::vector<size_t> order = heads;
  std::stable_sort(order.begin(), order.end(), [&](size_t a, size_t b) { return cost[a] > cost[b]; });
  if (m <= 1 || order.size() <= m) return order;
  std::vector<std::vector<size_t>> bin(m);
  std::vector<uint64_t> load(m, 0);
  for (size_t r : order) {
    const size_t k = std::min_element(load.begin(), load.end()) - load.begin();
    bin[k].push_back(r);
    load[k] += cost[r];
  }
  for (int pass = 0; pass < 64; pass++) {
    const size_t hi = std::max_element(load.begin(), load.end()) - load.begin();
    uint64_t best = load[hi];
    int bi = -1, bj = -1; size_t bk = 0;
    for (size_t k = 0; k < m; k++) {
      if (k == hi) continue;
      for (size_t i = 0; i < bin[hi].size(); i++) {
        const uint64_t ci = cost[bin[hi][i]];
        // move i to bin k
        uint64_t mk = std::max(load[hi] - ci, load[k] + ci);
        if (mk < best) { best = mk; bi = (int)i; bj = -1; bk = k; }
        for (size_t j = 0; j < bin[k].size(); j++) {
          const uint64_t cj = cost[bin[k][j]];
          if (cj >= ci) continue;
          mk = std::max(load[hi] - ci + cj, load[k] + ci - cj);
          if (mk < best) { best = mk; bi = (int)i; bj = (int)j; bk = k; }
        }
      }
    }
    if (bi < 0) break;
    const size_t ri = bin[hi][bi];
    load[hi] -= cost[ri]; load[bk] += cost[ri];
    if (bj >= 0) {
      const size_t rj = bin[bk][bj];
      load[bk] -= cost[rj]; load[hi] += cost[rj];
      bin[hi][bi] = rj; bin[bk][bj] = ri;
    } else {
      bin[hi].erase(bin[hi].begin() + bi);
      bin[bk].push_back(ri);
    }
  }
  struct Slot { uint64_t start; size_t bin, row; };
  std::vector<Slot> plan;
  for (size_t k = 0; k < m; k++) {
    std::stable_sort(bin[k].begin(), bin[k].end(), [&](size_t a, size_t b) { return cost[a] > cost[b]; });
    uint64_t t = 0;
    for (size_t r : bin[k]) { plan.push_back({ t, k, r }); t += cost[r]; }
  }
  std::stable_sort(plan.begin(), plan.end(), [](const Slot& a, const Slot& b) { return a.start != b.start ? a.start < b.start : a.bin < b.bin; });
  order.clear();
  for (const Slot& s : plan) order.push_back(s.row);
  return order;
}

// k_compress2x over `cjobs`, in `groups` grid rows: each row walks its share of an item's jobs with one field inversion
void Assembler::compress(const std::vector<afx_compress_job>& cjobs, uint32_t groups) {
  if (cjobs.empty()) return;
  Launch cl;
  cl.kind = L_COMPRESS;
  cl.njobs = (uint32_t)cjobs.size();
  cl.jobs_off = blob_alloc(sizeof(afx_compress_job) * cjobs.size(), 16);
  memcpy(blob_.data() + cl.jobs_off, cjobs.data(), sizeof(afx_compress_job) * cjobs.size());
  add_walk_rows(cl, groups > 1 ? (cl.njobs + groups - 1) / groups : 0);
  launches.push_back(cl);
  // the plain encodings were counted job by job: replace them by k_compress2x's share (two passes over e, f, g, h per job, one
  // inversion per row)
  // (plain jobs - points encoded as they are, moved here by Assembler::pointop - keep the share they were counted with)
  uint64_t halves = 0;
  for (const afx_compress_job& j : cjobs) halves += j.negate != AFX_COMPRESS_PLAIN;
  const uint64_t rows = cl.nrows;
  stats.field_mul += 22 * halves + 11 * rows; stats.field_mul -= AFX_ENCODE_MUL * halves;
  stats.field_sq += 8 * halves + 254 * rows; stats.field_sq -= AFX_ENCODE_SQ * halves;
  stats.chain_mul += AFX_CHAIN_INVERT_MUL * rows; stats.chain_mul -= AFX_CHAIN_SQRT_MUL * halves;
  stats.chain_sq += AFX_CHAIN_INVERT_SQ * rows; stats.chain_sq -= AFX_CHAIN_SQRT_SQ * halves;
}

// Multiscalar jobs of one statement stage.  Large passes: one lane per (job, item), every job a single chain (msm_list).  Small
// passes (at most afx_ctx_set_small_batch_items items, default 4096): the device is mostly idle and a call's duration is the
// LONGEST chain, so every variable-base term gets a chain of its own (msm_split).
// The queued encodings (compress_also(), the plain jobs of Assembler::pointop and msm_split, and - small passes - everything a stage's
// chains left for k_compress2x): one launch, before their first reader.
void Assembler::flush_encodings() {
  if (pending_cjobs_.empty()) return;
  for (const int32_t* v : pending_half_vars_)
    if (!half_bases_.count(v)) throw std::logic_error("compress_also: the variable does not hold a half");
  pending_half_vars_.clear();
  std::vector<afx_compress_job> cjobs;
  cjobs.swap(pending_cjobs_);
  compress(cjobs, ctx->walk_rows(count, cjobs.size(), small()));
}

void Assembler::msm(std::vector<afx_msm_job> jobs) {
  flush_maps();
  if (jobs.empty()) {
    // encodings queued by compress_also() ride in this call's k_compress2x launch even when it has no chains of its own (a small
    // pass keeps them for the launch in front of their reader: see the end of this function)
    if (!small()) flush_encodings();
    return;
  }
  for (size_t i = 0; i < jobs.size(); i++)
    if (jobs[i].chain_to >= 0 && ((size_t)jobs[i].chain_to >= jobs.size() || jobs[i].chain_to == (int32_t)i)) throw std::logic_error("bad msm chain");
  // Secret-independent addressing (afx_ctx_set_secret_independent_addressing): which terms carry a secret scalar.  In a prover-side
  // plan every scalar but the constant 1.  In a verifier-side plan every term of a job that multiplies by the issuer key (Z of
  // Issuer::verify): the key's own terms, and beside them the per-item scalars DERIVED from it by a public factor (y_i * m_i for a
  // revealed scalar attribute), whose digits would give the key away just the same.  Marked here, before a small pass splits the
  // job into one chain per term.  Such terms never run as a NAF schedule (whose table indices are the key's digits).  On a per-item
  // base they run 2-bit signed windows over a two-entry affine table of which every addition reads both entries and selects; on a
  // generator 6-bit signed windows over the positional tables AFX_SEC_*, the window's 32 multiples loaded one per lane and the digit's
  // taken from the lane that holds it (ds_bpermute_b32, kernels.hip msm_fixed_terms): no load address is made from a digit.
  auto is_key_scalar = [&](const afx_msm_term& t) { return t.scalar_stride == 0 && host_scalar_of(ctx, t.scalar) != nullptr; };
  // producers that leave their half (plan.h afx_msm_job.leave_half): known before the terms are marked, so that a consumer in this
  // same call doubles its scalar too
  for (afx_msm_job& j : jobs) {
    if (j.leave_half && !(j.out_var && j.out_enc && !j.addend && !j.half_var)) j.leave_half = 0;   // nothing to gain or not possible
    if (j.leave_half) half_bases_.insert(j.out_var);
  }
  // compress_also() encodes +-2 * var: var must be (or become, in this call) a half some job left - never a whole point
  for (const int32_t* v : pending_half_vars_)
    if (!half_bases_.count(v)) throw std::logic_error("compress_also: the variable does not hold a half (its producer's leave_half was dropped)");
  pending_half_vars_.clear();
  for (afx_msm_job& j : jobs) {
    for (uint32_t t = 0; t < j.n_var; t++) j.term[t].dbl = half_bases_.count(j.term[t].var) ? 1u : 0u;
    for (uint32_t t = j.n_var; t < j.n_terms; t++) j.term[t].dbl = 0;
    bool key_job = false;
    for (uint32_t t = 0; t < j.n_terms; t++) key_job |= is_key_scalar(j.term[t]);
    for (uint32_t t = 0; t < j.n_terms; t++) {
      j.term[t].secret = (secure() && (secret_scalars || key_job) && j.term[t].scalar != ctx->const_one()) ? 1u : 0u;
      stats.secret_terms += j.term[t].secret;
    }
  }
  std::vector<afx_compress_job> cjobs;
  cjobs.swap(pending_cjobs_);   // compress_also()
  const bool small = this->small();
  // Passes of up to four times that size: the device is full during the windowed launch but not during the one before it,
  // which holds the ONE job that multiplies by the issuer key (Z: a single grid row of ten NAF terms, 16-128 blocks on 256
  // compute units for 1.7 ms).  There that job alone is split into one NAF chain per term and summed, ahead of the others.
  const bool mid = !small && ctx->small_batch_items != 0 && count <= 4 * (uint64_t)ctx->small_batch_items;
  if (small) {
    // One chain per term while the stage's chains find room on the device (4 waves for each of its SIMDs: measured, 2048 ... 6144
    // tried - tools/experiments/r04_chain_room.sh); a wider stage - a few thousand items, or the stages of many merged passes - takes
    // up to four variable-base terms per chain: fewer doublings in all, and the chains would have queued anyway (one call of 4096
    // presentations 3.56 -> 3.15 ms, 8 shapes x 1024 in one request 3.9 -> 3.3 ms, 64 x 16 3.4 -> 3.25 ms).  (A stage of few jobs - Z of Issuer::verify - keeps one term per chain
    // however many passes are merged: its duration is its longest chain's.)
    const uint64_t row_waves = ctx->row_waves(count), room = 4ull * 4 * ctx->n_cu;
    uint32_t per_chain = 1;
    for (; per_chain < 4; per_chain++) {
      uint64_t chains = 0;
      for (const afx_msm_job& j : jobs) chains += (j.n_var + per_chain - 1) / per_chain + (j.n_terms - j.n_var + 5) / 6;
      if (chains * row_waves <= room) break;
    }
    const bool seg = per_chain == 1 && segment_bases(jobs);
    msm_split(std::move(jobs), cjobs, true, per_chain, seg);
  }
  else if (mid) {
    auto naf_term = [&](const afx_msm_term& t) { return t.scalar_stride == 0 && !ctx->fixed_key_schedule && !ctx->secure_plan(secret_scalars) && !t.dbl && host_scalar_of(ctx, t.scalar) != nullptr; };
    std::vector<afx_msm_job> first, rest;
    std::vector<int> new_index(jobs.size(), -1);
    bool ok = true;
    for (size_t i = 0; i < jobs.size(); i++) {
      uint32_t k = 0;
      for (uint32_t t = 0; t < jobs[i].n_var; t++) k += naf_term(jobs[i].term[t]);
      if (k >= 2) first.push_back(jobs[i]); else { new_index[i] = (int)rest.size(); rest.push_back(jobs[i]); }
    }
    for (size_t i = 0; i < jobs.size() && ok; i++) {
      if (jobs[i].chain_to < 0) continue;
      if (new_index[i] >= 0 && new_index[jobs[i].chain_to] < 0) ok = false;   // a split job would wait for one that is not
      if (new_index[i] < 0 && new_index[jobs[i].chain_to] < 0) ok = false;    // producer and consumer both split: msm_split levels them, but
                                                                               // `first` drops its chain links below - keep the list whole
    }
    if (first.empty() || !ok) msm_list(std::move(jobs), false, cjobs);
    else {
      for (afx_msm_job& j : first) j.chain_to = -1;   // their consumers are all in `rest`, which starts after their sums
      for (afx_msm_job& j : rest) j.chain_to = j.chain_to >= 0 ? new_index[j.chain_to] : -1;
      msm_split(std::move(first), cjobs, false);
      msm_list(std::move(rest), false, cjobs);
    }
  } else msm_list(std::move(jobs), false, cjobs);
  // Small passes: the item's commitments are encoded in a row each (8 rows when the device is busy), an inversion each, instead of
  // one serial walk - and not here but in front of their first reader (Assembler::hash, finish: nothing else reads an encoding a
  // stage produced), together with whatever k_pointop left to encode after this stage: one 65 us launch instead of two in a small show.
  if (small) { pending_cjobs_.insert(pending_cjobs_.end(), cjobs.begin(), cjobs.end()); return; }
  compress(cjobs, ctx->walk_rows(count, cjobs.size(), small));
}

// Into how many segments a small prover pass cuts a secret scalar on a per-item base: 8 (4 above 256 items), or what
// afx_ctx_set_plan_variants forces (AFX_VARIANT_SEGMENTS_1 | _2 | _4: tests)
// Does this pass keep secret scalars out of its addresses?  What the context's mode says (afx_ctx_set_secret_independent_addressing) -
// and, whatever it says, a PROVER pass small enough for the segmented chains: there the secret-independent plan is also the faster
// one (16-window chains over kept two-entry tables against 64 windows of four doublings), so mode 0 has nothing to offer it.
bool Assembler::secure() const { return ctx->secure_plan(secret_scalars) || (secret_scalars && segments() > 1); }
uint32_t Assembler::segments() const {
  const uint32_t env = (ctx->variants & AFX_VARIANT_SEGMENTS_1) ? 1u : (ctx->variants & AFX_VARIANT_SEGMENTS_2) ? 2u : (ctx->variants & AFX_VARIANT_SEGMENTS_4) ? 4u : 8u;
  static constexpr uint32_t wide = 32;   // waves of 64 items: measured, up to 2048 items the segments pay
  static_assert(AFX_SECVAR_WINDOWS % 8 == 0 && AFX_POWERS_MAX >= 7, "a scalar's windows divide into up to eight segments");
  // the passes whose chains run four waves each on a device they leave idle (kernels.hip afxk_msm): up to 256 items
  // Prover passes only.  A verifier's pass is ONE stage of public scalars on the presentation's own points: its powers (224 doublings)
  // cost what the whole chain's 252 do, and eight times the chains to sum cost more than the 8-window chains save - measured with
  // the same machinery (4-bit windows take win_off / wins like the 2-bit ones): 1 presentation 0.72 -> 0.73 ms, 256: 0.77 -> 0.90.
  if (env <= 1 || !small() || !secret_scalars || ctx->row_waves(count) > wide) return 1;
  return ctx->row_waves(count) > 4 ? std::min(env, 4u) : env;
}
bool Assembler::segment_bases(const std::vector<afx_msm_job>& jobs) {
  const uint32_t S = segments();
  if (S <= 1) return false;
  // a base one of THIS stage's jobs produces has no powers before the stage runs: its terms keep their full chains
  std::set<const int32_t*> produced;
  for (const afx_msm_job& j : jobs) { if (j.out_var) produced.insert(j.out_var); if (j.half_var) produced.insert(j.half_var); }
  std::vector<afx_powers_job> pj;
  bool any = false;
  for (const afx_msm_job& j : jobs)
    for (uint32_t t = 0; t < j.n_var; t++) {
      const afx_msm_term& tm = j.term[t];
      if (produced.count(tm.var)) continue;
      any = true;
      if (powers_.count(tm.var)) continue;
      afx_powers_job q;
      memset(&q, 0, sizeof q);
      q.src = tm.var;
      q.step = AFX_SECVAR_BITS * (AFX_SECVAR_WINDOWS / S);
      q.n_out = S - 1;
      std::vector<int32_t*> vs;
      for (uint32_t k = 0; k + 1 < S; k++) { vs.push_back(new_var()); q.out[k] = vs.back(); }
      powers_[tm.var] = vs;
      pj.push_back(q);
      const uint64_t d = (uint64_t)q.step * q.n_out;   // doublings: four squarings, four products each (kernels.hip quad_dbl)
      stats.doublings += d; stats.field_sq += 4 * d; stats.field_mul += 4 * d;
    }
  if (!pj.empty()) add_jobs(L_POWERS, pj);
  if (any && secret_scalars) segmenting_ = true;   // (a prover's pass has later stages on the same bases: its narrow tables stay)
  return any;
}

void Assembler::compress_also(const int32_t* var, uint8_t* out_enc, bool negate, uint32_t reject_identity) {
  afx_compress_job cj = { var, out_enc, reject_identity, negate ? 1u : 0u };
  pending_cjobs_.push_back(cj);
  pending_half_vars_.push_back(var);
  stats.encodings++; stats.field_mul += AFX_ENCODE_MUL; stats.field_sq += AFX_ENCODE_SQ; stats.chain_mul += AFX_CHAIN_SQRT_MUL; stats.chain_sq += AFX_CHAIN_SQRT_SQ;   // compress() rewrites this share
}

// One chain per term (var_per_part = 1; up to four terms per chain when the pass's chains would queue on the device:
// afx_ctx::terms_per_chain).  A job with more than one part - its variable-base terms var_per_part at a time, and its fixed-base
// terms six at a time - becomes that many single-part jobs writing partial sums, plus a k_pointsum row adding them (and the addend) up; a job that feeds
// another (chain_to) is summed before the consumer's chains start, so stages run level by level.  no_naf (small passes): no NAF
// schedules - a lone key term costs 64 additions instead of ~43, but shares the windowed launch with every other chain of its stage.
void Assembler::msm_split(std::vector<afx_msm_job> jobs, std::vector<afx_compress_job>& cjobs, bool no_naf, uint32_t var_per_part, bool segs) {
  if (var_per_part == 0) var_per_part = 1;
  const uint32_t S = (segs && var_per_part == 1) ? segments() : 1;
  auto segmented = [&](const afx_msm_term& t) { return S > 1 && t.fixed_idx < 0 && powers_.count(t.var) != 0; };
  const size_t n = jobs.size();
  for (size_t i = 0; i < n; i++)
    if (jobs[i].chain_to >= 0 && ((size_t)jobs[i].chain_to >= n || jobs[i].chain_to == (int32_t)i)) throw std::logic_error("bad msm chain");
  std::vector<int> level(n, 0);
  for (size_t pass = 0; pass <= n; pass++) {
    bool moved = false;
    for (size_t i = 0; i < n; i++)
      if (jobs[i].chain_to >= 0 && level[jobs[i].chain_to] < level[i] + 1) { level[jobs[i].chain_to] = level[i] + 1; moved = true; }
    if (!moved) break;
    if (pass == n) throw std::logic_error("msm chain cycle");
  }
  const int top = *std::max_element(level.begin(), level.end());
  for (int lv = 0; lv <= top; lv++) {
    std::vector<afx_msm_job> subs;
    std::vector<afx_pointsum_job> sums;
    for (size_t i = 0; i < n; i++) {
      if (level[i] != lv) continue;
      afx_msm_job j = jobs[i];
      j.chain_to = -1;
      // fixed-base terms go six to a part: 120 additions, well under the 252 doublings + 64 additions of a variable-base chain
      // (the second commitment of an issuance proof has n + 3 of them: one lane with 380 additions was the longest chain of a
      // small issue call)
      // (a segmenting pass, whose variable-base chains are a quarter or an eighth as long, takes them two or one to a part)
      const uint32_t FIXED_PER_PART = S >= 8 ? 1 : S > 1 ? 2 : 6;
      uint32_t parts = (j.n_var + var_per_part - 1) / var_per_part + (j.n_terms - j.n_var + FIXED_PER_PART - 1) / FIXED_PER_PART;
      for (uint32_t t = 0; t < j.n_var; t++) if (segmented(j.term[t])) parts += S - 1;
      const bool halved = j.out_enc && !j.addend && (!j.out_var || j.leave_half);   // only ever encoded, or leaving its half: halved scalars, k_compress2x (msm_list)
      // Small passes: an encoding that would run INSIDE the chain's kernel (a result with an addend, or one that is also a base) goes
      // to the k_compress2x launch behind the chains instead, as a plain job in a row of its own (as Assembler::pointop does): the
      // square-root chain runs beside the stage's other encodings instead of behind the longest chain, and a launch without encoders
      // may run four waves per chain (kernels.hip afxk_msm: show's first stage 757 -> 480 us).  Same point, same bytes.
      afx_compress_job plain = { nullptr, nullptr, 0, AFX_COMPRESS_PLAIN };
      if (no_naf && small() && j.out_enc && !halved && !j.half_var) {
        if (!j.out_var) j.out_var = new_var();
        plain.var = j.out_var; plain.out_enc = j.out_enc; plain.reject_identity = j.reject_identity;
        j.out_enc = nullptr; j.reject_identity = 0;
        cjobs.push_back(plain);
        stats.encodings++; stats.field_mul += AFX_ENCODE_MUL; stats.field_sq += AFX_ENCODE_SQ; stats.chain_mul += AFX_CHAIN_SQRT_MUL; stats.chain_sq += AFX_CHAIN_SQRT_SQ;
      }
      if (parts <= 1) { subs.push_back(j); continue; }   // a single chain already (its addend, if any, is added by its own lane)
      std::vector<const int32_t*> part_vars;
      auto sub_of = [&](uint32_t first, uint32_t nterms, uint32_t nvar) {
        afx_msm_job s;
        memset(&s, 0, sizeof s);
        s.n_terms = nterms; s.n_var = nvar; s.chain_to = -1;
        for (uint32_t t = 0; t < nterms; t++) s.term[t] = j.term[first + t];
        int32_t* v = new_var();
        if (halved) s.half_var = v; else s.out_var = v;
        part_vars.push_back(v);
        subs.push_back(s);
      };
      for (uint32_t t = 0; t < j.n_var; t += var_per_part) {
        const uint32_t k = std::min(var_per_part, j.n_var - t);
        if (!segmented(j.term[t])) { sub_of(t, k, k); continue; }
        // segment g: windows [g W, (g + 1) W) of the recoded scalar on 2^(bits * g W) times the base: a chain of W windows
        // (2-bit windows for a secret scalar, 4-bit ones otherwise: 256 / S bits either way, which is what the powers step by)
        const uint32_t W = (j.term[t].secret ? AFX_SECVAR_WINDOWS : 64u) / S;
        const std::vector<int32_t*>& pw = powers_[j.term[t].var];
        for (uint32_t g = 0; g < S; g++) {
          sub_of(t, 1, 1);
          afx_msm_job& sg = subs.back();
          if (g) sg.term[0].var = pw[g - 1];
          sg.term[0].win_off = g * W;
          sg.wins = W;
        }
      }
      for (uint32_t first = j.n_var; first < j.n_terms; first += FIXED_PER_PART) sub_of(first, std::min(FIXED_PER_PART, j.n_terms - first), 0);
      afx_pointsum_job sj;
      memset(&sj, 0, sizeof sj);
      sj.parts = put_ptrs(part_vars.data(), part_vars.size());
      sj.n_parts = (uint32_t)part_vars.size();
      sj.addend = j.addend; sj.addend_negate = j.addend_negate;
      sj.out_var = j.out_var;
      sj.reject_identity = j.reject_identity;
      if (halved) {
        sj.half_var = j.leave_half ? j.out_var : new_var();   // leave_half: the half IS what later terms find at out_var
        if (j.leave_half) sj.out_var = nullptr;
        afx_compress_job cj = { sj.half_var, j.out_enc, j.reject_identity, 0 };
        cjobs.push_back(cj);
        stats.encodings++; stats.field_mul += AFX_ENCODE_MUL; stats.field_sq += AFX_ENCODE_SQ; stats.chain_mul += AFX_CHAIN_SQRT_MUL; stats.chain_sq += AFX_CHAIN_SQRT_SQ;   // compress() rewrites this share
      } else {
        sj.out_enc = j.out_enc;
        if (j.out_enc) { stats.encodings++; stats.field_mul += AFX_ENCODE_MUL; stats.field_sq += AFX_ENCODE_SQ; stats.chain_mul += AFX_CHAIN_SQRT_MUL; stats.chain_sq += AFX_CHAIN_SQRT_SQ; }
      }
      stats.var_additions += sj.n_parts - 1 + (sj.addend ? 1 : 0);
      stats.field_mul += 9 * (uint64_t)(sj.n_parts - 1 + (sj.addend ? 1 : 0));
      sums.push_back(sj);
    }
    msm_list(std::move(subs), no_naf, cjobs);
    add_jobs(L_POINTSUM, sums);
    if (!sums.empty())   // (Launch::odd of a sum: some job has sixteen parts or more - kernels.hip afxk_pointsum)
      for (const afx_pointsum_job& sj : sums) if (sj.n_parts >= 16) launches.back().odd = 1;
  }
}

void Assembler::msm_list(std::vector<afx_msm_job> jobs, bool no_naf, std::vector<afx_compress_job>& cjobs) {
  if (jobs.empty()) return;
  // kilo-cycles of VALU issue per wave (profiles/r01_fe_rates_ubench.txt): 63 windows of 4 doublings + the final
  // encoding; 64 additions + the 9-entry table per variable base; AFX_POS_WINDOWS additions per fixed base
  auto cost = [](const afx_msm_job& j) {
    // a narrow job (a secret scalar on a variable base): AFX_SECVAR_WINDOWS additions per variable term, tables of AFX_SECVAR_STORED
    const uint32_t nwins = j.wins ? j.wins : j.narrow ? AFX_SECVAR_WINDOWS : 64u;
    const uint32_t per_var = j.narrow ? (nwins * 13u) / 2u + AFX_SECVAR_STORED * 7u : (470u * nwins) / 64u;
    uint32_t c = (j.n_var ? (1320u * nwins) / (j.narrow ? AFX_SECVAR_WINDOWS : 64u) : 160u) + j.n_uni * (330u + 45u /* more conversions to p3 */) + (j.n_var - j.n_uni) * per_var;
    for (uint32_t t = j.n_var; t < j.n_terms; t++) c += ((j.term[t].secret ? AFX_SEC_WINDOWS * 200u : AFX_POS_WINDOWS * 175u)) / 32u;
    for (uint32_t t = 0; t < j.n_var; t++) c += j.term[t].secret ? (nwins * AFX_SECVAR_STORED * 15u) / 64u : 0u;   // the table scans
    return c;
  };
  const size_t n = jobs.size();
  const bool sec_mode = secure();   // terms were marked by Assembler::msm, before any splitting
  // A segmenting pass on the four-wave chains keeps the entries in the CACHED form k_msm_tables leaves (kind 3): an addition's fourth
  // product runs on the fourth wave at no cost in rounds, and the pass does not wait for k_table_affine's inversion (64 us of a 1-item call)
  // (up to 512 items: such a launch takes the four-wave chains whatever its size, and wider passes are better off on one wave per
  // chain - a 2048-item show 2.59 -> 2.90 ms with them)
  const bool cached_narrow = segmenting_ && !(ctx->variants & AFX_VARIANT_ONE_WAVE_CHAINS) && ctx->row_waves(count) <= 8;
  for (size_t i = 0; i < n; i++)
    if (jobs[i].chain_to >= 0 && ((size_t)jobs[i].chain_to >= n || jobs[i].chain_to == (int32_t)i)) throw std::logic_error("bad msm chain");
  // Jobs whose result is only ever encoded (no consumer of the point itself, no addend) run on halved scalars; k_compress2x
  // encodes the doubles (below, and kernels.hip)
  auto encoded_only = [&](const afx_msm_job& j) { return j.out_enc && !j.addend && !j.half_var && (!j.out_var || j.leave_half); };   // leave_half: msm() vetted it
  // ... and the parts of such a job that msm_split cut up arrive with their half_var set: halved scalars, no encoding of their own
  auto halved = [&](const afx_msm_job& j) { return j.half_var != nullptr || encoded_only(j); };
  // variable bases whose scalar is a batch constant the host knows (the issuer key in Z and in the tag) go first
  // and run a width-5 NAF: ~43 additions each instead of 64, same schedule for every lane
  std::vector<std::vector<int8_t>> naf_of(jobs.size());
  std::vector<std::vector<uint8_t>> nafc_of(jobs.size());
  for (size_t ji = 0; ji < jobs.size(); ji++) {
    afx_msm_job& j = jobs[ji];
    j.n_uni = 0; j.top_bit = 0; j.naf_sched = nullptr;
    std::vector<afx_msm_term> uni, lane;
    std::vector<const uint8_t*> hs;
    for (uint32_t t = 0; t < j.n_var; t++) {
      // afx_ctx_set_fixed_key_schedule: key scalars take the per-item window path (64 additions each, whatever the key)
      const uint8_t* h = (j.term[t].scalar_stride == 0 && !ctx->fixed_key_schedule && !sec_mode && !no_naf && !j.term[t].dbl) ? host_scalar_of(ctx, j.term[t].scalar) : nullptr;
      if (h) { uni.push_back(j.term[t]); hs.push_back(h); } else lane.push_back(j.term[t]);
    }
    if (uni.empty()) continue;
    uint32_t k = 0;
    for (const afx_msm_term& t : uni) j.term[k++] = t;
    for (const afx_msm_term& t : lane) j.term[k++] = t;
    j.n_uni = (uint32_t)uni.size();
    naf_of[ji].assign(256 * uni.size(), 0);
    nafc_of[ji].assign(256, 0);
    int top = lane.empty() ? 0 : 252;   // per-item windows start at bit 252
    for (size_t u = 0; u < uni.size(); u++) {
      int8_t* d = naf_of[ji].data() + 256 * u;
      uint8_t hbuf[32];
      if (halved(j)) host_half(hbuf, hs[u]);   // the device halves the per-item scalars of such a job (msm_recode)
      top = std::max(top, naf5(d, halved(j) ? hbuf : hs[u]));
      secure_zero(hbuf, sizeof hbuf);
      for (int b = 0; b < 256; b++) {
        if (uni[u].negate) d[b] = (int8_t)-d[b];
        nafc_of[ji][b] += d[b] != 0;
      }
    }
    j.top_bit = top;
  }
  for (size_t ji = 0; ji < jobs.size(); ji++) {
    // a secret scalar on a variable base: the job's chain runs AFX_SECVAR_BITS-bit windows (kernels.hip msm_add_var)
    jobs[ji].narrow = 0;
    for (uint32_t t = 0; t < jobs[ji].n_var; t++) if (jobs[ji].term[t].secret) jobs[ji].narrow = 1;
    if (jobs[ji].narrow && jobs[ji].n_uni) throw std::logic_error("a NAF schedule in a job with secret scalars");
    if (!jobs[ji].wins) jobs[ji].wins = jobs[ji].narrow ? AFX_SECVAR_WINDOWS : 64;
    if (jobs[ji].wins > (jobs[ji].narrow ? AFX_SECVAR_WINDOWS : 64u)) throw std::logic_error("a chain longer than its scalar");
    const afx_msm_job& j = jobs[ji];
    const uint64_t nv = j.n_var;
    const uint64_t wins = j.wins, wbits = j.narrow ? AFX_SECVAR_BITS : 4, stored = j.narrow ? AFX_SECVAR_STORED : AFX_TABLE_STORED;
    uint64_t nfa = 0;   // additions of the fixed-base terms: AFX_POS_WINDOWS each, AFX_SEC_WINDOWS for a secret scalar
    for (uint32_t t = j.n_var; t < j.n_terms; t++) nfa += j.term[t].secret ? AFX_SEC_WINDOWS : AFX_POS_WINDOWS;
    stats.msm_jobs++;
    if (j.n_uni) {
      // bit-serial schedule of k_msm's NAF branch
      const uint64_t nl = nv - j.n_uni;
      uint64_t M = j.n_uni * (4 + 1 + 1 + 7 * 9) + nl * (1 + (AFX_TABLE_ENTRIES - 2) * 9), S = j.n_uni * 4;
      stats.table_additions += 7 * j.n_uni + (AFX_TABLE_ENTRIES - 2) * nl;
      for (int b = j.top_bit; b >= 0; b--) {
        const uint64_t nuni = nafc_of[ji][b];
        const uint64_t nadd = nuni + ((b & 3) == 0 ? nl : 0);
        if (b != j.top_bit) { stats.doublings++; S += 4; M += (nadd != 0 || b == 0) ? 4 : 3; }
        stats.var_additions += nadd;
        M += nadd * 4;
        M += nadd * 4 - ((nadd != 0 && b != 0) ? 1 : 0);
      }
      stats.fixed_additions += nfa;
      M += nfa * 7;
      stats.encodings += j.out_enc ? 1 : 0;
      stats.var_additions += j.addend ? 1 : 0;
      if (j.addend) M += 9;
      if (j.out_enc) { M += AFX_ENCODE_MUL; S += AFX_ENCODE_SQ; stats.chain_mul += AFX_CHAIN_SQRT_MUL; stats.chain_sq += AFX_CHAIN_SQRT_SQ; }
      stats.field_mul += M;
      stats.field_sq += S;
      continue;
    }
    stats.doublings += nv ? (wins - 1) * wbits : 0;
    stats.var_additions += wins * nv;
    stats.fixed_additions += nfa;
    stats.table_additions += (stored - 1) * nv;
    stats.encodings += j.out_enc ? 1 : 0;
    stats.var_additions += j.addend ? 1 : 0;
    // field operations, following k_msm's schedule statement by statement
    uint64_t M = 0, S = 0;
    if (nv) {
      // tables: 2dT of P, then (add 4M + to p3 4M + 2dT 1M) per entry; a narrow job's entries leave as X, Y, Z (k_table_affine's
      // share is counted where that launch is emitted)
      M += nv * (1 + (stored - 1) * (j.narrow ? 8 : 9));
      for (int w = (int)wins - 1; w >= 0; w--) {
        if (w != (int)wins - 1) { S += 4 * wbits; M += 3 * (wbits - 1) + 4; }   // the window's doublings to p2, its last to p3
        M += nv * ((j.narrow && !cached_narrow) ? 3 : 4);     // additions of window-table entries (affine ones in a narrow job)
        M += nv * 4 - (w != 0 ? 1 : 0);                       // back to p3; the window's last one skips T
      }
    }
    M += nfa * 7;                                             // fixed bases: positional tables, niels addition + to p3
    if (j.addend) M += 9;
    if (j.out_enc) { M += AFX_ENCODE_MUL; S += AFX_ENCODE_SQ; stats.chain_mul += AFX_CHAIN_SQRT_MUL; stats.chain_sq += AFX_CHAIN_SQRT_SQ; }
    stats.field_mul += M;
    stats.field_sq += S;
  }
  // Launch list.  One kernel per job class (kernels.hip MSM_*), and a job that consumes another job's out_var
  // (chain_to) must sit in a later launch than its producer.  Greedy over the classes in the order NAF, WINDOW,
  // FIXED: launch every pending job of the class whose producer (if any) has been launched; repeat until none is left.
  // For Issuer::verify that is: tables + NAF {Z}, then tables + WINDOW {every constraint, #1 (Z = z*I) included}.
  std::vector<int> producer(n, -1);
  for (size_t i = 0; i < n; i++)
    if (jobs[i].chain_to >= 0) producer[jobs[i].chain_to] = (int)i;
  // small passes (no_naf): jobs of fixed bases only ride in the windowed launch when there is one - the kernel skips their chain -
  // instead of a launch of their own ahead of it: one launch less, and the short rows run beside the long ones
  bool any_window = false;
  for (size_t i = 0; i < n; i++) any_window |= jobs[i].n_var != 0 && jobs[i].n_uni == 0;
  const bool merge_fixed = no_naf && any_window;
  auto kind_of = [&](const afx_msm_job& j) { return j.n_var == 0 ? (merge_fixed ? 1 : 0) : (j.n_uni ? 2 : 1); };
  // Jobs whose result is only ever encoded (no consumer of the point itself, no addend): they run on halved scalars and leave
  // half their sum in a workspace slot; one k_compress2x launch at the end of this list encodes the doubles with a single field
  // inversion per item (kernels.hip).  That is every recomputed or fresh commitment of a Schnorr proof (26 of the 35 encodings
  // of a C3 presentation), the tag's V and the scalar attributes' messages of an issuance.
  std::vector<int32_t*> half_of(n, nullptr);
  for (size_t i = 0; i < n; i++) {
    const afx_msm_job& j = jobs[i];
    if (j.half_var) { half_of[i] = j.half_var; continue; }   // a part of a split job: summed, then encoded with its siblings
    if (!encoded_only(j)) continue;
    half_of[i] = j.leave_half ? j.out_var : new_var();   // leave_half: later terms on this base find the half where they look
    afx_compress_job cj = { half_of[i], j.out_enc, j.reject_identity, 0 };
    cjobs.push_back(cj);   // Assembler::compress rewrites the encoding's share of the operation counts
  }
  std::vector<char> done(n, 0);
  const uint32_t blocks_per_row = (count + AFX_BLOCK - 1) / AFX_BLOCK, resident = 2 * ctx->n_cu;
  // the narrow tables of `tr`: built (k_msm_tables, kind 2) and made affine - one inversion per item and grid row over all of the
  // launch's tables (kernels.hip k_table_affine); a small pass, which waits for the serial walk, spreads them over up to 8 rows
  auto emit_narrow_tables = [&](const std::vector<afx_table_job>& tr) {
    Launch tl;
    tl.kind = L_MSM_TABLES;
    tl.odd = cached_narrow ? 3 : 2;
    tl.njobs = (uint32_t)tr.size();
    tl.jobs_off = blob_alloc(sizeof(afx_table_job) * tr.size(), 16);
    memcpy(blob_.data() + tl.jobs_off, tr.data(), sizeof(afx_table_job) * tr.size());
    launches.push_back(tl);
    if (cached_narrow) { stats.field_mul += (uint64_t)tr.size() * AFX_SECVAR_STORED; return; }   // (2dT of each entry)
    Launch al;
    al.kind = L_TABLE_AFFINE;
    al.njobs = tl.njobs;
    al.jobs_off = blob_alloc(sizeof(afx_table_job) * tr.size(), 16);
    memcpy(blob_.data() + al.jobs_off, tr.data(), sizeof(afx_table_job) * tr.size());
    const uint32_t groups = ctx->walk_rows(count, tr.size(), small());
    add_walk_rows(al, groups > 1 ? (al.njobs + groups - 1) / groups : 0);
    launches.push_back(al);
    // per entry: the prefix product, 1/Z and the running inverse, x and y, the niels form (4); per row the inversion
    stats.field_mul += (uint64_t)tr.size() * AFX_SECVAR_STORED * 9 + (uint64_t)al.nrows * 11;
    stats.field_sq += (uint64_t)al.nrows * 254;
    stats.chain_mul += (uint64_t)al.nrows * AFX_CHAIN_INVERT_MUL; stats.chain_sq += (uint64_t)al.nrows * AFX_CHAIN_INVERT_SQ;
  };
  // A segmenting pass (Assembler::segments) KEEPS its narrow tables: slots from the bottom of the table workspace that no later stage
  // hands out again, one table per base for the whole pass, the ones this stage is the first to use built by one launch pair ahead
  // of its chains.  The proof's commitments then find the tables of the bases the first stage multiplied by.
  std::set<const int32_t*> fresh_kept;
  if (segmenting_) {
    std::vector<afx_table_job> tr;
    for (const afx_msm_job& j : jobs) {
      if (!j.narrow) continue;
      for (uint32_t t = 0; t < j.n_var; t++) {
        const int32_t* v = j.term[t].var;
        if (kept_tables_.count(v)) continue;
        const uint32_t slot = (uint32_t)kept_tables_.size();
        kept_tables_[v] = slot;
        fresh_kept.insert(v);
        afx_table_job tj = { v, slot, 0 };
        tr.push_back(tj);
      }
    }
    if (!tr.empty()) emit_narrow_tables(tr);
  }
  uint32_t dslot = 0, tslot = (uint32_t)kept_tables_.size();
  std::map<std::tuple<const int32_t*, bool, bool>, uint32_t> table_of;   // (base, odd multiples?, the short table of a narrow job?) -> table slot
  size_t left = n;
  static const int class_order[3] = { 2, 1, 0 };
  static const LaunchKind class_launch[3] = { L_MSM_FIXED, L_MSM_WINDOW, L_MSM_NAF };
  for (int guard = 0; left; guard++) {
    if (guard > (int)(3 * n + 3)) throw std::logic_error("msm chain cycle");
    const int kind = class_order[guard % 3];
    std::vector<size_t> rows;
    for (size_t i = 0; i < n; i++)
      if (!done[i] && kind_of(jobs[i]) == kind && (producer[i] < 0 || done[producer[i]])) rows.push_back(i);
    if (rows.empty()) continue;
    std::vector<uint32_t> c(n, 0);
    for (size_t i : rows) c[i] = cost(jobs[i]);
    rows = balanced_row_order(rows, c, blocks_per_row >= resident ? 1u : (resident + blocks_per_row / 2) / blocks_per_row);
    std::vector<afx_msm_job> out;
    std::vector<afx_table_job> table_rows[3];   // by kind of table (plan.h afx_table_job): windows, odd multiples, narrow
    for (size_t i : rows) {
      afx_msm_job j = jobs[i];
      j.half_var = half_of[i];
      if (j.leave_half) j.out_var = nullptr;   // one store: the half, through half_var
      j.digit_slot = dslot; dslot += j.n_terms;
      // one window table per (base, kind of multiples) of this launch list: constraints that share a base share its table
      // (a proof of encryption uses C_y_2 and C_y_2' in two constraints each, encryption.rs:197,204)
      for (uint32_t t = 0; t < j.n_var; t++) {
        const bool odd = t < j.n_uni;
        const uint32_t nstored = j.narrow ? AFX_SECVAR_STORED : AFX_TABLE_STORED;
        if (segmenting_ && j.narrow) {
          j.term[t].table_slot = kept_tables_.at(j.term[t].var);
          if (!fresh_kept.erase(j.term[t].var)) {   // counted per term above; this table exists already
            stats.table_additions -= nstored - 1;
            stats.field_mul -= 1 + (nstored - 1) * 8;
          }
          continue;
        }
        const std::tuple<const int32_t*, bool, bool> tk(j.term[t].var, odd, j.narrow != 0);
        auto hit = table_of.find(tk);
        if (hit == table_of.end()) {
          hit = table_of.emplace(tk, tslot++).first;
          afx_table_job tj = { j.term[t].var, hit->second, 0 };
          table_rows[odd ? 1 : (j.narrow ? 2 : 0)].push_back(tj);
        } else {
          stats.table_additions -= odd ? 7 : (nstored - 1);   // counted per term above; this one is shared
          stats.field_mul -= odd ? (4 + 1 + 1 + 7 * 9) : (1 + (nstored - 1) * (j.narrow ? 8 : 9));
          stats.field_sq -= odd ? 4 : 0;
        }
        j.term[t].table_slot = hit->second;
      }
      if (j.n_uni) {
        std::vector<uint32_t> sched;
        const std::vector<int8_t>& nd = naf_of[i];
        for (int b = j.top_bit; b >= 0; b--)
          for (uint32_t u = 0; u < j.n_uni; u++) {
            const int d = nd[256 * u + b];
            if (d) sched.push_back(((uint32_t)b << 16) | (u << 8) | (d < 0 ? 0x80u : 0u) | (uint32_t)(((d < 0 ? -d : d) - 1) >> 1));
          }
        sched.push_back(0xffffffffu);
        j.naf_sched = put(sched.data(), sched.size());
        secure_zero(sched.data(), 4 * sched.size());                     // digits of the issuer key
        secure_zero(naf_of[i].data(), naf_of[i].size());
        secure_zero(nafc_of[i].data(), nafc_of[i].size());
      }
      out.push_back(j);
    }
    for (size_t i : rows) { done[i] = 1; left--; }   // launched: consumers may go into any later launch
    for (int odd = 2; odd >= 0; odd--) {
      const std::vector<afx_table_job>& tr = table_rows[odd];
      if (tr.empty()) continue;
      if (odd == 2) { emit_narrow_tables(tr); continue; }
      Launch tl;
      tl.kind = L_MSM_TABLES;
      tl.odd = odd;
      tl.njobs = (uint32_t)tr.size();
      tl.jobs_off = blob_alloc(sizeof(afx_table_job) * tr.size(), 16);
      memcpy(blob_.data() + tl.jobs_off, tr.data(), sizeof(afx_table_job) * tr.size());
      launches.push_back(tl);
    }
    // a launch none of whose jobs encodes inside the kernel (every windowed launch of Issuer::verify: results that are only
    // encoded go through k_compress2x) runs the kernel compiled without the encoder (kernels.hip, k_msm<KIND, ENC>).  Splitting
    // a mixed launch in two was measured as well: -1.7 % on a 2^16-item show (fewer rows per launch), so mixed launches stay whole.
    Launch l;
    l.kind = class_launch[kind];
    l.encodes = 0;
    for (const afx_msm_job& j : out) if (j.out_enc && !j.half_var) l.encodes = 1;
    for (const afx_msm_job& j : out)
      for (uint32_t t = 0; t < j.n_terms; t++) if (j.term[t].secret) l.secret = 1;
    l.njobs = (uint32_t)out.size();
    // the kernels' form of a job: its terms in a side array (plan.h afx_msm_djob)
    std::vector<afx_msm_djob> dj(out.size());
    std::vector<size_t> terms_at(out.size());
    for (size_t i = 0; i < out.size(); i++) {
      const afx_msm_job& j = out[i];
      afx_msm_djob& d = dj[i];
      memset(&d, 0, sizeof d);
      d.n_terms = j.n_terms; d.n_var = j.n_var; d.n_uni = j.n_uni; d.top_bit = j.top_bit;
      d.naf_sched = j.naf_sched;
      terms_at[i] = (size_t)((const uint8_t*)put(j.term, j.n_terms) - blob_base_);
      if (j.n_terms) term_tables_.push_back({ terms_at[i], j.n_terms });
      d.addend = j.addend; d.addend_negate = j.addend_negate; d.reject_identity = j.reject_identity;
      d.out_enc = j.out_enc; d.out_var = j.out_var; d.half_var = j.half_var;
      d.digit_slot = j.digit_slot; d.narrow = (j.narrow && cached_narrow) ? 2u : j.narrow; d.leave_half = j.leave_half; d.wins = j.wins;
      if (d.narrow == 2) l.secret |= 2;   // (kernels.h afxk_msm: the launch takes the four-wave chains whatever its size)
    }
    l.jobs_off = blob_alloc(sizeof(afx_msm_djob) * dj.size(), 16);
    for (size_t i = 0; i < dj.size(); i++) dj[i].term_off = (int32_t)((int64_t)terms_at[i] - (int64_t)(l.jobs_off + sizeof(afx_msm_djob) * i));   // from the job itself
    memcpy(blob_.data() + l.jobs_off, dj.data(), sizeof(afx_msm_djob) * dj.size());
    launches.push_back(l);
  }
  max_digit_slots = std::max<size_t>(max_digit_slots, dslot);
  max_table_slots = std::max<size_t>(max_table_slots, tslot);
}
void Assembler::from_uniform(const uint8_t* wide, uint8_t* out_enc, int32_t* out_var) {
  // A small pass on an idle device: the two Elligator maps (a square-root chain each) run side by side in two grid rows, k_pointop
  // adds them, and the encoding (a third chain) joins the stage's other encodings in k_compress2x (Assembler::pointop): the call
  // waits for one chain here instead of three (a 1-item issue: 184 -> 67 us).  Same point, same bytes.
  if (small() && (uint64_t)ctx->row_waves(count) * 2 <= 2ull * 4 * ctx->n_cu) {
    // ... and the two maps wait for the statement's decodings - a third kind of square-root chain, independent of them - to share one
    // launch (Assembler::decode; whatever else comes first sends them off by themselves: flush_maps)
    int32_t *h1 = new_var(), *h2 = new_var();
    const afx_decode_job a = { wide, h1, 0, 1 }, b = { wide, h2, 0, 2 };
    pending_maps_.push_back(a); pending_maps_.push_back(b);
    afx_pointop_job s;
    memset(&s, 0, sizeof s);
    s.a = h1; s.b = h2; s.sa = 1; s.sb = 1; s.out = out_var; s.out_enc = out_enc;
    pending_map_sums_.push_back(s);
    return;
  }
  const afx_uniform_job j = { wide, out_enc, out_var };
  add_jobs(L_FROM_UNIFORM, std::vector<afx_uniform_job>(1, j));
}
void Assembler::reduce_wide(const uint8_t* wide, uint8_t* out) {
  const afx_reduce_job j = { wide, out };
  add_jobs(L_REDUCE_WIDE, std::vector<afx_reduce_job>(1, j));
}
void Assembler::copy(uint8_t* dst, const uint8_t* src, size_t bytes) {
  Launch l; l.kind = L_COPY; l.in = src; l.out = dst; l.bytes = bytes;
  launches.push_back(l);
}
void Assembler::finish(uint8_t* status_dev, uint8_t fail_code) {
  flush_maps();
  flush_encodings();   // (results no transcript reads; their rejections are flags k_finish folds into the status)
  const afx_finish_job j = { bad_, status_dev, count, fail_code };
  add_jobs(L_FINISH, std::vector<afx_finish_job>(1, j));
}
int Assembler::finish_plan(Plan& out, uint8_t* in_base, size_t in_bytes, uint8_t* out_base, size_t out_bytes) {
  if (!plan_error.empty()) { set_error(plan_error); return AFX_E_BAD_ARGS; }
  if (!pending_cjobs_.empty()) { set_error("an encoding queued by compress_also() was never launched"); return AFX_E_BAD_ARGS; }
  if (!pending_maps_.empty() || !pending_map_sums_.empty()) { set_error("internal: from_uniform's maps were never launched"); return AFX_E_BAD_ARGS; }
  if (!pending_reductions_.empty()) { set_error("internal: blindings squeezed out wide were never reduced"); return AFX_E_BAD_ARGS; }
  if (launches.empty() || launches[0].kind != L_FILL_BAD) { set_error("internal: plan without its opening launch"); return AFX_E_BAD_ARGS; }
  const afx_fill_job fj = { bad_, fail_all ? AFX_BAD_SHAPE : 0u, count };
  memcpy(blob_.data() + launches[0].jobs_off, &fj, sizeof fj);
  afx_pass P;
  memset(&P, 0, sizeof P);
  P.count = count;
  P.bad = bad_;
  P.table_ws = (int32_t*)ws_alloc(max_table_slots * (size_t)count * AFX_VAR_TABLE_DWORDS * sizeof(int32_t));
  P.digit_ws = (uint32_t*)ws_alloc(max_digit_slots * AFX_DIGIT_WORDS * (size_t)count * sizeof(uint32_t));
  memcpy(blob_.data(), &P, sizeof P);
  out.blob.swap(blob_);
  secure_zero(blob_.data(), blob_.size());
  blob_.clear();
  out.launches.swap(launches);
  out.ptr_tables.swap(ptr_tables_);
  out.term_tables.swap(term_tables_);
  out.blob_base = blob_base_;
  out.ws_base = ws_base_;
  out.ws_bytes = ((ws_off_ + 255) & ~size_t(255)) + 256;
  out.in_base = in_base; out.in_bytes = in_bytes;
  out.out_base = out_base; out.out_bytes = out_bytes;
  out.pass_off = 0;
  out.count = count;
  out.small = small();
  out.stats = stats;
  return AFX_OK;
}

// ------------------------------------------------------------------------------------------------
// Plan: relocation and execution
// ------------------------------------------------------------------------------------------------
Plan::~Plan() { secure_zero(blob.data(), blob.size()); }

namespace {
// a pointer moves with the range it points into (end inclusive: one-past-the-end pointers of empty rows move along)
struct Mover {
  struct R { uintptr_t lo, hi; intptr_t delta; } r[4];
  template <class T>
  void fix(T*& p) const {
    const uintptr_t v = (uintptr_t)p;
    if (!v) return;
    for (const R& x : r)
      if (x.hi > x.lo && v >= x.lo && v <= x.hi) { p = (T*)(v + x.delta); return; }
  }
  template <class T>
  void fix(const T*& p) const { T* q = const_cast<T*>(p); fix(q); p = q; }
};
size_t job_size(LaunchKind k) {
  switch (k) {
    case L_FILL_BAD: return sizeof(afx_fill_job);
    case L_DECODE: return sizeof(afx_decode_job);
    case L_SCCHECK: return sizeof(afx_sccheck_job);
    case L_POINTOP: return sizeof(afx_pointop_job);
    case L_SCALAROP: return sizeof(afx_scalarop_job);
    case L_MSM_WINDOW: case L_MSM_FIXED: case L_MSM_NAF: return sizeof(afx_msm_djob);
    case L_HASH: return sizeof(afx_hash_program);
    case L_FROM_UNIFORM: return sizeof(afx_uniform_job);
    case L_REDUCE_WIDE: return sizeof(afx_reduce_job);
    case L_FINISH: return sizeof(afx_finish_job);
    case L_MSM_TABLES: return sizeof(afx_table_job);
    case L_COMPRESS: return sizeof(afx_compress_job);
    case L_POINTSUM: return sizeof(afx_pointsum_job);
    case L_NEGENC: return sizeof(afx_negenc_job);
    case L_TABLE_AFFINE: return sizeof(afx_table_job);
    case L_POWERS: return sizeof(afx_powers_job);
    default: return 0;
  }
}
}  // namespace

// Every pointer field of every plan structure, by type.  Side tables that hold pointers themselves (a job's terms, a sum's parts,
// a transcript's field and output tables) are listed by the Assembler as it writes them (put_ptrs, term_tables_).
void Plan::relocate(uint8_t* nblob, uint8_t* nws, uint8_t* nin, uint8_t* nout) {
  if (nblob == blob_base && nws == ws_base && (nin == in_base || !in_bytes) && (nout == out_base || !out_bytes)) return;
  Mover m;
  m.r[0] = { (uintptr_t)blob_base, (uintptr_t)blob_base + blob.size(), (intptr_t)((uintptr_t)nblob - (uintptr_t)blob_base) };
  m.r[1] = { (uintptr_t)ws_base, (uintptr_t)ws_base + ws_bytes, (intptr_t)((uintptr_t)nws - (uintptr_t)ws_base) };
  m.r[2] = { (uintptr_t)in_base, (uintptr_t)in_base + in_bytes, (intptr_t)((uintptr_t)nin - (uintptr_t)in_base) };
  m.r[3] = { (uintptr_t)out_base, (uintptr_t)out_base + out_bytes, (intptr_t)((uintptr_t)nout - (uintptr_t)out_base) };
  // side tables first, from the plan's own lists (a table no job refers to - the statement ended in fail_all before its jobs were
  // emitted - moves like any other: equal plans stay equal bytes)
  for (const auto& t : ptr_tables) {
    uint8_t** tab = (uint8_t**)(blob.data() + t.first);
    for (uint32_t k = 0; k < t.second; k++) m.fix(tab[k]);
  }
  for (const auto& t : term_tables) {
    afx_msm_term* tab = (afx_msm_term*)(blob.data() + t.first);
    for (uint32_t k = 0; k < t.second; k++) { m.fix(tab[k].scalar); m.fix(tab[k].var); }
  }
  afx_pass* P = (afx_pass*)(blob.data() + pass_off);
  m.fix(P->bad); m.fix(P->table_ws); m.fix(P->digit_ws);
  for (Launch& l : launches) {
    uint8_t* J = blob.data() + l.jobs_off;
    switch (l.kind) {
      case L_FILL_BAD: for (uint32_t i = 0; i < l.njobs; i++) m.fix(((afx_fill_job*)J)[i].p); break;
      case L_DECODE: for (uint32_t i = 0; i < l.njobs; i++) { afx_decode_job& j = ((afx_decode_job*)J)[i]; m.fix(j.enc); m.fix(j.out); } break;
      case L_SCCHECK: for (uint32_t i = 0; i < l.njobs; i++) m.fix(((afx_sccheck_job*)J)[i].sc); break;
      case L_POINTOP:
        for (uint32_t i = 0; i < l.njobs; i++) { afx_pointop_job& j = ((afx_pointop_job*)J)[i]; m.fix(j.a); m.fix(j.b); m.fix(j.b_const); m.fix(j.out); m.fix(j.out_enc); }
        break;
      case L_SCALAROP: for (uint32_t i = 0; i < l.njobs; i++) { afx_scalarop_job& j = ((afx_scalarop_job*)J)[i]; m.fix(j.a); m.fix(j.b); m.fix(j.c); m.fix(j.out); } break;
      case L_MSM_WINDOW: case L_MSM_FIXED: case L_MSM_NAF:
        for (uint32_t i = 0; i < l.njobs; i++) {
          afx_msm_djob& j = ((afx_msm_djob*)J)[i];
          m.fix(j.naf_sched); m.fix(j.addend); m.fix(j.out_enc); m.fix(j.out_var); m.fix(j.half_var);
        }
        break;
      case L_MSM_TABLES: case L_TABLE_AFFINE: for (uint32_t i = 0; i < l.njobs; i++) m.fix(((afx_table_job*)J)[i].var); break;
      case L_COMPRESS: for (uint32_t i = 0; i < l.njobs; i++) { afx_compress_job& j = ((afx_compress_job*)J)[i]; m.fix(j.var); m.fix(j.out_enc); } break;
      case L_NEGENC: for (uint32_t i = 0; i < l.njobs; i++) { afx_negenc_job& j = ((afx_negenc_job*)J)[i]; m.fix(j.enc); m.fix(j.var); m.fix(j.out_enc); } break;
      case L_POINTSUM:
        for (uint32_t i = 0; i < l.njobs; i++) {
          afx_pointsum_job& j = ((afx_pointsum_job*)J)[i];
          m.fix(j.parts); m.fix(j.addend); m.fix(j.out_var); m.fix(j.half_var); m.fix(j.out_enc);
        }
        break;
      case L_HASH:
        for (uint32_t i = 0; i < l.njobs; i++) {
          afx_hash_program& j = ((afx_hash_program*)J)[i];
          m.fix(j.init_state); m.fix(j.load_state); m.fix(j.save_state); m.fix(j.records); m.fix(j.fields); m.fix(j.outs); m.fix(j.challenge); m.fix(j.trace);
        }
        break;
      case L_FROM_UNIFORM: for (uint32_t i = 0; i < l.njobs; i++) { afx_uniform_job& j = ((afx_uniform_job*)J)[i]; m.fix(j.wide); m.fix(j.out_enc); m.fix(j.out_var); } break;
      case L_REDUCE_WIDE: for (uint32_t i = 0; i < l.njobs; i++) { afx_reduce_job& j = ((afx_reduce_job*)J)[i]; m.fix(j.wide); m.fix(j.out); } break;
      case L_FINISH: for (uint32_t i = 0; i < l.njobs; i++) { afx_finish_job& j = ((afx_finish_job*)J)[i]; m.fix(j.bad); m.fix(j.status); } break;
      case L_POWERS:
        for (uint32_t i = 0; i < l.njobs; i++) { afx_powers_job& j = ((afx_powers_job*)J)[i]; m.fix(j.src); for (uint32_t k = 0; k < AFX_POWERS_MAX; k++) m.fix(j.out[k]); }
        break;
      case L_COPY: m.fix(l.in); m.fix(l.out); break;
      case L_KINDS: break;
    }
    if (l.nrows) {
      afx_walk_row* rows = (afx_walk_row*)(blob.data() + l.rows_off);
      for (uint32_t i = 0; i < l.nrows; i++) m.fix(rows[i].prefix_ws);
    }
  }
  blob_base = nblob; ws_base = nws;
  if (in_bytes) in_base = nin;
  if (out_bytes) out_base = nout;
}

bool Plan::same_as(const Plan& o, std::string* why) const {
  auto no = [&](const char* w) { if (why) *why = w; return false; };
  if (blob.size() != o.blob.size()) return no("blob sizes differ");
  if (ws_bytes != o.ws_bytes || count != o.count || small != o.small) return no("workspace size / count differ");
  if (launches.size() != o.launches.size()) return no("launch lists differ in length");
  for (size_t i = 0; i < launches.size(); i++) {
    const Launch &a = launches[i], &b = o.launches[i];
    if (a.kind != b.kind || a.jobs_off != b.jobs_off || a.njobs != b.njobs || a.rows_off != b.rows_off || a.nrows != b.nrows || a.in != b.in || a.out != b.out ||
        a.bytes != b.bytes || a.odd != b.odd || a.encodes != b.encodes || a.secret != b.secret) return no("a launch differs");
  }
  std::vector<std::pair<size_t, std::string>> regions;
  for (size_t i = 0; i < blob.size(); i++)
    if (blob[i] != o.blob[i]) {
      if (why) {
        *why = "blob byte " + std::to_string(i) + " differs (a pointer field the relocation does not know?)";
        if ((i & ~size_t(7)) + 8 <= blob.size()) {
          uint64_t a, b;
          memcpy(&a, blob.data() + (i & ~size_t(7)), 8); memcpy(&b, o.blob.data() + (i & ~size_t(7)), 8);
          char buf[80];
          snprintf(buf, sizeof buf, " [word %016llx vs %016llx]", (unsigned long long)a, (unsigned long long)b);
          *why += buf;
        }
        auto in = [&](const void* dev, size_t bytes) {
          const uintptr_t v = (uintptr_t)dev, b = (uintptr_t)blob_base;
          if (dev && v >= b) regions.push_back({ (size_t)(v - b), "side table of " + std::to_string(bytes) + " bytes" });
          return dev && v >= b && i >= v - b && i < v - b + bytes;
        };
        for (const Launch& l : launches) {
          const size_t js = job_size(l.kind);
          if (js && i >= l.jobs_off && i < l.jobs_off + js * l.njobs)
            *why += ": launch kind " + std::to_string((int)l.kind) + ", job " + std::to_string((i - l.jobs_off) / js) + ", byte " + std::to_string((i - l.jobs_off) % js) + " of its struct";
          if (l.nrows && i >= l.rows_off && i < l.rows_off + sizeof(afx_walk_row) * l.nrows) *why += ": walk rows of launch kind " + std::to_string((int)l.kind);
          regions.push_back({ l.jobs_off, "jobs of launch kind " + std::to_string((int)l.kind) + " x" + std::to_string(l.njobs) });
          const uint8_t* J = blob.data() + l.jobs_off;
          for (uint32_t k = 0; k < l.njobs; k++) {
            if (l.kind == L_MSM_WINDOW || l.kind == L_MSM_FIXED || l.kind == L_MSM_NAF) {
              const afx_msm_djob& j = ((const afx_msm_djob*)J)[k];
              if (in(blob_base + (((const uint8_t*)&j - blob.data()) + j.term_off), sizeof(afx_msm_term) * j.n_terms)) *why += ": terms of msm job " + std::to_string(k) + " (launch kind " + std::to_string((int)l.kind) + ")";
            } else if (l.kind == L_POINTSUM) {
              const afx_pointsum_job& j = ((const afx_pointsum_job*)J)[k];
              if (in(j.parts, 8 * j.n_parts)) *why += ": parts of pointsum job " + std::to_string(k);
            } else if (l.kind == L_HASH) {
              const afx_hash_program& j = ((const afx_hash_program*)J)[k];
              if (in(j.fields, 8 * j.n_fields)) *why += ": fields of hash program " + std::to_string(k);
              if (in(j.outs, 8 * j.n_outs)) *why += ": outs of hash program " + std::to_string(k);
              if (in(j.records, sizeof(afx_hash_record) * j.n_records)) *why += ": records of hash program " + std::to_string(k);
            }
          }
        }
      }
      if (why) {
        std::sort(regions.begin(), regions.end());
        for (size_t r = 0; r < regions.size(); r++)
          if (regions[r].first <= i && (r + 1 == regions.size() || regions[r + 1].first > i)) *why += " {last known region before it: @" + std::to_string(regions[r].first) + " " + regions[r].second + "}";
      }
      return false;
    }
  return true;
}

static int grow_blob(afx_ctx::Lane& L, int slot, size_t bytes) {
  if (bytes <= L.blob_dev[slot].cap && bytes <= L.blob_host_cap[slot]) return AFX_OK;
  const size_t want = (bytes + (size_t(1) << 20) - 1) & ~((size_t(1) << 20) - 1);
  int rc = L.blob_dev[slot].ensure(want);   // waits for the device before it frees the old buffer
  if (rc) return rc;
  if (want > L.blob_host_cap[slot]) {
    if (L.blob_host[slot]) { memset(L.blob_host[slot], 0, L.blob_host_cap[slot]); (void)hipHostFree(L.blob_host[slot]); L.blob_host[slot] = nullptr; L.blob_host_cap[slot] = 0; }
    void* p = nullptr;
    AFX_HIP(hipHostMalloc(&p, want, hipHostMallocDefault));
    L.blob_host[slot] = p;
    L.blob_host_cap[slot] = want;
  }
  return AFX_OK;
}

int run_plans(afx_ctx* ctx, int lane, Plan* const* plans, size_t n) {
  if (n == 0) return AFX_OK;
  afx_ctx::Lane& L = ctx->lane[lane];
  ctx->last_stats = plans[0]->stats;
  // where every plan's blob and workspace go
  std::vector<size_t> boff(n), woff(n);
  size_t blob_total = 0, ws_total = 0;
  for (size_t i = 0; i < n; i++) {
    boff[i] = (blob_total + 255) & ~size_t(255);
    blob_total = boff[i] + plans[i]->blob.size();
    woff[i] = (ws_total + 255) & ~size_t(255);
    ws_total = woff[i] + plans[i]->ws_bytes;
  }
  // Merged launch list.  Every plan's launches run in the plan's own order; launches of the same kernel at the heads of several
  // plans' lists become one launch over all their rows.  Greedy: take the kernel at the head of the first unfinished plan and
  // every other plan whose head is that kernel.
  struct Part { uint32_t plan, launch; };
  struct Merged { LaunchKind kind; int odd; std::vector<Part> parts; size_t rows_off = 0; uint32_t nrows = 0; };
  std::vector<Merged> sched;
  if (n > 1) {
    std::vector<size_t> head(n, 0);
    for (;;) {
      size_t lead = n;
      for (size_t i = 0; i < n; i++) if (head[i] < plans[i]->launches.size()) { lead = i; break; }
      if (lead == n) break;
      const Launch& h = plans[lead]->launches[head[lead]];
      Merged mg;
      mg.kind = h.kind; mg.odd = h.odd;
      for (size_t i = lead; i < n; i++) {
        if (head[i] >= plans[i]->launches.size()) continue;
        const Launch& c = plans[i]->launches[head[i]];
        const bool same = c.kind == h.kind && (h.kind != L_MSM_TABLES || c.odd == h.odd) && (h.kind != L_COPY || i == lead);
        if (!same) continue;
        if ((h.kind == L_POINTSUM || h.kind == L_DECODE) && c.odd) mg.odd = 1;   // (some job of the merged launch has many parts / is an Elligator map)
        mg.parts.push_back({ (uint32_t)i, (uint32_t)head[i] });
        head[i]++;
      }
      sched.push_back(std::move(mg));
    }
    // the merged launches' row tables and the pass table, behind the plans' blobs
    blob_total = (blob_total + 255) & ~size_t(255);
    for (Merged& mg : sched) {
      if (mg.kind == L_COPY) continue;
      const bool walk = walks(mg.kind);
      uint32_t rows = 0;
      for (const Part& p : mg.parts) { const Launch& l = plans[p.plan]->launches[p.launch]; rows += walk ? l.nrows : l.njobs; }
      mg.nrows = rows;
      mg.rows_off = blob_total;
      blob_total += ((walk ? sizeof(afx_walk_row) : sizeof(afx_row)) * (size_t)rows + 15) & ~size_t(15);
    }
  }
  const size_t passes_off = (blob_total + 15) & ~size_t(15);
  if (n > 1) blob_total = passes_off + sizeof(afx_pass) * n;
  if (blob_total > 0xffffffffull) { set_error("plan blob too large"); return AFX_E_BAD_ARGS; }
  // place and relocate
  const int slot = L.blob_next;
  AFX_HIP(hipEventSynchronize(L.blob_event[slot]));   // the pinned mirror of this slot may still be the source of a copy from two calls ago
  int rc = grow_blob(L, slot, blob_total);
  if (rc) return rc;
  if ((rc = L.ws.ensure(ws_total))) return rc;
  uint8_t* const bdev = (uint8_t*)L.blob_dev[slot].p;
  uint8_t* const bhost = (uint8_t*)L.blob_host[slot];
  for (size_t i = 0; i < n; i++) {
    plans[i]->relocate(bdev + boff[i], (uint8_t*)L.ws.p + woff[i], plans[i]->in_base, plans[i]->out_base);
    memcpy(bhost + boff[i], plans[i]->blob.data(), plans[i]->blob.size());
  }
  if (n > 1) {
    for (const Merged& mg : sched) {
      if (mg.kind == L_COPY) continue;
      const bool walk = walks(mg.kind);
      size_t w = mg.rows_off;
      for (const Part& p : mg.parts) {
        const Plan& pl = *plans[p.plan];
        const Launch& l = pl.launches[p.launch];
        if (walk) {
          const afx_walk_row* src = (const afx_walk_row*)(pl.blob.data() + l.rows_off);
          for (uint32_t r = 0; r < l.nrows; r++) {
            afx_walk_row row = src[r];
            row.job_off = (uint32_t)(boff[p.plan] + l.jobs_off + row.job_off);
            row.pass = p.plan;
            memcpy(bhost + w, &row, sizeof row);
            w += sizeof row;
          }
        } else {
          const size_t js = job_size(mg.kind);
          for (uint32_t r = 0; r < l.njobs; r++) {
            const afx_row row = { p.plan, (uint32_t)(boff[p.plan] + l.jobs_off + r * js) };
            memcpy(bhost + w, &row, sizeof row);
            w += sizeof row;
          }
        }
      }
    }
    for (size_t i = 0; i < n; i++) memcpy(bhost + passes_off + sizeof(afx_pass) * i, plans[i]->blob.data() + plans[i]->pass_off, sizeof(afx_pass));
  }
  hipStream_t s = L.stream;
  AFX_HIP(hipMemcpyAsync(bdev, bhost, blob_total, hipMemcpyHostToDevice, s));
  AFX_HIP(hipEventRecord(L.blob_event[slot], s));
  L.blob_next ^= 1;

  // one launch: `jobs` / `rows` / `passes` as the kernels take them (kernels.h)
  auto launch = [&](LaunchKind kind, int odd, int encodes, int secret, const uint8_t* jobs, uint32_t nrows, const void* rows, const afx_pass* passes,
                    const afx_pass* pass_host, uint32_t max_count, bool coop, const Launch* copy) -> int {
    afx_ctx::TimedLaunch tl = { (int)kind, nullptr, nullptr };
    if (ctx->timing) {
      for (hipEvent_t* e : { &tl.start, &tl.stop }) {
        if (ctx->event_pool.empty()) AFX_HIP(hipEventCreate(e));
        else { *e = ctx->event_pool.back(); ctx->event_pool.pop_back(); }
      }
      AFX_HIP(hipEventRecord(tl.start, s));
    }
    const afx_row* rw = (const afx_row*)rows;
    switch (kind) {
      case L_FILL_BAD: AFX_HIP(afxk_fill_u32(s, (const afx_fill_job*)jobs, nrows, rw, max_count)); break;
      case L_DECODE: AFX_HIP(afxk_decode(s, (const afx_decode_job*)jobs, nrows, rw, passes, max_count, odd)); break;
      case L_SCCHECK: AFX_HIP(afxk_sccheck(s, (const afx_sccheck_job*)jobs, nrows, rw, passes, max_count)); break;
      case L_POINTOP: AFX_HIP(afxk_pointop(s, (const afx_pointop_job*)jobs, nrows, rw, passes, max_count)); break;
      case L_SCALAROP: AFX_HIP(afxk_scalarop(s, (const afx_scalarop_job*)jobs, nrows, rw, passes, max_count)); break;
      case L_COMPRESS: AFX_HIP(afxk_compress2x(s, (const afx_compress_job*)jobs, (const afx_walk_row*)rows, nrows, passes, max_count)); break;
      case L_NEGENC: AFX_HIP(afxk_negenc(s, (const afx_negenc_job*)jobs, (const afx_walk_row*)rows, nrows, passes, max_count)); break;
      case L_TABLE_AFFINE: AFX_HIP(afxk_table_affine(s, (const afx_table_job*)jobs, (const afx_walk_row*)rows, nrows, passes, max_count)); break;
      case L_POINTSUM: AFX_HIP(afxk_pointsum(s, (const afx_pointsum_job*)jobs, nrows, rw, passes, max_count, odd, ctx->variants)); break;
      case L_POWERS: AFX_HIP(afxk_powers(s, (const afx_powers_job*)jobs, nrows, rw, passes, max_count)); break;
      case L_MSM_TABLES: AFX_HIP(afxk_msm_tables(s, odd, (const afx_table_job*)jobs, nrows, rw, passes, max_count)); break;
      case L_MSM_FIXED: case L_MSM_WINDOW: case L_MSM_NAF: {
        // pipelined lanes: the heavy kernel of one lane never runs beside the other lane's (only the light kernels
        // overlap it), which keeps per-launch timings meaningful and the VALU free of two competing table working sets
        afx_ctx::Lane& other = ctx->lane[lane < 2 ? lane ^ 1 : 0];   // (pipelined calls run on lanes 0 and 1)
        if (ctx->pipelining && other.msm_recorded) AFX_HIP(hipStreamWaitEvent(s, other.msm_done, 0));
        AFX_HIP(afxk_msm(s, kind == L_MSM_FIXED ? 0 : kind == L_MSM_WINDOW ? 1 : 2, encodes, secret, (const afx_msm_djob*)jobs, nrows,
                         (const int32_t*)ctx->d_pos_tables.p, (const int32_t*)ctx->d_sec_tables.p, rw, passes, pass_host, max_count,
                         (ctx->timing && kind == L_MSM_WINDOW) ? ctx->clock_probe() : nullptr, ctx->variants));
        if (ctx->pipelining) { AFX_HIP(hipEventRecord(L.msm_done, s)); L.msm_recorded = true; }
        break;
      }
      case L_HASH:
        // small passes: the permutation spread over 32 lanes per item (kernels.hip k_hash_coop), while the device has lanes to spare
        if (coop) AFX_HIP(afxk_hash_coop(s, (const afx_hash_program*)jobs, nrows, rw, passes, max_count, ctx->variants));
        else AFX_HIP(afxk_hash(s, (const afx_hash_program*)jobs, nrows, rw, passes, max_count));
        break;
      case L_FROM_UNIFORM: AFX_HIP(afxk_from_uniform_jobs(s, (const afx_uniform_job*)jobs, nrows, rw, passes, max_count)); break;
      case L_REDUCE_WIDE: AFX_HIP(afxk_reduce_wide_jobs(s, (const afx_reduce_job*)jobs, nrows, rw, passes, max_count)); break;
      case L_COPY: AFX_HIP(hipMemcpyAsync(copy->out, copy->in, copy->bytes, hipMemcpyDeviceToDevice, s)); break;
      case L_FINISH: AFX_HIP(afxk_finish(s, (const afx_finish_job*)jobs, nrows, rw, max_count)); break;
      case L_KINDS: break;
    }
    if (ctx->timing) {
      AFX_HIP(hipEventRecord(tl.stop, s));
      ctx->timed.push_back(tl);
    }
    return AFX_OK;
  };
  if (n == 1) {
    const Plan& pl = *plans[0];
    const afx_pass* passes = (const afx_pass*)(bdev + boff[0] + pl.pass_off);
    for (const Launch& l : pl.launches) {
      const bool walk = walks(l.kind);
      const bool coop = l.kind == L_HASH && pl.small && (uint64_t)pl.count * l.njobs <= AFX_HASH_COOP_GROUPS;
      if ((rc = launch(l.kind, l.odd, l.encodes, l.secret, bdev + boff[0] + l.jobs_off, walk ? l.nrows : l.njobs, walk ? (const void*)(bdev + boff[0] + l.rows_off) : nullptr,
                       passes, (const afx_pass*)(pl.blob.data() + pl.pass_off), pl.count, coop, &l)))
        return rc;
    }
    return AFX_OK;
  }
  const afx_pass* passes = (const afx_pass*)(bdev + passes_off);
  for (const Merged& mg : sched) {
    int encodes = 0, secret = 0;
    uint32_t max_count = 0;
    uint64_t hash_groups = 0;
    bool all_small = true;
    for (const Part& p : mg.parts) {
      const Plan& pl = *plans[p.plan];
      const Launch& l = pl.launches[p.launch];
      encodes |= l.encodes; secret |= l.secret;
      max_count = std::max(max_count, pl.count);
      hash_groups += (uint64_t)pl.count * l.njobs;
      all_small = all_small && pl.small;
    }
    const Launch* first = &plans[mg.parts[0].plan]->launches[mg.parts[0].launch];
    // jobs = the blob itself: every row names its job by byte offset (plan.h afx_row)
    if ((rc = launch(mg.kind, mg.odd, encodes, secret, bdev, mg.nrows, bdev + mg.rows_off, passes, nullptr, max_count,
                     mg.kind == L_HASH && all_small && hash_groups <= AFX_HASH_COOP_GROUPS, first)))
      return rc;
  }
  return AFX_OK;
}

// ------------------------------------------------------------------------------------------------
// SchnorrBuilder
// ------------------------------------------------------------------------------------------------
SchnorrBuilder::SchnorrBuilder(Assembler& as, const char* transcript_label, const char* proof_label)
    : as_(as), sim_(transcript_label) {
  // TranscriptProtocol::domain_sep (zkp toolbox)
  sim_.append_message_const("dom-sep", (const uint8_t*)"schnorrzkp/1.0/ristretto255", 27);
  sim_.append_message_const("dom-sep", (const uint8_t*)proof_label, strlen(proof_label));
}
int SchnorrBuilder::field_of(const uint8_t* dev) {
  for (size_t i = 0; i < fields_.size(); i++)
    if (fields_[i] == dev) return (int)i;
  fields_.push_back(dev);
  return (int)fields_.size() - 1;
}
int SchnorrBuilder::allocate_scalar(const char* label, const ScalarVar& v) {
  sim_.append_message_const("scvar", (const uint8_t*)label, strlen(label));
  scalars_.push_back(v);
  return (int)scalars_.size() - 1;
}
int SchnorrBuilder::allocate_point(const char* label, const PointVar& p) {
  sim_.append_message_const("ptvar", (const uint8_t*)label, strlen(label));
  if (p.is_const) {
    const Enc& e = p.neg ? as_.ctx->gen_neg_enc[p.gen] : as_.ctx->gen_enc[p.gen];
    // validate_and_append_point_var rejects the identity encoding: with a constant point, for every item
    bool nz = false;
    for (uint8_t b : e) nz |= (b != 0);
    if (!nz) as_.fail_all = true;
    sim_.append_message_const("val", e.data(), 32);
  } else {
    sim_.append_message_hole32("val", field_of(p.enc_dev));
  }
  points_.push_back(p);
  point_labels_.push_back(label);
  return (int)points_.size() - 1;
}
void SchnorrBuilder::constrain(int lhs, const std::vector<std::pair<int, int>>& terms) { constraints_.push_back({ lhs, terms }); }

afx_msm_term SchnorrBuilder::term_for(const uint8_t* scalar, uint32_t stride, const PointVar& p, bool negate, std::vector<afx_scalarop_job>* pre_ops) {
  afx_msm_term t;
  memset(&t, 0, sizeof t);
  if (!p.is_const && p.has_alt && pre_ops) {
    // s * (m * G) = (s*m) * G: a fixed-base term (8-bit windows, no per-lane table) instead of a variable-base one
    uint8_t* prod = as_.new_enc();
    afx_scalarop_job o;
    memset(&o, 0, sizeof o);
    o.a = scalar; o.a_stride = stride; o.b = p.alt_scalar; o.b_stride = 32; o.out = prod;
    pre_ops->push_back(o);
    t.scalar = prod; t.scalar_stride = 32; t.fixed_idx = (int32_t)p.alt_gen; t.var = nullptr; t.negate = negate ? 1u : 0u;
    return t;
  }
  t.scalar = scalar;
  t.scalar_stride = stride;
  if (p.is_const) { t.fixed_idx = (int32_t)p.gen; t.var = nullptr; t.negate = (negate != p.neg) ? 1u : 0u; }
  else {
    if (!p.var) throw std::logic_error("a term needs the coordinates of a point that was only encoded");
    t.fixed_idx = -1; t.var = p.var; t.negate = (negate != p.var_negated) ? 1u : 0u;
  }
  return t;
}
static void order_terms(afx_msm_job& j, const std::vector<afx_msm_term>& terms) {
  if (terms.size() > AFX_MSM_MAX_TERMS) throw std::length_error("too many terms in one multiscalar job");
  j.n_terms = (uint32_t)terms.size();
  j.n_var = 0;
  j.chain_to = -1;
  uint32_t k = 0;
  for (const afx_msm_term& t : terms) if (t.fixed_idx < 0) { j.term[k++] = t; j.n_var++; }
  for (const afx_msm_term& t : terms) if (t.fixed_idx >= 0) j.term[k++] = t;
}
afx_hash_program SchnorrBuilder::make_program(const StrobeSim& sim) { return make_hash_program(as_, sim, fields_); }
afx_hash_program make_hash_program(Assembler& as_, const StrobeSim& sim, const std::vector<const uint8_t*>& fields_) {
  // The transcript's leading all-constant blocks are applied on the host, once per distinct prefix and context (SURVEY.md section 7
  // step 4): the device starts from the folded state.  A prover's prefix can hold key material (witnesses rekey the rng clone,
  // zkp's TranscriptRngBuilder): the cache is wiped with the context, the state travels in the plan blob like the constants did.
  const size_t k = sim.constant_prefix();
  uint64_t folded[25];
  if (k == 0) memcpy(folded, sim.init_state, sizeof folded);
  else {
    std::string key;
    sim.prefix_key(k, key);
    afx_ctx* c = as_.ctx;
    auto hit = c->folded_states.find(key);
    if (hit == c->folded_states.end()) {
      sim.fold_prefix(k, folded);
      std::array<uint64_t, 25> a;
      memcpy(a.data(), folded, sizeof folded);
      if (c->folded_states.size() < 1024) c->folded_states.emplace(key, a);
      secure_zero(a.data(), sizeof folded);
    } else memcpy(folded, hit->second.data(), sizeof folded);
    secure_zero(&key[0], key.size());
  }
  std::vector<afx_hash_record> recs;
  sim.emit(recs, k);
  afx_hash_program p;
  memset(&p, 0, sizeof p);
  p.init_state = as_.put(folded, 25);
  secure_zero(folded, sizeof folded);
  p.n_records = (uint32_t)recs.size();
  p.records = as_.put(recs.data(), recs.size());
  p.fields = as_.put_ptrs(fields_.data(), fields_.size());
  p.n_fields = (uint32_t)fields_.size();
  return p;
}

void SchnorrBuilder::verify_compact(const uint8_t* challenge_dev, uint32_t trace_row, size_t total, size_t off, std::vector<afx_msm_job>& msm_out, std::vector<afx_hash_program>& hash_out,
                                    std::vector<afx_scalarop_job>* pre_ops, std::vector<afx_scalarop_job>* expand_ops) {
  // R_j = sum resp[s] * P  - c * LHS, appended as "blindcom"
  for (auto& cn : constraints_) {
    std::vector<afx_msm_term> terms;
    for (auto& sp : cn.second) terms.push_back(term_for(scalars_[sp.first].dev, scalars_[sp.first].stride, points_[sp.second], false, pre_ops));
    const PointVar& lhs = points_[cn.first];
    if (expand_ops && !lhs.parts.empty()) {
      // -c * LHS over the parts of LHS: -c * (+-coef * base) = -+(c * coef) * base, so no term needs LHS's own coordinates
      for (const PointVar::Part& pt : lhs.parts) {
        afx_msm_term t;
        memset(&t, 0, sizeof t);
        if (pt.coef) {
          uint8_t* prod = as_.new_enc();
          afx_scalarop_job o;
          memset(&o, 0, sizeof o);
          o.a = pt.coef; o.a_stride = pt.coef_stride; o.b = challenge_dev; o.b_stride = 32; o.out = prod;
          expand_ops->push_back(o);
          t.scalar = prod;
        } else {
          t.scalar = challenge_dev;
        }
        t.scalar_stride = 32; t.var = pt.var; t.fixed_idx = pt.var ? -1 : pt.fixed; t.negate = pt.neg ? 0u : 1u;
        terms.push_back(t);
      }
    } else {
      terms.push_back(term_for(challenge_dev, 32, lhs, true, pre_ops));
    }
    afx_msm_job j;
    memset(&j, 0, sizeof j);
    order_terms(j, terms);
    j.out_enc = as_.new_enc();
    msm_out.push_back(j);
    sim_.append_message_const("blindcom", (const uint8_t*)point_labels_[cn.first].c_str(), point_labels_[cn.first].size());
    sim_.append_message_hole32("val", field_of(j.out_enc));
  }
  sim_.challenge64("chal", AFX_SQ_CHALLENGE_COMPARE, 0);
  afx_hash_program p = make_program(sim_);
  p.challenge = challenge_dev;
  afx_ctx* c = as_.ctx;
  if (c->trace) {
    if (total > c->trace_count || trace_row >= c->trace_rows) as_.plan_error = "challenge trace array too small for this call";
    else p.trace = c->trace + ((size_t)trace_row * c->trace_count + off) * 32;
  }
  hash_out.push_back(p);
}

void SchnorrBuilder::prove_compact(const uint8_t* rng_seed_dev, uint8_t* challenge_out, uint8_t* responses_out, size_t response_row_stride,
                                   std::vector<afx_hash_program>& rng_hash, std::vector<afx_msm_job>& msm_out,
                                   std::vector<afx_hash_program>& chal_hash, std::vector<afx_scalarop_job>& resp_ops,
                                   std::vector<afx_scalarop_job>* pre_ops) {
  const size_t ns = scalars_.size();
  // TranscriptRngBuilder: clone, rekey with every witness, finalize with the external 32 bytes
  StrobeSim rng = sim_;
  const uint8_t len32[4] = { 32, 0, 0, 0 }, len64[4] = { 64, 0, 0, 0 };
  for (size_t i = 0; i < ns; i++) {
    rng.meta_ad_const((const uint8_t*)"", 0, false);
    rng.meta_ad_const(len32, 4, true);
    if (scalars_[i].stride == 0) rng.key_const(scalars_[i].host.data(), 32);
    else rng.key_hole32(field_of(scalars_[i].dev));
  }
  rng.meta_ad_const((const uint8_t*)"rng", 3, false);
  rng.key_hole32(field_of(rng_seed_dev));
  std::vector<uint8_t*> blind(ns), wide(ns, nullptr);
  for (size_t i = 0; i < ns; i++) {
    blind[i] = as_.new_enc();
    rng.meta_ad_const(len64, 4, false);
    // a small pass: the 64 bytes leave the sponge as they are and ONE launch behind the hash reduces all the blindings side by side
    // (Assembler::hash) - the reduction of each between two permutations, on the sponge's lane, was 2.7 us of an issuance's 21
    if (as_.small()) {
      wide[i] = as_.new_wide();
      rng.prf64(AFX_SQ_WIDE_OUT, (uint32_t)i);
      const afx_reduce_job rj = { wide[i], blind[i] };
      as_.pending_reductions_.push_back(rj);
    } else {
      rng.prf64(AFX_SQ_SCALAR_OUT, (uint32_t)i);
    }
  }
  // the rng's last prf closes on a completed record, so emit() sees everything
  // commitments R_j = sum blind[s] * P
  for (auto& cn : constraints_) {
    std::vector<afx_msm_term> terms;
    for (auto& sp : cn.second) {
      const PointVar& P = points_[sp.second];
      // A segmenting pass (Assembler::segments): a base the prover knows as a sum of parts - a point an earlier stage of this pass
      // computed from its inputs (t*U, C_x_0 = z*G_x_0 + U, ...) - is multiplied part by part, blind * P = sum (blind * coef_k) * base_k:
      // the terms land on the pass's INPUT points, whose powers and tables the first stage made, and on generators.  Same group
      // element, hence the same commitment bytes.
      if (!P.is_const && !P.parts.empty() && pre_ops && as_.segments() > 1) {
        for (const PointVar::Part& pt : P.parts) {
          afx_msm_term t;
          memset(&t, 0, sizeof t);
          if (pt.coef) {
            uint8_t* prod = as_.new_enc();
            afx_scalarop_job o;
            memset(&o, 0, sizeof o);
            o.a = blind[sp.first]; o.a_stride = 32; o.b = pt.coef; o.b_stride = pt.coef_stride; o.out = prod;
            pre_ops->push_back(o);
            t.scalar = prod;
          } else {
            t.scalar = blind[sp.first];
          }
          t.scalar_stride = 32; t.var = pt.var; t.fixed_idx = pt.var ? -1 : pt.fixed; t.negate = pt.neg ? 1u : 0u;
          terms.push_back(t);
        }
        continue;
      }
      terms.push_back(term_for(blind[sp.first], 32, P, false, pre_ops));
    }
    afx_msm_job j;
    memset(&j, 0, sizeof j);
    order_terms(j, terms);
    j.out_enc = as_.new_enc();
    msm_out.push_back(j);
    sim_.append_message_const("blindcom", (const uint8_t*)point_labels_[cn.first].c_str(), point_labels_[cn.first].size());
    sim_.append_message_hole32("val", field_of(j.out_enc));
  }
  sim_.challenge64("chal", AFX_SQ_SCALAR_OUT, 0);
  // k_hash_coop requests a record's field bytes while the record before it permutes (kernels.hip): nothing a program squeezes out may
  // be a field of the same table
  for (const uint8_t* f : fields_) {
    if (f == challenge_out) throw std::logic_error("a transcript reads its own challenge");
    for (const uint8_t* b : blind) if (f == b) throw std::logic_error("a transcript reads a blinding it squeezed");
  }
  // programs are made after all field_of() calls so both share the final field table
  afx_hash_program pr = make_program(rng);
  pr.outs = as_.put_ptrs(as_.small() ? wide.data() : blind.data(), blind.size());
  pr.n_outs = (uint32_t)blind.size();
  rng_hash.push_back(pr);
  afx_hash_program pc = make_program(sim_);
  uint8_t* couts[1] = { challenge_out };
  pc.outs = as_.put_ptrs(couts, 1);
  pc.n_outs = 1;
  chal_hash.push_back(pc);
  // responses s*c + b
  for (size_t i = 0; i < ns; i++) {
    afx_scalarop_job o;
    memset(&o, 0, sizeof o);
    o.a = scalars_[i].dev; o.a_stride = scalars_[i].stride;
    o.b = challenge_out; o.b_stride = 32;
    o.c = blind[i]; o.c_stride = 32;
    o.out = responses_out + i * response_row_stride;
    resp_ops.push_back(o);
  }
}

}  // namespace afx
