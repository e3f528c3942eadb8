// HIP kernels of the aeonflux batch NIZK engine for gfx950 (MI355X).  One proof (or one constraint of one
// proof) per lane; all control flow is uniform across a wave because the statement shape is batch-uniform.
//
// Kernel inventory (SURVEY.md §8a row K* -> kernel):
//   k_setup_generators, k_setup_posbase, k_setup_postables   decompress the generators once, build their positional tables
//   k_decode            CompressedRistretto::decompress, coalesced 32-byte loads          (P4 ii, E1)
//   k_sccheck           Scalar canonicity
//   k_pointop           +-P +-Q (+ compress): C_V - W, C_y+M, C_y_1-E2, -E1               (P1, E1)
//   k_scalarop          a*b+c mod l: y_i*m_i, -t*z, responses s*c+b                       (P1, S1, I2)
//   k_msm_tables<kind>  per-lane window tables of the variable bases of a launch list (multiples 1..8, the odd multiples 1..15
//                       for width-5 NAF terms, or the two multiples of a narrow job), one (base) per grid row, shared by the
//                       jobs that use it
//   k_msm<KIND>         R = sum s_k P_k (+-addend) -> compress, one job class per kernel: MSM_FIXED (positional tables only),
//                       MSM_WINDOW (per-item scalars: signed 4-bit windows over the per-lane tables, shared doublings),
//                       MSM_NAF (batch-constant scalars - the issuer key - as a wave-uniform width-5 NAF schedule) (P1, P4 iii, I1)
//   k_hash              STROBE-128/merlin transcript over Keccak-f[1600] driven by a precompiled byte
//                       schedule; squeezes challenges / blinding factors                   (P2, P4 iv-v)
//   k_finish            per-item status byte
//   k_from_uniform, k_reduce_wide   RistrettoPoint::from_uniform_bytes, Scalar::from_bytes_mod_order_wide
#include <hip/hip_runtime.h>
#include "ge.cuh"
#include "keccak.cuh"
#include "plan.h"
#include "sc.cuh"

// ---------------------------------------------------------------------------------------------
// memory helpers
// ---------------------------------------------------------------------------------------------
AFX_DEV void enc_load(uint32_t w[8], const uint8_t* arr, uint32_t item) {
  const uint4* p = reinterpret_cast<const uint4*>(arr + 32ull * item);
  const uint4 a = p[0], b = p[1];
  w[0] = a.x; w[1] = a.y; w[2] = a.z; w[3] = a.w; w[4] = b.x; w[5] = b.y; w[6] = b.z; w[7] = b.w;
}
AFX_DEV void enc_store(uint8_t* arr, uint32_t item, const uint32_t w[8]) {
  uint4* p = reinterpret_cast<uint4*>(arr + 32ull * item);
  p[0] = make_uint4(w[0], w[1], w[2], w[3]);
  p[1] = make_uint4(w[4], w[5], w[6], w[7]);
}
AFX_DEV sc sc_load_item(const uint8_t* arr, uint32_t stride, uint32_t item) {
  uint32_t w[8];
  enc_load(w, arr, stride ? item : 0);
  sc r;
#pragma unroll
  for (int i = 0; i < 8; i++) r.v[i] = w[i];
  return r;
}
AFX_DEV fe fe_load_soa(const int32_t* base, uint32_t c, uint32_t count, uint32_t item) {
  fe r;
#pragma unroll
  for (int l = 0; l < AFX_FE_LIMBS; l++) r.v[l] = base[(size_t)(c * AFX_FE_LIMBS + l) * count + item];
  return r;
}
AFX_DEV ge_p3 var_load(const int32_t* base, uint32_t count, uint32_t item) {
  ge_p3 p;
  p.X = fe_load_soa(base, 0, count, item);
  p.Y = fe_load_soa(base, 1, count, item);
  p.Z = fe_load_soa(base, 2, count, item);
  p.T = fe_load_soa(base, 3, count, item);
  return p;
}
AFX_DEV void fe_store_soa(int32_t* base, uint32_t c, uint32_t count, uint32_t item, const fe& f) {
#pragma unroll
  for (int l = 0; l < AFX_FE_LIMBS; l++) base[(size_t)(c * AFX_FE_LIMBS + l) * count + item] = f.v[l];
}
AFX_DEV void var_store(int32_t* base, uint32_t count, uint32_t item, const ge_p3& p) {
  fe_store_soa(base, 0, count, item, p.X);
  fe_store_soa(base, 1, count, item, p.Y);
  fe_store_soa(base, 2, count, item, p.Z);
  fe_store_soa(base, 3, count, item, p.T);
}
AFX_DEV ge_p3 p3_load_uniform(const int32_t* c36) {
  ge_p3 p;
#pragma unroll
  for (int l = 0; l < AFX_FE_LIMBS; l++) { p.X.v[l] = c36[l]; p.Y.v[l] = c36[9 + l]; p.Z.v[l] = c36[18 + l]; p.T.v[l] = c36[27 + l]; }
  return p;
}
// one window-table entry (cached form): four field elements in canonical 32-byte form, 128 contiguous bytes,
// 16-byte aligned = exactly two 64-byte HBM sectors per gather (the limb form straddled 3.5 on average)
// `chunk` = dwords between the entry's consecutive 16-byte pieces: 4 (the 128 bytes contiguous: a lane's own entry of an
// item-major table) or 4 * count (piece-major: the entries of neighbouring items interleaved piece by piece, so that each of the
// eight loads of a wave that reads ONE entry index - a NAF table - is 1 KB contiguous instead of 64 pieces 128 bytes apart)
AFX_DEV void cached_store(int32_t* p, size_t chunk, const ge_cached& q) {
  uint32_t w[32];
  fe_tobytes(w, q.YpX);
  fe_tobytes(w + 8, q.YmX);
  fe_tobytes(w + 16, q.Z2);
  fe_tobytes(w + 24, q.T2d);
#pragma unroll
  for (int i = 0; i < 8; i++) *reinterpret_cast<uint4*>(p + i * chunk) = make_uint4(w[4 * i], w[4 * i + 1], w[4 * i + 2], w[4 * i + 3]);
}
// one field element as a canonical 32-byte word in pieces `first`, `first` + 1 of an entry
AFX_DEV void fe_store_pieces(int32_t* p, size_t chunk, int first, const fe& f) {
  uint32_t w[8];
  fe_tobytes(w, f);
  *reinterpret_cast<uint4*>(p + first * chunk) = make_uint4(w[0], w[1], w[2], w[3]);
  *reinterpret_cast<uint4*>(p + (first + 1) * chunk) = make_uint4(w[4], w[5], w[6], w[7]);
}
AFX_DEV fe fe_load_pieces(const int32_t* p, size_t chunk, int first) {
  const uint4 a = *reinterpret_cast<const uint4*>(p + first * chunk), b = *reinterpret_cast<const uint4*>(p + (first + 1) * chunk);
  const uint32_t w[8] = { a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w };
  return fe_frombytes(w);
}
// an entry of a narrow job's table before k_table_affine: X, Y, Z in pieces 0..5 (6, 7: that kernel's prefix product)
AFX_DEV void xyz_store(int32_t* p, size_t chunk, const ge_p3& q) {
  fe_store_pieces(p, chunk, 0, q.X);
  fe_store_pieces(p, chunk, 2, q.Y);
  fe_store_pieces(p, chunk, 4, q.Z);
}
AFX_DEV void cached_load_words(uint32_t w[32], const int32_t* p, size_t chunk) {
#pragma unroll
  for (int i = 0; i < 8; i++) {
    const uint4 t = *reinterpret_cast<const uint4*>(p + i * chunk);
    w[4 * i] = t.x; w[4 * i + 1] = t.y; w[4 * i + 2] = t.z; w[4 * i + 3] = t.w;
  }
}
AFX_DEV ge_cached cached_load(const int32_t* p, size_t chunk) {
  uint32_t w[32];
  cached_load_words(w, p, chunk);
  ge_cached q;
  q.YpX = fe_frombytes(w);
  q.YmX = fe_frombytes(w + 8);
  q.Z2 = fe_frombytes(w + 16);
  q.T2d = fe_frombytes(w + 24);
  return q;
}

// ---------------------------------------------------------------------------------------------
// setup (context creation): generators -> extended coords + encoding of the negation; positional tables
// ---------------------------------------------------------------------------------------------
__global__ void k_setup_generators(const uint8_t* __restrict__ enc, uint32_t ngen, int32_t* __restrict__ ext, uint8_t* __restrict__ neg_enc,
                                   uint32_t* __restrict__ ok) {
  const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= ngen) return;
  uint32_t w[8];
  enc_load(w, enc, g);
  ge_p3 P;
  const bool good = ristretto_decode(P, w);
  ok[g] = good ? 1u : 0u;
  if (!good) P = ge_identity();
  int32_t* e = ext + (size_t)g * AFX_VAR_DWORDS;
#pragma unroll
  for (int l = 0; l < AFX_FE_LIMBS; l++) { e[l] = P.X.v[l]; e[9 + l] = P.Y.v[l]; e[18 + l] = P.Z.v[l]; e[27 + l] = P.T.v[l]; }
  uint32_t nw[8];
  ristretto_encode(nw, ge_neg(P));
  enc_store(neg_enc, g, nw);
}

// Positional tables: for generator g and window position j, the entries d * 2^(BITS*j) * G_g for
// d = 0 .. 2^(BITS-1) as halved affine niels ((y+x)/2, (y-x)/2, dxy; ge.cuh).  Two kernels: the window bases B_{g,j} = 2^(BITS*j) G_g
// (thread per (g, j)), then thread (g, j, c) writes the 16 entries 16c .. 16c+15 with ONE field inversion
// (Montgomery's trick over the 16 Z coordinates).
// The same two kernels build the 13-bit tables of the public path and the 6-bit tables of the secret-independent one
// (AFX_SEC_*): `bits`, `windows`, `entries` (= 2^(bits-1) + 1) and the dword strides are launch arguments.
__global__ void k_setup_posbase(const int32_t* __restrict__ ext, uint32_t ngen, int32_t* __restrict__ base, uint32_t bits, uint32_t windows) {
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= ngen * windows) return;
  const uint32_t g = t / windows, j = t % windows;
  ge_p3 P = p3_load_uniform(ext + (size_t)g * AFX_VAR_DWORDS);
#pragma unroll 1
  for (uint32_t k = 0; k < bits * j; k++) P = ge_double(P);
  int32_t* e = base + (size_t)t * AFX_VAR_DWORDS;
#pragma unroll
  for (int l = 0; l < AFX_FE_LIMBS; l++) { e[l] = P.X.v[l]; e[9 + l] = P.Y.v[l]; e[18 + l] = P.Z.v[l]; e[27 + l] = P.T.v[l]; }
}
#define AFX_POS_CHUNK 16
__global__ void k_setup_postables(const int32_t* __restrict__ base, uint32_t ngen, int32_t* __restrict__ postab, uint32_t windows, uint32_t entries,
                                  uint32_t window_dwords) {
  const uint32_t chunks = (entries + AFX_POS_CHUNK - 1) / AFX_POS_CHUNK;
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= ngen * windows * chunks) return;
  const uint32_t gj = t / chunks, c = t % chunks;
  const ge_p3 B = p3_load_uniform(base + (size_t)gj * AFX_VAR_DWORDS);
  const ge_cached cB = ge_p3_to_cached(B);
  // Q = (16 c) * B by double-and-add over the bits of c (c < chunks), then four doublings
  ge_p3 Q = ge_identity();
  const int cbits = chunks > 1 ? 32 - __builtin_clz(chunks - 1) : 0;
#pragma unroll 1
  for (int bit = cbits - 1; bit >= 0; bit--) {
    Q = ge_double(Q);
    if ((c >> bit) & 1u) Q = ge_p1p1_to_p3(ge_add_cached(Q, cB, false));
  }
#pragma unroll 1
  for (int k = 0; k < 4; k++) Q = ge_double(Q);
  int32_t* tab = postab + (size_t)(gj / windows) * windows * window_dwords + (size_t)(gj % windows) * window_dwords;
  const uint32_t first = c * AFX_POS_CHUNK, n = min((uint32_t)AFX_POS_CHUNK, entries - first);
  // pass 1: the chunk's points (kept in per-thread scratch: this kernel runs once per context) and the running
  // products of their Z coordinates
  fe X[AFX_POS_CHUNK], Y[AFX_POS_CHUNK], Z[AFX_POS_CHUNK], pre[AFX_POS_CHUNK];
  fe prod = fe_one();
#pragma unroll 1
  for (uint32_t k = 0; k < n; k++) {
    X[k] = Q.X; Y[k] = Q.Y; Z[k] = Q.Z; pre[k] = prod;
    prod = fe_mul(prod, Q.Z);
    if (k + 1 < n) Q = ge_p1p1_to_p3(ge_add_cached(Q, cB, false));
  }
  // pass 2, backwards: inv = 1 / (Z_0 ... Z_k), so 1/Z_k = inv * (Z_0 ... Z_{k-1}); then inv *= Z_k
  fe inv = fe_invert(prod);
#pragma unroll 1
  for (int k = (int)n - 1; k >= 0; k--) {
    const fe zinv = fe_mul(inv, pre[k]);
    inv = fe_mul(inv, Z[k]);
    const ge_niels q = ge_niels_from_affine(fe_mul(X[k], zinv), fe_mul(Y[k], zinv));
    int32_t* e = tab + (size_t)(first + k) * AFX_NIELS_DWORDS;
#pragma unroll
    for (int l = 0; l < AFX_FE_LIMBS; l++) { e[l] = q.ypx.v[l]; e[9 + l] = q.ymx.v[l]; e[18 + l] = q.xyd.v[l]; }
    e[27] = 0;
  }
}

// ---------------------------------------------------------------------------------------------
// grid row -> (job, pass).  A plan's own launch: rows == null, row r runs jobs[r] of pass 0.  A launch merged from several plans
// (engine.cpp run_plans): rows[r] gives the pass and where the row's job lies, as a byte offset from `jobs` (then the base of the
// blob that holds every plan's job arrays where the plans left them - nothing is copied to merge).
// ---------------------------------------------------------------------------------------------
template <class T>
AFX_DEV const T* row_job(const T* __restrict__ jobs, const afx_row* __restrict__ rows) {
  if (rows) return reinterpret_cast<const T*>(reinterpret_cast<const uint8_t*>(jobs) + rows[blockIdx.y].job_off);
  return jobs + blockIdx.y;
}
AFX_DEV uint32_t row_pass_index(const afx_row* __restrict__ rows) { return rows ? rows[blockIdx.y].pass : 0u; }

// ---------------------------------------------------------------------------------------------
// decode / scalar checks / small point and scalar ops
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(AFX_BLOCK, 2) k_decode(const afx_decode_job* __restrict__ jobs, const afx_row* __restrict__ rows, const afx_pass* __restrict__ passes) {
  const afx_decode_job job = *row_job(jobs, rows);
  const afx_pass pass = passes[row_pass_index(rows)];   // wave-uniform: scalar loads
  const uint32_t count = pass.count;
  uint32_t* __restrict__ bad = pass.bad;
  const uint32_t item = blockIdx.x * blockDim.x + threadIdx.x;
  if (item >= count) return;
  uint32_t w[8];
  enc_load(w, job.enc, item);
  ge_p3 P;
  const bool ok = ristretto_decode(P, w);
  uint32_t flags = ok ? 0u : AFX_BAD_DECODE;
  if (job.reject_identity && is_identity_encoding(w)) flags |= AFX_BAD_IDENTITY;
  if (flags) atomicOr(&bad[item], flags);
  if (job.out) var_store(job.out, count, item, ok ? P : ge_identity());
}

// the same for a small pass's launch that also holds Elligator jobs (afx_decode_job.elligator): two kinds of square-root chain side by
// side in one launch, a grid row each, instead of one launch after the other
__global__ void __launch_bounds__(AFX_BLOCK) k_decode_mixed(const afx_decode_job* __restrict__ jobs, const afx_row* __restrict__ rows, const afx_pass* __restrict__ passes) {
  const afx_decode_job job = *row_job(jobs, rows);
  const afx_pass pass = passes[row_pass_index(rows)];   // wave-uniform: scalar loads
  const uint32_t count = pass.count;
  uint32_t* __restrict__ bad = pass.bad;
  const uint32_t item = blockIdx.x * blockDim.x + threadIdx.x;
  if (item >= count) return;
  uint32_t w[8];
  if (job.elligator) {   // (uniform)
    enc_load(w, job.enc, 2 * item + (job.elligator - 1));
    var_store(job.out, count, item, ristretto_elligator(fe_frombytes(w)));
    return;
  }
  enc_load(w, job.enc, item);
  ge_p3 P;
  const bool ok = ristretto_decode(P, w);
  uint32_t flags = ok ? 0u : AFX_BAD_DECODE;
  if (job.reject_identity && is_identity_encoding(w)) flags |= AFX_BAD_IDENTITY;
  if (flags) atomicOr(&bad[item], flags);
  if (job.out) var_store(job.out, count, item, ok ? P : ge_identity());
}

__global__ void __launch_bounds__(AFX_BLOCK) k_sccheck(const afx_sccheck_job* __restrict__ jobs, const afx_row* __restrict__ rows, const afx_pass* __restrict__ passes) {
  const afx_sccheck_job job = *row_job(jobs, rows);
  const afx_pass pass = passes[row_pass_index(rows)];   // wave-uniform: scalar loads
  const uint32_t count = pass.count;
  uint32_t* __restrict__ bad = pass.bad;
  const uint32_t item = blockIdx.x * blockDim.x + threadIdx.x;
  if (item >= count) return;
  const sc s = sc_load_item(job.sc, 32, item);
  if (!sc_is_canonical(s)) atomicOr(&bad[item], AFX_BAD_SCALAR);
}

__global__ void __launch_bounds__(AFX_BLOCK, 2) k_pointop(const afx_pointop_job* __restrict__ jobs, const afx_row* __restrict__ rows, const afx_pass* __restrict__ passes) {
  const afx_pointop_job job = *row_job(jobs, rows);
  const afx_pass pass = passes[row_pass_index(rows)];   // wave-uniform: scalar loads
  const uint32_t count = pass.count;
  uint32_t* __restrict__ bad = pass.bad;
  const uint32_t item = blockIdx.x * blockDim.x + threadIdx.x;
  if (item >= count) return;
  ge_p3 A = var_load(job.a, count, item);
  if (job.sa < 0) A = ge_neg(A);
  ge_p3 R = A;
  if (job.sb != 0) {
    const ge_p3 B = job.b ? var_load(job.b, count, item) : p3_load_uniform(job.b_const);
    R = ge_p1p1_to_p3(ge_add_cached(A, ge_p3_to_cached(B), job.sb < 0));
  } else {
    R = ge_carry(R);
  }
  if (job.out) var_store(job.out, count, item, R);
  if (job.out_enc) {
    uint32_t w[8];
    ristretto_encode(w, R);
    enc_store(job.out_enc, item, w);
    // 1: the point enters a transcript, the identity is rejected; 2: the point must BE the identity (strict-mode equality check)
    if (job.reject_identity == 1 ? is_identity_encoding(w) : (job.reject_identity == 2 && !is_identity_encoding(w))) atomicOr(&bad[item], AFX_BAD_IDENTITY);
  }
}

__global__ void __launch_bounds__(AFX_BLOCK) k_scalarop(const afx_scalarop_job* __restrict__ jobs, const afx_row* __restrict__ rows, const afx_pass* __restrict__ passes) {
  const afx_scalarop_job job = *row_job(jobs, rows);
  const afx_pass pass = passes[row_pass_index(rows)];   // wave-uniform: scalar loads
  const uint32_t count = pass.count;
  const uint32_t item = blockIdx.x * blockDim.x + threadIdx.x;
  if (item >= count) return;
  const sc a = sc_load_item(job.a, job.a_stride, item);
  const sc b = sc_load_item(job.b, job.b_stride, item);
  sc r;
  if (job.c) r = sc_muladd(a, b, sc_load_item(job.c, job.c_stride, item));
  else r = sc_mul(a, b);
  if (job.negate) r = sc_neg(r);
  uint32_t w[8];
#pragma unroll
  for (int i = 0; i < 8; i++) w[i] = r.v[i];
  enc_store(job.out, item, w);
}

// ---------------------------------------------------------------------------------------------
// k_msm: one small multiscalar multiplication per lane
// ---------------------------------------------------------------------------------------------
// Straus with a shared doubling chain.  Per-item scalars are recoded without carries (s + 0x88..88 for 4-bit
// signed digits in [-8,7], s + 0x80..80 for 8-bit signed digits in [-128,127]), so every lane adds at every window:
// no divergence; a zero digit adds table entry 0 = the identity.
//   variable bases, per-item scalar : 4-bit windows, 9-entry cached tables built per lane into HBM workspace (9 x 128 B
//                   per base per item, entries in canonical 32-byte form), one entry = eight 16-byte loads
//   variable bases, batch-constant scalar (the issuer key): width-5 NAF computed on the host, identical for every lane,
//                   odd multiples 1..15 in the same table slots; bit-serial chain with uniform branches
//   fixed bases   : positional tables built at context creation: for every window position j the affine-niels
//                   entries d * 2^(AFX_POS_BITS*j) * G, d = 0 .. 2^(AFX_POS_BITS-1).  A fixed base costs AFX_POS_WINDOWS
//                   (20 at 13 bits) additions and takes no part in the doubling chain: they are added after it.
//                   (Round 1 first ran them inside the chain with 8-bit windows from LDS-staged tables; reading the
//                   same tables from L2 measured equally fast, and positional tables beat both.)
// Field work per 4-bit window: 4 doublings (4 x 4S + 3 x 3M + 4M), 8M per variable term (the last addition of a
// window skips the T coordinate, -1M); 7M per fixed-base addition.  Assembler::msm (engine.cpp) counts the same
// schedule for afx_ctx_get_plan_stats.
// the identity in window-table entry form (Y+X = 1, Y-X = 1, 2Z = 2, 2dT = 0 as canonical 32-byte words): what digit 0 adds
__device__ __attribute__((aligned(16))) const int32_t AFX_IDENTITY_ENTRY[AFX_TABLE_ENTRY_DWORDS] = {
  1, 0, 0, 0, 0, 0, 0, 0, 1, 0, 0, 0, 0, 0, 0, 0, 2, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0 };
// per-lane context of one job inside k_msm
struct msm_env {
  const afx_msm_djob* job;
  const afx_msm_term* __restrict__ term;   // the job's terms (plan.h afx_job_terms: an offset from the job, itself reached from a kernel argument)
  const int32_t* table_ws;
  const uint32_t* digit_ws;
  uint32_t count, item, dslot, tslot;
  bool narrow;   // SEC instances: the job's variable terms run AFX_SECVAR_BITS-bit windows (afx_msm_job.narrow; wave-uniform)
};
// acc += (4-bit signed digit of window w) * (variable base t), from the lane's own window table
// `next` (wave-uniform): what consumes the result, GE_FOR_* (ge.cuh)
AFX_DEV ge_p3 msm_add_var(const msm_env& e, const ge_p3& acc, uint32_t t, int w, int next) {
  const int32_t* table = e.table_ws + ((size_t)e.term[t].table_slot * e.count + e.item) * AFX_VAR_TABLE_DWORDS;
  const uint32_t wd = (uint32_t)w + e.term[t].win_off;   // (a segment's window w is digit w + win_off of the scalar; uniform)
  const uint32_t word = e.digit_ws[((size_t)(e.dslot + t) * AFX_DIGIT_WORDS + (wd >> 3)) * e.count + e.item];
  const int d = (int)((word >> ((wd & 7) * 4)) & 15u) - 8;
  const uint32_t idx = (uint32_t)(d < 0 ? -d : d);
  const bool neg = (d < 0) != (e.term[t].negate != 0);
  // multiples 1..8 are stored (at 0..7); digit 0 reads the one identity entry every lane shares (an address select, no table bytes)
  const uint32_t stored = idx ? idx - 1 : 0;
  const int32_t* ent = idx ? table + stored * AFX_TABLE_ENTRY_DWORDS : AFX_IDENTITY_ENTRY;
  return ge_p1p1_to_p3_next(ge_add_cached(acc, cached_load(ent, 4), neg), next);
}
// A job with a secret scalar on a variable base (afx_msm_job.narrow, SEC instances): AFX_SECVAR_BITS-bit signed digits (msm_recode)
// over tables of AFX_SECVAR_STORED multiples at the head of the slot, each an AFFINE entry in the halved niels form of the
// positional tables - (y+x)/2, (y-x)/2, dxy as three canonical 32-byte words, 96 bytes (k_msm_tables<TABLE_NARROW> leaves X, Y, Z
// there and k_table_affine divides, one inversion per item for all of its tables) - so that an addition is ge_madd's 7 products
// instead of 8 and reads three quarters of the bytes.  Every addition of such a job reads ALL the stored entries of its table, in
// order (narrow_fetch: no address depends on a digit), and keeps the digit's with selects, the identity for digit 0
// (narrow_select) - what dalek's constant-time LookupTable::select does on the CPU (/root/reference/src/amacs.rs:267-270
// multiplies by the key with it).  The chain is software-pipelined: once an addition's entries have been reduced to the selected
// one, the words of the NEXT addition's table are requested, before this addition computes (msm_chain_narrow), so the reads
// overlap the arithmetic instead of preceding it.
#define AFX_NARROW_ENTRY_WORDS 24   // of the entry's 32 dwords (AFX_TABLE_ENTRY_DWORDS: the slot layout is the cached tables')
// the identity in that form: (1/2, 1/2, 0), 1/2 = (p + 1)/2 = 2^254 - 9
__device__ __attribute__((aligned(16))) const uint32_t AFX_IDENTITY_NIELS[AFX_NARROW_ENTRY_WORDS] = {
  0xfffffff7u, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0x3fffffffu,
  0xfffffff7u, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0x3fffffffu,
  0, 0, 0, 0, 0, 0, 0, 0 };
AFX_DEV void narrow_fetch(uint32_t (&buf)[AFX_SECVAR_STORED * AFX_NARROW_ENTRY_WORDS], uint64_t& digits, const msm_env& e, uint32_t t, int w) {
  // the digit's word(s) first: the loads come back in order, and the addition that consumes this fetch starts from the digit
  // (a segment's window w is digit w + win_off of the scalar: afx_msm_term.win_off, uniform)
  const uint32_t o = AFX_SECVAR_BITS * ((uint32_t)w + e.term[t].win_off), k = o >> 5, sh = o & 31u;
  const uint32_t* dw = e.digit_ws + ((size_t)(e.dslot + t) * AFX_DIGIT_WORDS + k) * e.count + e.item;
  digits = dw[0];
  if (sh + AFX_SECVAR_BITS > 32) digits |= (uint64_t)dw[e.count] << 32;   // uniform condition; k + 1 <= 8
  // [entry][piece][item]: the 64 lanes of a wave read 1 KB contiguous per load
  const int32_t* table = e.table_ws + (size_t)e.term[t].table_slot * e.count * AFX_VAR_TABLE_DWORDS + (size_t)e.item * 4;
#pragma unroll
  for (uint32_t m = 0; m < AFX_SECVAR_STORED; m++) {
    const int32_t* p = table + (size_t)m * e.count * AFX_TABLE_ENTRY_DWORDS;
#pragma unroll
    for (int i = 0; i < AFX_NARROW_ENTRY_WORDS / 4; i++) {
      const uint4 q = *reinterpret_cast<const uint4*>(p + i * (size_t)e.count * 4);
      buf[AFX_NARROW_ENTRY_WORDS * m + 4 * i] = q.x; buf[AFX_NARROW_ENTRY_WORDS * m + 4 * i + 1] = q.y;
      buf[AFX_NARROW_ENTRY_WORDS * m + 4 * i + 2] = q.z; buf[AFX_NARROW_ENTRY_WORDS * m + 4 * i + 3] = q.w;
    }
  }
}
// the entry the digit of window w names among the fetched ones (the identity for digit 0), and whether it is subtracted
AFX_DEV void narrow_select(ge_niels& q, bool& neg, const msm_env& e, uint32_t t, int w, const uint32_t (&buf)[AFX_SECVAR_STORED * AFX_NARROW_ENTRY_WORDS], uint64_t digits) {
  const uint32_t sh = (AFX_SECVAR_BITS * ((uint32_t)w + e.term[t].win_off)) & 31u;
  const int d = (int)((uint32_t)(digits >> sh) & ((1u << AFX_SECVAR_BITS) - 1)) - (1 << (AFX_SECVAR_BITS - 1));
  const uint32_t idx = (uint32_t)(d < 0 ? -d : d);
  neg = (d < 0) != (e.term[t].negate != 0);
  uint32_t sel[AFX_NARROW_ENTRY_WORDS];
#pragma unroll
  for (int i = 0; i < AFX_NARROW_ENTRY_WORDS; i++) sel[i] = AFX_IDENTITY_NIELS[i];
#pragma unroll
  for (uint32_t m = 0; m < AFX_SECVAR_STORED; m++) {
    const bool hit = idx == m + 1;
#pragma unroll
    for (int i = 0; i < AFX_NARROW_ENTRY_WORDS; i++) sel[i] = hit ? buf[AFX_NARROW_ENTRY_WORDS * m + i] : sel[i];
  }
  q.ypx = fe_frombytes(sel); q.ymx = fe_frombytes(sel + 8); q.xyd = fe_frombytes(sel + 16);
}
AFX_DEV ge_p3 msm_chain_narrow(const msm_env& e, ge_p3 acc, uint32_t nv) {
  uint32_t buf[AFX_SECVAR_STORED * AFX_NARROW_ENTRY_WORDS];
  uint64_t digits;
  const int top = (int)e.job->wins - 1;   // (AFX_SECVAR_WINDOWS - 1, or a segment's share: uniform)
  narrow_fetch(buf, digits, e, 0, top);
#pragma unroll 1
  for (int w = top; w >= 0; w--) {
    if (w != top) {
      ge_p2 a2 = ge_p3_to_p2(acc);
#pragma unroll 1
      for (int k = 0; k < AFX_SECVAR_BITS - 1; k++) a2 = ge_p1p1_to_p2_before_dbl(ge_p2_dbl(a2));
      acc = ge_p1p1_to_p3_for<GE_FOR_ADD>(ge_p2_dbl(a2));
    }
#pragma unroll 1
    for (uint32_t t = 0; t < nv; t++) {
      ge_niels q;
      bool neg;
      narrow_select(q, neg, e, t, w, buf, digits);
      const bool last = t + 1 == nv;
      if (!(last && w == 0)) narrow_fetch(buf, digits, e, last ? 0 : t + 1, last ? w - 1 : w);
      acc = ge_p1p1_to_p3_next(ge_madd(acc, q, neg), !last ? GE_FOR_ADD : (w == 0 ? GE_FOR_ANY : GE_FOR_DBL));
    }
  }
  return acc;
}
// acc += (AFX_POS_BITS-bit signed digit j) * 2^(AFX_POS_BITS*j) * (generator of term t), from the positional tables
AFX_DEV ge_p3 msm_add_positional(const msm_env& e, const int32_t* __restrict__ pos_tables, const ge_p3& acc, uint32_t t, uint32_t j, int next) {
  const uint32_t o = AFX_POS_BITS * j, k = o >> 5, sh = o & 31u;
  const uint32_t* dw = e.digit_ws + ((size_t)(e.dslot + t) * AFX_DIGIT_WORDS + k) * e.count + e.item;
  uint64_t w = dw[0];
  if (sh + AFX_POS_BITS > 32) w |= (uint64_t)dw[e.count] << 32;   // uniform condition (j is uniform); k + 1 <= 8
  const int d = (int)((uint32_t)(w >> sh) & ((1u << AFX_POS_BITS) - 1)) - (1 << (AFX_POS_BITS - 1));
  const uint32_t idx = (uint32_t)(d < 0 ? -d : d);
  const bool neg = (d < 0) != (e.term[t].negate != 0);
  const int4* p = reinterpret_cast<const int4*>(pos_tables + (size_t)e.term[t].fixed_idx * AFX_POS_TABLE_DWORDS +
                                                (size_t)j * AFX_POS_WINDOW_DWORDS + idx * AFX_NIELS_DWORDS);
  int32_t v[AFX_NIELS_DWORDS];   // 7 x 16 bytes: 27 limbs + padding
#pragma unroll
  for (int l = 0; l < AFX_NIELS_DWORDS / 4; l++) { const int4 x = p[l]; v[4 * l] = x.x; v[4 * l + 1] = x.y; v[4 * l + 2] = x.z; v[4 * l + 3] = x.w; }
  ge_niels q;
#pragma unroll
  for (int l = 0; l < AFX_FE_LIMBS; l++) { q.ypx.v[l] = v[l]; q.ymx.v[l] = v[9 + l]; q.xyd.v[l] = v[18 + l]; }
  return ge_p1p1_to_p3_next(ge_madd(acc, q, neg), next);
}

// acc += (signed AFX_SEC_BITS-bit digit j) * 2^(AFX_SEC_BITS*j) * (generator of term t) for a SECRET scalar.  No memory address
// depends on the digit: the 32 stored multiples of window j are read ONE PER LANE (lane l and lane l + 32 read multiple (l & 31) + 1:
// the address is a function of the lane's id), and each lane then takes the dwords of the multiple its digit names from the lane
// that holds it with ds_bpermute_b32 - a register-to-register exchange through the LDS crossbar that touches no memory.  Its source
// slots are lanes 0..31 only, one per LDS bank, so any pattern of digits is served without a bank conflict (equal digits read one
// slot: a broadcast) - tools/ubench/bperm_lookup.hip times the patterns, SQ_LDS_BANK_CONFLICT stays 0.  Digit 0 takes the
// identity (entry 0 of the window, read through scalar registers) with a select.  Every lane of a live wave is active here
// (msm_body: lanes past the end shadow the last item), which the exchange needs of its source lanes.
// (Rounds 3-4 read all nine entries of a 4-bit window through scalar registers and selected: 64 additions per term, 216 selects each.)
static_assert(AFX_SEC_ENTRIES == 33, "one stored multiple per lane of a 32-lane half: 6-bit signed digits");
AFX_DEV ge_p3 msm_add_positional_secret(const msm_env& e, const int32_t* __restrict__ sec_tables, const ge_p3& acc, uint32_t t, uint32_t j, int next) {
  const uint32_t o = AFX_SEC_BITS * j, k = o >> 5, sh = o & 31u;
  const uint32_t* dw = e.digit_ws + ((size_t)(e.dslot + t) * AFX_DIGIT_WORDS + k) * e.count + e.item;
  uint64_t w = dw[0];
  if (sh + AFX_SEC_BITS > 32) w |= (uint64_t)dw[e.count] << 32;   // uniform condition (j is uniform); k + 1 <= 8
  const int d = (int)((uint32_t)(w >> sh) & ((1u << AFX_SEC_BITS) - 1)) - (1 << (AFX_SEC_BITS - 1));
  const uint32_t idx = (uint32_t)(d < 0 ? -d : d);
  const bool neg = (d < 0) != (e.term[t].negate != 0);
  const int32_t* win = sec_tables + (size_t)e.term[t].fixed_idx * AFX_SEC_TABLE_DWORDS + (size_t)j * AFX_SEC_WINDOW_DWORDS;
  // this lane's share of the window: multiple (lane & 31) + 1, seven 16-byte loads
  const int4* mine = reinterpret_cast<const int4*>(win + ((threadIdx.x & 31u) + 1u) * AFX_NIELS_DWORDS);
  int32_t held[AFX_NIELS_DWORDS];
#pragma unroll
  for (int l = 0; l < AFX_NIELS_DWORDS / 4; l++) { const int4 x = mine[l]; held[4 * l] = x.x; held[4 * l + 1] = x.y; held[4 * l + 2] = x.z; held[4 * l + 3] = x.w; }
  const int src = (int)(((idx - 1u) & 31u) << 2);   // byte offset of the source lane's slot; digit 0 reads some lane and drops it below
  uint32_t keep = idx != 0 ? 0xffffffffu : 0u;
  asm volatile("" : "+v"(keep));   // opaque: the compiler would turn the mask arithmetic back into a select on a lane mask and a copy
  int32_t v[27];
#pragma unroll
  for (int l = 0; l < 27; l++) {
    const int32_t id = win[l];   // entry 0, the identity in niels form: wave-uniform, scalar loads
    const int32_t got = __builtin_amdgcn_ds_bpermute(src, held[l]);
    v[l] = (int32_t)(((uint32_t)got & keep) | ((uint32_t)id & ~keep));
  }
  ge_niels q;
#pragma unroll
  for (int l = 0; l < AFX_FE_LIMBS; l++) { q.ypx.v[l] = v[l]; q.ymx.v[l] = v[9 + l]; q.xyd.v[l] = v[18 + l]; }
  return ge_p1p1_to_p3_next(ge_madd(acc, q, neg), next);
}

// One kernel per job class (the host sorts a launch list's jobs by class, engine.cpp Assembler::msm), so that each
// class gets a register allocation of its own: the three chain variants, table building and the ristretto encoding in
// one loop body had the allocator pay for their union (round 1: 68 VGPRs spilled, 276 B/lane of scratch).
//   MSM_FIXED   fixed bases only: positional tables, no doubling chain
//   MSM_WINDOW  variable bases with per-item scalars: signed 4-bit windows, 63 x 4 doublings
//   MSM_NAF     some variable bases carry a batch-constant scalar (the issuer key): bit-serial chain with their width-5
//               NAF digits as a uniform schedule; per-item terms keep their windows
// A job that consumes another job's out_var (Z -> constraint #1 of the presentation proof) sits in a later launch.
enum { MSM_FIXED = 0, MSM_WINDOW = 1, MSM_NAF = 2 };

// recode the per-item scalars of terms [from, nt), stored [slot][AFX_DIGIT_WORDS][count]
AFX_DEV void msm_recode(const afx_msm_djob* job, uint32_t* __restrict__ digit_ws, uint32_t count, uint32_t item, uint32_t from, uint32_t nv, uint32_t nt, bool narrow) {
  const uint32_t dslot = job->digit_slot;
  const afx_msm_term* __restrict__ terms = afx_job_terms(job);
#pragma unroll 1
  for (uint32_t t = from; t < nt; t++) {
    sc s = sc_load_item(terms[t].scalar, terms[t].scalar_stride, item);
    // the job computes half of its sum (k_compress2x encodes the double); the term's base holds half its point (leave_half):
    // each alone changes the scalar, both together cancel
    const bool halve = job->half_var != nullptr, twice = terms[t].dbl != 0;
    if (halve && !twice) s = sc_half(s);
    if (twice && !halve) s = sc_dbl(s);
    uint32_t b[9];
    b[8] = 0;
    if (t < nv && narrow) sc_bias_wide<AFX_SECVAR_BITS, AFX_SECVAR_WINDOWS>(b, s);   // variable bases of a job with a secret on one
    else if (t < nv) sc_bias(b, s, 0x88888888u);   // signed 4-bit digits: variable bases
    else if (terms[t].secret) sc_bias_wide<AFX_SEC_BITS, AFX_SEC_WINDOWS>(b, s);   // secret scalars on fixed bases
    else sc_bias_wide<AFX_POS_BITS, AFX_POS_WINDOWS>(b, s);
#pragma unroll
    for (int i = 0; i < AFX_DIGIT_WORDS; i++) digit_ws[((size_t)(dslot + t) * AFX_DIGIT_WORDS + i) * count + item] = b[i];
  }
}
// per-lane window table of one variable base: multiples 0..8 (signed 4-bit windows), or the odd multiples
// 1, 3, ..., 15 (ODD: the terms that run a width-5 NAF)
// Layout inside a slot's AFX_VAR_TABLE_DWORDS * count dwords: 4-bit-window tables are item-major ([item][entry]: a lane's 8 stored
// entries are contiguous, its digit picks one), NAF tables entry-major ([entry][item]: every lane of a wave reads the SAME
// entry; they are stored piece-major inside an entry, see cached_store).  `stride` = dwords between consecutive entries.
enum { TABLE_WINDOW = 0, TABLE_ODD = 1, TABLE_NARROW = 2, TABLE_NARROW_CACHED = 3 };
template <int TK>
AFX_DEV void msm_build_table(int32_t* __restrict__ tab, size_t stride, size_t chunk, const ge_p3& P) {
  ge_p3 Q = P;
  if (TK == TABLE_ODD) {
    const ge_cached c2 = ge_p3_to_cached(ge_double(P));
    cached_store(tab, chunk, ge_p3_to_cached(P));   // the canonical packing carries: no separate reduction
#pragma unroll 1
    for (int k = 1; k < 8; k++) {
      Q = ge_p1p1_to_p3(ge_add_cached(Q, c2, false));
      cached_store(tab + k * stride, chunk, ge_p3_to_cached(Q));
    }
  } else {
    const ge_cached cP = ge_p3_to_cached(P);   // P centred: Y+X, Y-X, 2Z within 1 unit, what ge_add_cached expects
    // k*P at entry k - 1; the identity (digit 0) is not stored.  A narrow job's entries: X, Y, Z for k_table_affine to divide - or,
    // TABLE_NARROW_CACHED, the cached form as it is (four-wave chains only: a product more per addition costs them no round)
    if (TK == TABLE_NARROW) xyz_store(tab, chunk, P); else cached_store(tab, chunk, cP);
#pragma unroll 1
    for (int k = 2; k <= ((TK == TABLE_NARROW || TK == TABLE_NARROW_CACHED) ? AFX_SECVAR_STORED : AFX_TABLE_STORED); k++) {
      Q = ge_p1p1_to_p3(ge_add_cached(Q, cP, false));
      if (TK == TABLE_NARROW) xyz_store(tab + (k - 1) * stride, chunk, Q); else cached_store(tab + (k - 1) * stride, chunk, ge_p3_to_cached(Q));
    }
  }
}
// the fixed bases of a job (terms [from, nt)): positional tables, no doubling; the sum's last step leaves acc centred.
// SEC: terms with a secret scalar take the AFX_SEC_BITS-bit tables through the lane exchange (AFX_SEC_WINDOWS additions each), the others the 13-bit ones.
template <bool SEC>
AFX_DEV ge_p3 msm_fixed_terms(const msm_env& e, const int32_t* __restrict__ pos_tables, const int32_t* __restrict__ sec_tables, ge_p3 acc, uint32_t from, uint32_t nt) {
  int last_pub = -1, last_sec = -1;   // wave-uniform
  if constexpr (SEC) {
#pragma unroll 1
    for (uint32_t t = from; t < nt; t++) { if (e.term[t].secret) last_sec = (int)t; else last_pub = (int)t; }
#pragma unroll 1
    for (uint32_t j = 0; j < AFX_SEC_WINDOWS && last_sec >= 0; j++) {
#pragma unroll 1
      for (uint32_t t = from; t < nt; t++)
        if (e.term[t].secret)
          acc = msm_add_positional_secret(e, sec_tables, acc, t, j, (last_pub < 0 && j + 1 == AFX_SEC_WINDOWS && (int)t == last_sec) ? GE_FOR_ANY : GE_FOR_ADD);
    }
    if (last_pub < 0) return acc;
  } else {
    last_pub = (int)nt - 1;
  }
#pragma unroll 1
  for (uint32_t j = 0; j < AFX_POS_WINDOWS; j++) {
#pragma unroll 1
    for (uint32_t t = from; t < nt; t++) {
      if (SEC && e.term[t].secret) continue;
      acc = msm_add_positional(e, pos_tables, acc, t, j, (j + 1 == AFX_POS_WINDOWS && (int)t == last_pub) ? GE_FOR_ANY : GE_FOR_ADD);
    }
  }
  return acc;
}
// addend, extended-coordinate output, compressed output (ENC = false: a launch none of whose jobs encodes here)
template <bool ENC>
AFX_DEV void msm_finish(const afx_msm_djob* job, ge_p3 acc, uint32_t* __restrict__ bad, uint32_t count, uint32_t item) {
  if (job->addend) {
    const ge_p3 A = var_load(job->addend, count, item);
    acc = ge_p1p1_to_p3(ge_add_cached(acc, ge_p3_to_cached(A), job->addend_negate != 0));
  }
  if (job->out_var) var_store(job->out_var, count, item, acc);
  if (job->half_var) { var_store(job->half_var, count, item, acc); return; }   // encoded by k_compress2x
  if (ENC && job->out_enc) {
    uint32_t wenc[8];
    ristretto_encode(wenc, acc);
    enc_store(job->out_enc, item, wenc);
    if (job->reject_identity && is_identity_encoding(wenc)) atomicOr(&bad[item], AFX_BAD_IDENTITY);
  }
}

// window tables of the variable bases of one k_msm launch, one (job, term) pair per grid row: the chain kernels only read them
// TK: TABLE_WINDOW the multiples 1..8 (signed 4-bit windows), TABLE_ODD the odd multiples 1..15 (width-5 NAF terms), TABLE_NARROW
// the multiples 1..AFX_SECVAR_STORED of a job with a secret scalar on a variable base.  A kernel instance each: one body with
// the layouts and the entry count as run-time values took 256 registers and scratch.
template <int TK>
__global__ void __launch_bounds__(AFX_BLOCK, 2)
k_msm_tables(const afx_table_job* __restrict__ jobs, const afx_row* __restrict__ rows, const afx_pass* __restrict__ passes) {
  const afx_pass pass = passes[row_pass_index(rows)];   // wave-uniform: scalar loads
  const uint32_t count = pass.count;
  int32_t* __restrict__ table_ws = pass.table_ws;
  // a WAVE past the end of this row's pass retires (the grid is sized for the launch's largest pass, and a pass of 16 items is one wave
  // of its block's four; no kernel here has a barrier).  Inside the last live wave, lanes past the end shadow the last item.
  if (blockIdx.x * blockDim.x + ((uint32_t)__builtin_amdgcn_readfirstlane((int)threadIdx.x) & ~63u) >= count) return;   // (first lane's id: the branch is scalar)
  const uint32_t item = min(blockIdx.x * blockDim.x + threadIdx.x, count - 1);
  const afx_table_job row = *row_job(jobs, rows);
  int32_t* slot = table_ws + (size_t)row.table_slot * count * AFX_VAR_TABLE_DWORDS;
  const ge_p3 P = var_load(row.var, count, item);
  // NAF tables and the tables of narrow jobs: [entry][piece][item][16 B]; window tables: [item][entry][128 B]
  if (TK == TABLE_WINDOW) msm_build_table<TK>(slot + (size_t)item * AFX_VAR_TABLE_DWORDS, AFX_TABLE_ENTRY_DWORDS, 4, P);
  else msm_build_table<TK>(slot + (size_t)item * 4, (size_t)count * AFX_TABLE_ENTRY_DWORDS, (size_t)count * 4, P);
}

// The tables of narrow jobs, second step: every entry (X : Y : Z) that k_msm_tables<TABLE_NARROW> left becomes the affine entry
// ((y+x)/2, (y-x)/2, dxy) the narrow chain adds with 7 products.  Lane = item walks the entries of its row's tables twice, like
// k_compress2x: prefix products of the Z forwards (kept in the entry's last two pieces), ONE inversion, the quotients backwards.
// Z is never 0 on the curve.  Rows: one per launch for large passes (one inversion per item); small passes, where the serial walk
// is what a call waits for, spread the tables over up to 8 rows (engine.cpp msm_list).
__global__ void __launch_bounds__(AFX_BLOCK, 2)
k_table_affine(const afx_table_job* __restrict__ jobs, const afx_walk_row* __restrict__ rows, const afx_pass* __restrict__ passes) {
  const afx_walk_row row = rows[blockIdx.y];   // wave-uniform
  const afx_pass pass = passes[row.pass];
  const uint32_t count = pass.count;
  const uint32_t item = blockIdx.x * blockDim.x + threadIdx.x;
  if (item >= count) return;
  jobs = reinterpret_cast<decltype(jobs)>(reinterpret_cast<const uint8_t*>(jobs) + row.job_off);
  const uint32_t n = row.n_jobs * AFX_SECVAR_STORED;
  const size_t chunk = (size_t)count * 4;
  int32_t* const mine = pass.table_ws + (size_t)item * 4;
  fe prod = fe_one();
#pragma unroll 1
  for (uint32_t k = 0; k < n; k++) {
    int32_t* e = mine + (size_t)jobs[k / AFX_SECVAR_STORED].table_slot * count * AFX_VAR_TABLE_DWORDS + (size_t)(k % AFX_SECVAR_STORED) * count * AFX_TABLE_ENTRY_DWORDS;
    fe_store_pieces(e, chunk, 6, prod);   // product of the Z before this entry
    prod = fe_mul(prod, fe_load_pieces(e, chunk, 4));
  }
  fe inv = fe_invert(prod);
#pragma unroll 1
  for (uint32_t kk = n; kk > 0; kk--) {
    const uint32_t k = kk - 1;
    int32_t* e = mine + (size_t)jobs[k / AFX_SECVAR_STORED].table_slot * count * AFX_VAR_TABLE_DWORDS + (size_t)(k % AFX_SECVAR_STORED) * count * AFX_TABLE_ENTRY_DWORDS;
    const fe zinv = fe_mul(inv, fe_load_pieces(e, chunk, 6));
    inv = fe_mul(inv, fe_load_pieces(e, chunk, 4));
    const ge_niels q = ge_niels_from_affine(fe_mul(fe_load_pieces(e, chunk, 0), zinv), fe_mul(fe_load_pieces(e, chunk, 2), zinv));
    fe_store_pieces(e, chunk, 0, q.ypx);
    fe_store_pieces(e, chunk, 2, q.ymx);
    fe_store_pieces(e, chunk, 4, q.xyd);
  }
}

// ENC: the launch has jobs that encode their result in this kernel (those with an addend or an extended-coordinate output as
// well; results that are only encoded go through k_compress2x).  A launch without such jobs - every windowed one of
// Issuer::verify, every one of Issuer::issue - runs the instance compiled without the encoder's inversion, which fits three
// blocks per CU without scratch (166 registers windowed and fixed, 146 NAF; the windowed chain 1.3 % faster than with two
// blocks, the fixed-base sums 4.5 %, same box).
// SEC: the launch has terms with secret scalars under afx_ctx_set_secret_independent_addressing (the prover paths, the key's
// terms of Issuer::verify): an instance of its own, so that the table scans cost the ordinary launches no registers.
// The body is shared by two kernels: k_msm, a plan's own launch - the pass's item count and workspace pointers are kernel
// ARGUMENTS, as they always were (restrict-qualified parameters: the 2^19-item passes of the throughput path measured 0.3-2 %
// slower with them read from a pass table: profiles/r04_ab_pass_descriptors.txt) - and k_msm_rows, a launch merged from several
// small plans, where every grid row finds its job and its pass through the row table (plan.h afx_row).
template <int KIND, bool ENC, bool SEC>
__device__ __forceinline__ void msm_body(const afx_msm_djob* __restrict__ job, const int32_t* __restrict__ pos_tables, const int32_t* __restrict__ sec_tables,
                                         int32_t* __restrict__ table_ws, uint32_t* __restrict__ digit_ws, uint32_t* __restrict__ bad, uint32_t count,
                                         unsigned long long* __restrict__ clock_probe) {
  // Clock probe (measurement aid): lane 0 of AFX_CLOCK_SLOTS blocks of the launch reads the shader-clock counter (s_memtime) and
  // the constant 100 MHz counter (s_memrealtime) around its chain; their ratio is the core clock that block actually ran at (the
  // kernels run at the socket power cap, below the nominal clock: DESIGN.md section 4).  The probing blocks are spread evenly
  // over the launch's linear block order - that is, over its DURATION (blocks start in order: the first ones run at the boost
  // clock the launch starts with, the later ones at what the power cap leaves) and over the eight XCDs (a block's XCD is its
  // linear id mod 8: slot k probes a block whose id is k mod 8).  One slot per probing block; the host takes the median.
  // (the choice is the block's, not a lane's: the counters stay in scalar registers - as lane 0's alone they cost the three-block
  // instances four vector registers they do not have)
  bool probe = false;
  uint32_t pslot = 0;
  if (clock_probe) {
    const uint32_t total = gridDim.x * gridDim.y, b = blockIdx.y * gridDim.x + blockIdx.x;
    if (total < 8 * AFX_CLOCK_SLOTS) { probe = b < AFX_CLOCK_SLOTS; pslot = b; }
    else {
      const uint32_t stride = (total / AFX_CLOCK_SLOTS) & ~7u;
      pslot = b / stride;
      probe = pslot < AFX_CLOCK_SLOTS && b - pslot * stride == (pslot & 7u);
    }
  }
  unsigned long long c0 = 0, r0 = 0;
  if (probe) { c0 = clock64(); r0 = wall_clock64(); }
  // a WAVE past the end of this row's pass retires (the grid is sized for the launch's largest pass, and a pass of 16 items is one wave
  // of its block's four; no kernel here has a barrier).  Inside the last live wave, lanes past the end shadow the last item.
  if (blockIdx.x * blockDim.x + ((uint32_t)__builtin_amdgcn_readfirstlane((int)threadIdx.x) & ~63u) >= count) return;   // (first lane's id: the branch is scalar)
  // lanes past the end of the batch shadow the last item (identical values, identical stores)
  const uint32_t item = min(blockIdx.x * blockDim.x + threadIdx.x, count - 1);
  const uint32_t nt = job->n_terms, nv = job->n_var, nu = job->n_uni;
  msm_env env;
  env.job = job; env.term = afx_job_terms(job); env.table_ws = table_ws; env.digit_ws = digit_ws;
  env.count = count; env.item = item; env.dslot = job->digit_slot; env.tslot = 0;
  env.narrow = SEC && KIND == MSM_WINDOW && job->narrow != 0;
  msm_recode(job, digit_ws, count, item, nu, nv, nt, env.narrow);   // batch-constant NAF terms need no digits
  ge_p3 acc = ge_identity();
  if constexpr (KIND == MSM_FIXED) {
    acc = msm_fixed_terms<SEC>(env, pos_tables, sec_tables, acc, 0, nt);
  } else {
    if constexpr (KIND == MSM_NAF) {
      // bit-serial chain: the batch-constant scalars' width-5 NAF digits (the same for every lane, so the branches are
      // uniform) add odd multiples at ~1/6 of the bit positions; per-item terms keep their 4-bit windows
      const uint32_t* sched = job->naf_sched;   // uniform: scalar loads
      const int top = job->top_bit;
      const uint32_t nvl = nv - nu;
      uint32_t ei = 0, ev = sched[0];
#pragma unroll 1
      for (int bit = top; bit >= 0; bit--) {
        const bool lane_adds = (bit & 3) == 0 && nvl != 0;
        const bool any_adds = lane_adds || (ev >> 16) == (uint32_t)bit;
        // what follows the last step at this bit position: the next doubling, or (bit 0) whatever comes after the chain
        const int after = bit == 0 ? GE_FOR_ANY : GE_FOR_DBL;
        if (bit != top) acc = ge_p1p1_to_p3_next(ge_p2_dbl(ge_p3_to_p2(acc)), any_adds ? GE_FOR_ADD : after);
#pragma unroll 1
        while ((ev >> 16) == (uint32_t)bit) {
          const uint32_t t = (ev >> 8) & 0xffu, idx = ev & 7u;
          const bool neg = (ev & 0x80u) != 0;
          ev = sched[++ei];
          const bool last = !lane_adds && (ev >> 16) != (uint32_t)bit;
          const int32_t* ent = table_ws + (size_t)env.term[t].table_slot * count * AFX_VAR_TABLE_DWORDS + (size_t)idx * count * AFX_TABLE_ENTRY_DWORDS + (size_t)item * 4;
          acc = ge_p1p1_to_p3_next(ge_add_cached(acc, cached_load(ent, (size_t)count * 4), neg), last ? after : GE_FOR_ADD);
        }
        if (lane_adds) {
#pragma unroll 1
          for (uint32_t t = nu; t < nv; t++) acc = msm_add_var(env, acc, t, bit >> 2, t + 1 != nv ? GE_FOR_ADD : after);
        }
      }
    } else {
      // (nv == 0: a job of fixed bases only that a small pass put into this launch, so that it runs beside the chains instead
      // of in a launch of its own before them - engine.cpp msm_list)
      bool chained = false;
      if constexpr (SEC) {
        if (env.narrow) { acc = msm_chain_narrow(env, acc, nv); chained = true; }   // wave-uniform: a property of the job
      }
      const int top = (int)job->wins - 1;   // 63, or a segment's share of the windows (uniform)
#pragma unroll 1
      for (int w = (nv && !chained) ? top : -1; w >= 0; w--) {
        if (w != top) {
          ge_p2 a2 = ge_p3_to_p2(acc);
#pragma unroll 1
          for (int k = 0; k < 3; k++) a2 = ge_p1p1_to_p2_before_dbl(ge_p2_dbl(a2));
          acc = ge_p1p1_to_p3_for<GE_FOR_ADD>(ge_p2_dbl(a2));
        }
#pragma unroll 1
        for (uint32_t t = 0; t + 1 < nv; t++) acc = msm_add_var(env, acc, t, w, GE_FOR_ADD);
        acc = msm_add_var(env, acc, nv - 1, w, w == 0 ? GE_FOR_ANY : GE_FOR_DBL);   // the window's last addition
      }
    }
    // the fixed bases of a job with variable bases: after the chain (any order gives the same sum)
    if (nt != nv) acc = msm_fixed_terms<SEC>(env, pos_tables, sec_tables, acc, nv, nt);
  }
  msm_finish<ENC>(job, acc, bad, count, item);
  if (probe) {
    const unsigned long long dc = (unsigned long long)clock64() - c0, dr = (unsigned long long)wall_clock64() - r0;
    if (threadIdx.x == 0) {
      atomicAdd(&clock_probe[2 * pslot], dc);
      atomicAdd(&clock_probe[2 * pslot + 1], dr);
    }
  }
}
// blocks of 256 per CU the instances are compiled for: three (168 registers) without the encoder, two with it; the windowed SEC
// instance (the narrow chain's entries in flight) AFX_SEC_WINDOW_OCCUPANCY
#ifndef AFX_SEC_WINDOW_OCCUPANCY
#define AFX_SEC_WINDOW_OCCUPANCY 2
#endif
#define AFX_MSM_OCCUPANCY(KIND, ENC, SEC) ((ENC) ? 2 : !(SEC) || (KIND) == MSM_FIXED ? 3 : (KIND) == MSM_WINDOW ? AFX_SEC_WINDOW_OCCUPANCY : 2)
template <int KIND, bool ENC, bool SEC>
__global__ void __launch_bounds__(AFX_BLOCK, AFX_MSM_OCCUPANCY(KIND, ENC, SEC))
k_msm(const afx_msm_djob* __restrict__ jobs, const int32_t* __restrict__ pos_tables, const int32_t* __restrict__ sec_tables, int32_t* __restrict__ table_ws,
      uint32_t* __restrict__ digit_ws, uint32_t* __restrict__ bad, uint32_t count, unsigned long long* __restrict__ clock_probe) {
  msm_body<KIND, ENC, SEC>(&jobs[blockIdx.y], pos_tables, sec_tables, table_ws, digit_ws, bad, count, clock_probe);
}
// (small plans only: they have no NAF schedules - engine.cpp msm_split(no_naf) - so there is no merged NAF instance)
template <int KIND, bool ENC, bool SEC>
__global__ void __launch_bounds__(AFX_BLOCK, AFX_MSM_OCCUPANCY(KIND, ENC, SEC))
k_msm_rows(const uint8_t* __restrict__ blob, const int32_t* __restrict__ pos_tables, const int32_t* __restrict__ sec_tables, const afx_row* __restrict__ rows,
           const afx_pass* __restrict__ passes, unsigned long long* __restrict__ clock_probe) {
  const afx_row row = rows[blockIdx.y];      // wave-uniform: scalar loads
  const afx_pass pass = passes[row.pass];
  msm_body<KIND, ENC, SEC>(reinterpret_cast<const afx_msm_djob*>(blob + row.job_off), pos_tables, sec_tables, pass.table_ws, pass.digit_ws, pass.bad, pass.count, clock_probe);
}

// ---------------------------------------------------------------------------------------------
// k_msm_quad: the same jobs with FOUR waves per item chain, for passes that leave the device idle
// ---------------------------------------------------------------------------------------------
// What one small call waits for is its longest chain - 252 doublings and 64 additions in a row on a lane whose wave is alone on
// its SIMD, where a wave issues at less than half the SIMD's rate.  Here a block is four waves = four ROLES over the same 64
// items: every point operation is two rounds of four independent field products (a doubling: X^2, Y^2, Z^2, (Y-X)^2, then the
// four products of the completed point; an addition: its four products with the table entry, then the same four), each role
// computes one product per round on its own SIMD and the four results change hands through LDS (two buffers, one barrier per
// round).  A round costs one product and the exchange instead of four products: the chain is ~3x shorter in time at 4x the
// waves, which a small pass has to spare.  The arithmetic is the centred flavour throughout (fe_mul, fe_sq: every operand
// within the bounds ge.cuh states for the completed point), the digits, table layouts and job fields are k_msm's, and so is the
// result, bit for bit (a field element has one centred representation).  Role 0 recodes the scalars and finishes the job.
// Taken by afxk_msm for windowed and fixed-base launches without in-kernel encodings whose grid is small.
struct quad_lds { int4 v[2][4][3][64]; };   // [buffer][slot][16-byte piece][lane]: 9 limbs + padding per element, 24 KB
AFX_DEV void quad_put(quad_lds& L, int buf, uint32_t slot, uint32_t lane, const fe& f) {
  L.v[buf][slot][0][lane] = make_int4(f.v[0], f.v[1], f.v[2], f.v[3]);
  L.v[buf][slot][1][lane] = make_int4(f.v[4], f.v[5], f.v[6], f.v[7]);
  L.v[buf][slot][2][lane] = make_int4(f.v[8], 0, 0, 0);
}
AFX_DEV fe quad_get(const quad_lds& L, int buf, uint32_t slot, uint32_t lane) {
  const int4 a = L.v[buf][slot][0][lane], b = L.v[buf][slot][1][lane], c = L.v[buf][slot][2][lane];
  fe f;
  f.v[0] = a.x; f.v[1] = a.y; f.v[2] = a.z; f.v[3] = a.w; f.v[4] = b.x; f.v[5] = b.y; f.v[6] = b.z; f.v[7] = b.w; f.v[8] = c.x;
  return f;
}
// every role hands in one element; afterwards each reads the ones it needs.  The exchange is ordered by workgroup-scope fences on
// the LDS address space alone, on both sides of the barrier: a release after the store (s_waitcnt lgkmcnt(0): LDS is coherent
// inside the CU), an acquire before the loads - what __syncthreads() does, minus its wait for this wave's outstanding GLOBAL
// loads: the next table entry is already on its way (msm_quad_body) and must stay in flight across the exchange.  (The bare
// s_barrier intrinsic touches no memory as far as the compiler knows: without the fences nothing but may-alias analysis kept
// the loads behind it.  tests/test_kernel_isa.py reads the order ds_write .. s_waitcnt lgkmcnt(0) .. s_barrier .. ds_read back.)
AFX_DEV void quad_post(quad_lds& L, int buf, uint32_t role, uint32_t lane, const fe& mine) {
  quad_put(L, buf, role, lane, mine);
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}
// completed point -> extended, a product per role: T*X, Y*Z, T*Z, Y*X (ge_p1p1_to_p3); the next round writes the other buffer -
// whoever writes this one again has passed the next barrier, behind every reader.  (Every role reads all four back: reading only
// the coordinates its part of the next operation takes - 1 or 2 of them - measured SLOWER, 0.42 against 0.38 ms per chain stage:
// the branches cost more than the loads.)
AFX_DEV ge_p3 quad_complete(quad_lds& L, int& buf, uint32_t role, uint32_t lane, const ge_p1p1& r) {
  const fe m = fe_mul((role == 0 || role == 2) ? r.T : r.Y, (role == 0 || role == 3) ? r.X : r.Z);
  quad_post(L, buf, role, lane, m);
  ge_p3 o;
  o.X = quad_get(L, buf, 0, lane); o.Y = quad_get(L, buf, 1, lane); o.Z = quad_get(L, buf, 2, lane); o.T = quad_get(L, buf, 3, lane);
  buf ^= 1;
  return o;
}
// 2 * p (ge_p2_dbl with centred squares: 2XY from (Y - X)^2)
AFX_DEV ge_p3 quad_dbl(quad_lds& L, int& buf, uint32_t role, uint32_t lane, const ge_p3& p) {
  const fe sq = fe_sq(role == 0 ? p.X : role == 1 ? p.Y : role == 2 ? p.Z : fe_sub(p.Y, p.X));
  quad_post(L, buf, role, lane, sq);
  const fe XX = quad_get(L, buf, 0, lane), YY = quad_get(L, buf, 1, lane), ZZ = quad_get(L, buf, 2, lane), AA = quad_get(L, buf, 3, lane);
  buf ^= 1;
  ge_p1p1 r;
  r.Y = fe_add(YY, XX);
  r.Z = fe_sub(YY, XX);
  r.X = fe_sub(r.Y, AA);
  r.T = fe_sub(fe_add(ZZ, ZZ), r.Z);
  return quad_complete(L, buf, role, lane, r);
}
// p +- (an entry): `e0` is what this role multiplies by - role 0 the entry's Y+X (Y-X when subtracting), role 1 its Y-X (Y+X),
// role 2 its 2dT, role 3 its 2Z (a cached entry) or nothing (NIELS: a halved affine entry, D = Z)
template <bool NIELS>
AFX_DEV ge_p3 quad_add(quad_lds& L, int& buf, uint32_t role, uint32_t lane, const ge_p3& p, const fe& e0, bool neg) {
  fe m;
  if (role == 0) m = fe_mul(fe_add(p.Y, p.X), e0);
  else if (role == 1) m = fe_mul(fe_sub(p.Y, p.X), e0);
  else if (role == 2) m = fe_mul(fe_cneg(e0, !neg), p.T);   // -C (the sign flipped again when subtracting): ge_add_cached
  else m = NIELS ? p.Z : fe_mul(p.Z, e0);
  quad_post(L, buf, role, lane, m);
  const fe A = quad_get(L, buf, 0, lane), B = quad_get(L, buf, 1, lane), Cn = quad_get(L, buf, 2, lane), D = quad_get(L, buf, 3, lane);
  buf ^= 1;
  ge_p1p1 r;
  r.X = fe_sub(A, B);
  r.Y = fe_add(A, B);
  r.Z = fe_sub(D, Cn);
  r.T = fe_add(D, Cn);
  return quad_complete(L, buf, role, lane, r);
}
// this role's 32 bytes of a window-table entry (cached form: Y+X | Y-X | 2Z | 2dT, canonical), as loaded words
struct quad_words { uint4 a, b; };
AFX_DEV quad_words quad_entry_load(const int32_t* ent, uint32_t role, bool neg) {
  const uint32_t part = role == 0 ? (neg ? 1u : 0u) : role == 1 ? (neg ? 0u : 1u) : role == 2 ? 3u : 2u;
  quad_words q;
  q.a = *reinterpret_cast<const uint4*>(ent + 8 * part);
  q.b = *reinterpret_cast<const uint4*>(ent + 8 * part + 4);
  return q;
}
AFX_DEV fe quad_entry_fe(const quad_words& q) {
  const uint32_t w[8] = { q.a.x, q.a.y, q.a.z, q.a.w, q.b.x, q.b.y, q.b.z, q.b.w };
  return fe_frombytes(w);
}
// SEC: the launch has terms with secret scalars (afx_ctx_set_secret_independent_addressing): the same two mechanisms as k_msm's SEC
// instances, a role's share of each.  A job with a secret on a per-item base (afx_msm_job.narrow) runs the 2-bit windows: roles 0
// and 1 read (y+x)/2 AND (y-x)/2 of BOTH stored entries of the lane's table, role 2 dxy of both - every byte either could need,
// at addresses made of the item only - and keep the digit's with selects (the identity for digit 0; the sign picks between the
// two halves by a select as well, where the public path picks an address).  A secret on a generator takes its limbs from the
// lane that holds the digit's multiple (ds_bpermute_b32, source lanes 0..31: msm_add_positional_secret), 18 limbs for roles 0
// and 1, 9 for role 2.  tests/test_kernel_isa.py reads the instances for it.
// [entry][16-byte half]: roles 0, 1: a = (y+x)/2, b = (y-x)/2; role 2: a = dxy, b unused
struct quad_narrow { uint4 a[2][2], b[2][2]; };
// a lane mask the compiler cannot see through (it would turn mask arithmetic back into selects, and selects on a lane mask into
// branches around the work: a branch on a secret)
AFX_DEV uint32_t quad_mask(bool b) { uint32_t m = b ? 0xffffffffu : 0u; asm volatile("" : "+v"(m)); return m; }
AFX_DEV uint32_t quad_pick(uint32_t m, uint32_t yes, uint32_t no) { return (yes & m) | (no & ~m); }
template <bool SEC>
__device__ __forceinline__ void msm_quad_body(const afx_msm_djob* __restrict__ job, const int32_t* __restrict__ pos_tables, const int32_t* __restrict__ sec_tables,
                                              int32_t* __restrict__ table_ws, uint32_t* __restrict__ digit_ws, uint32_t* __restrict__ bad, uint32_t count) {
  __shared__ quad_lds L;
  if (blockIdx.x * 64u >= count) return;   // block-uniform: the grid is sized for the launch's largest pass
  const uint32_t lane = threadIdx.x & 63u, role = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const uint32_t item = min(blockIdx.x * 64u + lane, count - 1);   // lanes past the end shadow the last item
  const uint32_t nt = job->n_terms, nv = job->n_var;
  const bool narrow = SEC && job->narrow != 0;
  msm_env e;
  e.job = job; e.term = afx_job_terms(job); e.table_ws = table_ws; e.digit_ws = digit_ws;
  e.count = count; e.item = item; e.dslot = job->digit_slot; e.tslot = 0; e.narrow = narrow;
  if (role == 0) msm_recode(job, digit_ws, count, item, 0, nv, nt, narrow);
  __syncthreads();   // the digits are in memory for the other roles (same CU: one vector cache)
  int buf = 0;
  ge_p3 acc = ge_identity();
  bool chained = false;
  if constexpr (SEC) {
    if (narrow) {
      chained = true;
      const bool cached = job->narrow == 2;   // (uniform) a segmenting pass: the tables were never made affine
      // everything addition (w, t) could need of the lane's two-entry table, and its digit word; requested an addition ahead
      auto fetch = [&](int w, uint32_t t, uint32_t& dword) {
        const uint32_t o = AFX_SECVAR_BITS * ((uint32_t)w + e.term[t].win_off);
        dword = digit_ws[((size_t)(e.dslot + t) * AFX_DIGIT_WORDS + (o >> 5)) * count + item];
        const int32_t* table = table_ws + (size_t)e.term[t].table_slot * count * AFX_VAR_TABLE_DWORDS + (size_t)item * 4;
        const size_t piece = (size_t)count * 4, entry = (size_t)count * AFX_TABLE_ENTRY_DWORDS;
        // roles 0, 1: pieces 0, 1 and 2, 3; role 2 (and 3, which drops them): pieces 4, 5 twice - uniform offsets, every field set.
        // Cached entries (job->narrow == 2: Y+X | Y-X | 2Z | 2dT, k_msm_tables<TABLE_NARROW_CACHED>): role 2 takes 2dT (6, 7), role 3 2Z (4, 5)
        const size_t first = role < 2 ? 0 : (cached && role == 2) ? 6 : 4, other = role < 2 ? 2 : first;
        quad_narrow q;
#pragma unroll
        for (int m = 0; m < 2; m++)
#pragma unroll
          for (int h = 0; h < 2; h++) {
            q.a[m][h] = *reinterpret_cast<const uint4*>(table + m * entry + (first + h) * piece);
            q.b[m][h] = *reinterpret_cast<const uint4*>(table + m * entry + (other + h) * piece);
          }
        return q;
      };
      static_assert(AFX_SECVAR_BITS == 2 && AFX_SECVAR_STORED == 2, "k_msm_quad's narrow chain reads the two stored entries of a 2-bit window");
      uint32_t nword = 0;
      const int top = (int)job->wins - 1;   // (a segment's share of the windows, or all of them)
      quad_narrow nxt = fetch(top, 0, nword);
#pragma unroll 1
      for (int w = top; w >= 0; w--) {
        if (w != top) {
#pragma unroll 1
          for (int k = 0; k < AFX_SECVAR_BITS; k++) acc = quad_dbl(L, buf, role, lane, acc);
        }
#pragma unroll 1
        for (uint32_t t = 0; t < nv; t++) {
          const quad_narrow cur = nxt;
          const uint32_t word = nword;
          const bool last = t + 1 == nv;
          if (!(last && w == 0)) nxt = fetch(last ? w - 1 : w, last ? 0 : t + 1, nword);
          const int d = (int)((word >> ((AFX_SECVAR_BITS * ((uint32_t)w + e.term[t].win_off)) & 31u)) & 3u) - 2;
          const uint32_t idx = (uint32_t)(d < 0 ? -d : d);
          const bool neg = (d < 0) != (e.term[t].negate != 0);
          // the 8 words this role multiplies by: of entry idx (the identity for 0); roles 0, 1: the half the sign names (role 0
          // takes (y-x)/2 when subtracting, role 1 when adding).  Mask arithmetic, no branch and no address from the digit.
          const uint32_t second = quad_mask(role < 2 && (role == 0 ? neg : !neg));
          // the identity: (1/2, 1/2, 0) with D = Z for the halved affine form, (1, 1, 0, 2) for the cached one.  Role 3 multiplies Z
          // by the entry's 2Z - or, affine entries, by the constant 1, so that both forms take the same addition (its product is on
          // the fourth wave, which the affine form would leave idle)
          uint32_t sel[8];
#pragma unroll
          for (int i = 0; i < 8; i++) sel[i] = (role < 2 && !cached) ? AFX_IDENTITY_NIELS[i] : 0u;
          if (role == 3 || (role < 2 && cached)) sel[0] = (role == 3 && cached) ? 2u : 1u;
          const uint32_t takes = quad_mask(role != 3 || cached);   // (uniform) an affine entry has nothing for role 3
#pragma unroll
          for (uint32_t m = 0; m < 2; m++) {
            const uint32_t hit = quad_mask(idx == m + 1) & takes;
#pragma unroll
            for (int h = 0; h < 2; h++) {
              const uint4 a = cur.a[m][h], b = cur.b[m][h];
              sel[4 * h] = quad_pick(hit, quad_pick(second, b.x, a.x), sel[4 * h]);
              sel[4 * h + 1] = quad_pick(hit, quad_pick(second, b.y, a.y), sel[4 * h + 1]);
              sel[4 * h + 2] = quad_pick(hit, quad_pick(second, b.z, a.z), sel[4 * h + 2]);
              sel[4 * h + 3] = quad_pick(hit, quad_pick(second, b.w, a.w), sel[4 * h + 3]);
            }
          }
          acc = quad_add<false>(L, buf, role, lane, acc, fe_frombytes(sel), neg);
        }
      }
    }
  }
  // the entry of addition (w, t), requested ahead of its use: before the window's doublings for t = 0, an addition ahead otherwise
  auto var_entry = [&](int w, uint32_t t, bool& neg) {
    const int32_t* table = table_ws + ((size_t)e.term[t].table_slot * count + item) * AFX_VAR_TABLE_DWORDS;
    const uint32_t wd = (uint32_t)w + e.term[t].win_off;
    const uint32_t word = digit_ws[((size_t)(e.dslot + t) * AFX_DIGIT_WORDS + (wd >> 3)) * count + item];
    const int d = (int)((word >> ((wd & 7) * 4)) & 15u) - 8;
    const uint32_t idx = (uint32_t)(d < 0 ? -d : d);
    neg = (d < 0) != (e.term[t].negate != 0);
    return quad_entry_load(idx ? table + (idx - 1) * AFX_TABLE_ENTRY_DWORDS : AFX_IDENTITY_ENTRY, role, neg);
  };
  bool nneg = false;
  quad_words nxt = {};
  const int wtop = (int)job->wins - 1;   // 63, or a segment's share
  if (nv && !chained) nxt = var_entry(wtop, 0, nneg);
#pragma unroll 1
  for (int w = (nv && !chained) ? wtop : -1; w >= 0; w--) {
    if (w != wtop) {
#pragma unroll 1
      for (int k = 0; k < 4; k++) acc = quad_dbl(L, buf, role, lane, acc);
    }
#pragma unroll 1
    for (uint32_t t = 0; t < nv; t++) {
      const quad_words cur = nxt;
      const bool neg = nneg;
      const bool last = t + 1 == nv;
      if (!(last && w == 0)) nxt = var_entry(last ? w - 1 : w, last ? 0 : t + 1, nneg);
      acc = quad_add<false>(L, buf, role, lane, acc, quad_entry_fe(cur), neg);
    }
  }
  // fixed bases with secret scalars: the lane exchange, a role's limbs of the multiple
  if constexpr (SEC) {
    bool any_secret = false;
#pragma unroll 1
    for (uint32_t t = nv; t < nt; t++) any_secret |= e.term[t].secret != 0;
#pragma unroll 1
    for (uint32_t j = 0; j < AFX_SEC_WINDOWS && any_secret; j++) {
#pragma unroll 1
      for (uint32_t t = nv; t < nt; t++) {
        if (!e.term[t].secret) continue;   // uniform
        const uint32_t o = AFX_SEC_BITS * j, k = o >> 5, sh = o & 31u;
        const uint32_t* dw = digit_ws + ((size_t)(e.dslot + t) * AFX_DIGIT_WORDS + k) * count + item;
        uint64_t ww = dw[0];
        if (sh + AFX_SEC_BITS > 32) ww |= (uint64_t)dw[count] << 32;
        const int d = (int)((uint32_t)(ww >> sh) & ((1u << AFX_SEC_BITS) - 1)) - (1 << (AFX_SEC_BITS - 1));
        const uint32_t idx = (uint32_t)(d < 0 ? -d : d);
        const bool neg = (d < 0) != (e.term[t].negate != 0);
        const int32_t* win = sec_tables + (size_t)e.term[t].fixed_idx * AFX_SEC_TABLE_DWORDS + (size_t)j * AFX_SEC_WINDOW_DWORDS;
        const int32_t* mine = win + ((lane & 31u) + 1u) * AFX_NIELS_DWORDS;   // this lane's multiple: an address made of the lane's id
        const int src = (int)(((idx - 1u) & 31u) << 2);
        const uint32_t keep = quad_mask(idx != 0);
        fe q = fe_zero();
        if (role < 2) {
          const uint32_t second = quad_mask(role == 0 ? neg : !neg);
#pragma unroll
          for (int l = 0; l < AFX_FE_LIMBS; l++) {
            const uint32_t a = (uint32_t)__builtin_amdgcn_ds_bpermute(src, mine[l]), b = (uint32_t)__builtin_amdgcn_ds_bpermute(src, mine[9 + l]);
            // entry 0, the identity, has (y+x)/2 = (y-x)/2: one wave-uniform limb serves both halves
            q.v[l] = (int32_t)quad_pick(keep, quad_pick(second, b, a), (uint32_t)win[l]);
          }
        } else if (role == 2) {
#pragma unroll
          for (int l = 0; l < AFX_FE_LIMBS; l++) q.v[l] = (int32_t)quad_pick(keep, (uint32_t)__builtin_amdgcn_ds_bpermute(src, mine[18 + l]), (uint32_t)win[18 + l]);
        }
        acc = quad_add<true>(L, buf, role, lane, acc, q, neg);
      }
    }
  }
  // fixed bases: the same one-ahead request of the 9 limbs this role multiplies by ((y+x)/2 | (y-x)/2 | dxy; role 3 reads what
  // role 2 does and drops it)
  auto pos_entry = [&](uint32_t j, uint32_t t, bool& neg, fe& q) {
    const uint32_t o = AFX_POS_BITS * j, k = o >> 5, sh = o & 31u;
    const uint32_t* dw = digit_ws + ((size_t)(e.dslot + t) * AFX_DIGIT_WORDS + k) * count + item;
    uint64_t ww = dw[0];
    if (sh + AFX_POS_BITS > 32) ww |= (uint64_t)dw[count] << 32;
    const int d = (int)((uint32_t)(ww >> sh) & ((1u << AFX_POS_BITS) - 1)) - (1 << (AFX_POS_BITS - 1));
    const uint32_t idx = (uint32_t)(d < 0 ? -d : d);
    neg = (d < 0) != (e.term[t].negate != 0);
    const int32_t* ent = pos_tables + (size_t)e.term[t].fixed_idx * AFX_POS_TABLE_DWORDS + (size_t)j * AFX_POS_WINDOW_DWORDS + idx * AFX_NIELS_DWORDS;
    const uint32_t part = role == 0 ? (neg ? 1u : 0u) : role == 1 ? (neg ? 0u : 1u) : 2u;
#pragma unroll
    for (int l = 0; l < AFX_FE_LIMBS; l++) q.v[l] = ent[9 * part + l];
  };
  // (the public fixed-base terms of the job: all of them unless SEC)
  uint32_t first_pub = nt, last_pub = nt;
#pragma unroll 1
  for (uint32_t t = nv; t < nt; t++)
    if (!SEC || !e.term[t].secret) { if (first_pub == nt) first_pub = t; last_pub = t; }
  auto next_pub = [&](uint32_t t) { t++; while (SEC && t < nt && e.term[t].secret) t++; return t; };
  if (first_pub != nt) {
    fe qn;
    pos_entry(0, first_pub, nneg, qn);
#pragma unroll 1
    for (uint32_t j = 0; j < AFX_POS_WINDOWS; j++) {
#pragma unroll 1
      for (uint32_t t = first_pub; t <= last_pub; t = next_pub(t)) {
        const fe q = qn;
        const bool neg = nneg;
        const bool last = t == last_pub;
        if (!(last && j + 1 == AFX_POS_WINDOWS)) pos_entry(last ? j + 1 : j, last ? first_pub : next_pub(t), nneg, qn);
        acc = quad_add<true>(L, buf, role, lane, acc, q, neg);
      }
    }
  }
  if (role == 0) msm_finish<false>(job, acc, bad, count, item);
}
template <bool SEC>
__global__ void __launch_bounds__(256, 2)
k_msm_quad(const afx_msm_djob* __restrict__ jobs, const int32_t* __restrict__ pos_tables, const int32_t* __restrict__ sec_tables, int32_t* __restrict__ table_ws,
           uint32_t* __restrict__ digit_ws, uint32_t* __restrict__ bad, uint32_t count) {
  msm_quad_body<SEC>(&jobs[blockIdx.y], pos_tables, sec_tables, table_ws, digit_ws, bad, count);
}
template <bool SEC>
__global__ void __launch_bounds__(256, 2)
k_msm_quad_rows(const uint8_t* __restrict__ blob, const int32_t* __restrict__ pos_tables, const int32_t* __restrict__ sec_tables, const afx_row* __restrict__ rows,
                const afx_pass* __restrict__ passes) {
  const afx_row row = rows[blockIdx.y];      // block-uniform: scalar loads
  const afx_pass pass = passes[row.pass];
  msm_quad_body<SEC>(reinterpret_cast<const afx_msm_djob*>(blob + row.job_off), pos_tables, sec_tables, pass.table_ws, pass.digit_ws, pass.bad, pass.count);
}

// k_pointsum for the same idle device: a job's partial sums added up by four waves - role r takes parts r, r + 4, ..., each
// requesting its next part while it adds the current one, then two levels through LDS (1 -> 0 and 3 -> 2, then 2 -> 0) - instead
// of one lane adding up to fifteen parts in a row, each behind its own load.  Role 0 adds the addend and finishes as k_pointsum does.
AFX_DEV void quad_put_point(quad_lds& L, int buf, uint32_t lane, const ge_p3& p) {
  quad_put(L, buf, 0, lane, p.X); quad_put(L, buf, 1, lane, p.Y); quad_put(L, buf, 2, lane, p.Z); quad_put(L, buf, 3, lane, p.T);
}
AFX_DEV ge_p3 quad_get_point(const quad_lds& L, int buf, uint32_t lane) {
  ge_p3 p;
  p.X = quad_get(L, buf, 0, lane); p.Y = quad_get(L, buf, 1, lane); p.Z = quad_get(L, buf, 2, lane); p.T = quad_get(L, buf, 3, lane);
  return p;
}
__global__ void __launch_bounds__(256, 2)
k_pointsum_quad(const afx_pointsum_job* __restrict__ jobs, const afx_row* __restrict__ rows, const afx_pass* __restrict__ passes) {
  __shared__ quad_lds L;
  const afx_pointsum_job job = *row_job(jobs, rows);
  const afx_pass pass = passes[row_pass_index(rows)];   // block-uniform: scalar loads
  const uint32_t count = pass.count;
  uint32_t* __restrict__ bad = pass.bad;
  if (blockIdx.x * 64u >= count) return;   // block-uniform
  const uint32_t lane = threadIdx.x & 63u, role = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const uint32_t item = min(blockIdx.x * 64u + lane, count - 1);   // lanes past the end shadow the last item
  ge_p3 acc = ge_identity();
  if (role < job.n_parts) {
    acc = var_load(job.parts[role], count, item);
    ge_p3 nxt = acc;
    if (role + 4 < job.n_parts) nxt = var_load(job.parts[role + 4], count, item);
#pragma unroll 1
    for (uint32_t k = role + 4; k < job.n_parts; k += 4) {
      const ge_p3 cur = nxt;
      if (k + 4 < job.n_parts) nxt = var_load(job.parts[k + 4], count, item);
      acc = ge_p1p1_to_p3(ge_add_cached(acc, ge_p3_to_cached(cur), false));
    }
  }
  // (fewer than 3 parts: the upper roles hold the identity, whose addition changes nothing)
  if (role & 1u) quad_put_point(L, (int)(role >> 1), lane, acc);
  __syncthreads();
  if (!(role & 1u)) acc = ge_p1p1_to_p3(ge_add_cached(acc, ge_p3_to_cached(quad_get_point(L, (int)(role >> 1), lane)), false));
  __syncthreads();
  if (role == 2) quad_put_point(L, 0, lane, acc);
  __syncthreads();
  if (role != 0) return;
  acc = ge_p1p1_to_p3(ge_add_cached(acc, ge_p3_to_cached(quad_get_point(L, 0, lane)), false));
  if (job.addend) acc = ge_p1p1_to_p3(ge_add_cached(acc, ge_p3_to_cached(var_load(job.addend, count, item)), job.addend_negate != 0));
  if (job.out_var) var_store(job.out_var, count, item, acc);
  if (job.half_var) { var_store(job.half_var, count, item, acc); return; }   // encoded by k_compress2x
  if (job.out_enc) {
    uint32_t w[8];
    ristretto_encode(w, acc);
    enc_store(job.out_enc, item, w);
    if (job.reject_identity && is_identity_encoding(w)) atomicOr(&bad[item], AFX_BAD_IDENTITY);
  }
}

// k_pointsum for a job of MANY parts in a pass of few items (a segmenting pass: eight segments per base and a chain per generator
// make 80-odd parts of an issuance's commitment): a block per (job, item), a lane per part, and a tree through LDS - seven
// additions deep for up to 128 parts where each role of k_pointsum_quad adds twenty in a row.  Lane 0 finishes as k_pointsum does.
__global__ void __launch_bounds__(256)
k_pointsum_tree(const afx_pointsum_job* __restrict__ jobs, const afx_row* __restrict__ rows, const afx_pass* __restrict__ passes) {
  __shared__ int32_t S[4 * AFX_FE_LIMBS][128];
  const afx_pointsum_job job = *row_job(jobs, rows);
  const afx_pass pass = passes[row_pass_index(rows)];   // block-uniform: scalar loads
  const uint32_t count = pass.count, item = blockIdx.x;
  uint32_t* __restrict__ bad = pass.bad;
  if (item >= count) return;   // block-uniform
  const uint32_t tid = threadIdx.x, n = min(job.n_parts, 256u);
  ge_p3 acc = ge_identity();
  if (tid < job.n_parts) acc = var_load(job.parts[tid], count, item);
#pragma unroll 1
  for (uint32_t k = tid + 256u; k < job.n_parts; k += 256u) acc = ge_p1p1_to_p3(ge_add_cached(acc, ge_p3_to_cached(var_load(job.parts[k], count, item)), false));
#pragma unroll 1
  for (uint32_t stride = 128; stride >= 1; stride >>= 1) {
    if (stride >= n) continue;   // (uniform) nothing up there
    if (tid >= stride && tid < 2 * stride && tid < n) {
#pragma unroll
      for (int l = 0; l < AFX_FE_LIMBS; l++) {
        S[l][tid - stride] = acc.X.v[l]; S[AFX_FE_LIMBS + l][tid - stride] = acc.Y.v[l];
        S[2 * AFX_FE_LIMBS + l][tid - stride] = acc.Z.v[l]; S[3 * AFX_FE_LIMBS + l][tid - stride] = acc.T.v[l];
      }
    }
    __syncthreads();
    if (tid < stride && tid + stride < n) {
      ge_p3 o;
#pragma unroll
      for (int l = 0; l < AFX_FE_LIMBS; l++) {
        o.X.v[l] = S[l][tid]; o.Y.v[l] = S[AFX_FE_LIMBS + l][tid]; o.Z.v[l] = S[2 * AFX_FE_LIMBS + l][tid]; o.T.v[l] = S[3 * AFX_FE_LIMBS + l][tid];
      }
      acc = ge_p1p1_to_p3(ge_add_cached(acc, ge_p3_to_cached(o), false));
    }
    __syncthreads();
  }
  if (tid != 0) return;
  if (job.addend) acc = ge_p1p1_to_p3(ge_add_cached(acc, ge_p3_to_cached(var_load(job.addend, count, item)), job.addend_negate != 0));
  if (job.out_var) var_store(job.out_var, count, item, acc);
  if (job.half_var) { var_store(job.half_var, count, item, acc); return; }   // encoded by k_compress2x
  if (job.out_enc) {
    uint32_t w[8];
    ristretto_encode(w, acc);
    enc_store(job.out_enc, item, w);
    if (job.reject_identity && is_identity_encoding(w)) atomicOr(&bad[item], AFX_BAD_IDENTITY);
  }
}

// k_powers: src * 2^step, * 2^(2 step), ... (plan.h afx_powers_job), four waves per item chain like k_msm_quad - these doublings are
// what a small prover pass waits for once instead of in every stage (Assembler::msm_split cuts a secret scalar on a per-item base
// into segments over these points).  Nothing here depends on a secret.
__global__ void __launch_bounds__(256, 2)
k_powers_quad(const afx_powers_job* __restrict__ jobs, const afx_row* __restrict__ rows, const afx_pass* __restrict__ passes) {
  __shared__ quad_lds L;
  const afx_powers_job job = *row_job(jobs, rows);
  const afx_pass pass = passes[row_pass_index(rows)];   // block-uniform: scalar loads
  const uint32_t count = pass.count;
  if (blockIdx.x * 64u >= count) return;   // block-uniform
  const uint32_t lane = threadIdx.x & 63u, role = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const uint32_t item = min(blockIdx.x * 64u + lane, count - 1);   // lanes past the end shadow the last item
  ge_p3 acc = var_load(job.src, count, item);
  int buf = 0;
#pragma unroll 1
  for (uint32_t i = 0; i < job.n_out; i++) {
#pragma unroll 1
    for (uint32_t k = 0; k < job.step; k++) acc = quad_dbl(L, buf, role, lane, acc);
    if (role == 0) var_store(job.out[i], count, item, acc);
  }
}

// ---------------------------------------------------------------------------------------------
// k_compress2x: the encodings of 2*P_j for all the points P_j an item's jobs left in half_var, with ONE field inversion per item
// ---------------------------------------------------------------------------------------------
// RistrettoPoint::compress needs an inverse square root per point (254 squarings).  The encoding of TWICE a point does not: with
//   e = 2XY, f = Z^2 + dT^2, g = Y^2 + X^2, h = Z^2 - dT^2     (2P = (e*f : g*h : f*g : e*h) in extended coordinates)
// it is |(h' - g') * magic * g' / (f*h)| after two sign/rotation decisions that only need 1/(e*g) and 1/(f*h) - plain inversions,
// which Montgomery's trick shares (curve25519-dalek's RistrettoPoint::double_and_compress_batch [3P] is the same computation).
// So a job whose result is only ever ENCODED (every recomputed / fresh commitment of a Schnorr proof) runs on halved scalars
// (msm_recode) and leaves R/2 here; lane = item walks its jobs twice: prefix products of e*f*g*h forwards, one inversion, the
// encodings backwards.  e*f*g*h = 0 exactly for the identity's representatives (X = 0 or Y = 0; f, g, h never vanish on the
// even subgroup), whose encoding is all zeros; such a factor is left out of the product.  tests/: every byte-parity test of the
// commitments and challenges goes through this kernel; tests/pyref checks the formula against encode(P + P).
// Grid row y walks the jobs [y * per_row, (y + 1) * per_row) of the item: one row (and one inversion per item) for large passes;
// small passes, where the serial walk is what a call waits for, spread an item's commitments over up to 8 rows.
__global__ void __launch_bounds__(AFX_BLOCK, 2)
k_compress2x(const afx_compress_job* __restrict__ jobs, const afx_walk_row* __restrict__ rows, const afx_pass* __restrict__ passes) {
  const afx_walk_row row = rows[blockIdx.y];   // wave-uniform
  const afx_pass pass = passes[row.pass];
  const uint32_t count = pass.count;
  uint32_t* __restrict__ bad = pass.bad;
  const uint32_t item = blockIdx.x * blockDim.x + threadIdx.x;
  if (item >= count) return;
  const uint32_t njobs = row.n_jobs;
  jobs = reinterpret_cast<decltype(jobs)>(reinterpret_cast<const uint8_t*>(jobs) + row.job_off);
  int32_t* __restrict__ prefix_ws = row.prefix_ws;
  fe prod = fe_one();
  bool halves = false;   // (uniform) the row has jobs that share the inversion
#pragma unroll 1
  for (uint32_t j = 0; j < njobs; j++) {
    ge_p3 P = var_load(jobs[j].var, count, item);
    if (jobs[j].negate == AFX_COMPRESS_PLAIN) {   // (uniform) the point itself, with its own inverse square root: no part of the shared inversion
      uint32_t w[8];
      ristretto_encode(w, P);
      enc_store(jobs[j].out_enc, item, w);
      if (jobs[j].reject_identity && is_identity_encoding(w)) atomicOr(&bad[item], AFX_BAD_IDENTITY);
      continue;
    }
    if (jobs[j].negate) P = ge_neg(P);
    halves = true;
    const c2x_state s = c2x_from(P);
    fe_store_soa(prefix_ws + (size_t)j * AFX_FE_LIMBS * count, 0, count, item, prod);   // product of the factors before j
    prod = fe_mul(prod, s.efgh);
  }
  if (!halves) return;
  fe inv = fe_invert(prod);
#pragma unroll 1
  for (uint32_t jj = njobs; jj > 0; jj--) {
    const uint32_t j = jj - 1;
    if (jobs[j].negate == AFX_COMPRESS_PLAIN) continue;
    ge_p3 P = var_load(jobs[j].var, count, item);
    if (jobs[j].negate) P = ge_neg(P);
    const c2x_state s = c2x_from(P);
    const fe inv_j = fe_mul(inv, fe_load_soa(prefix_ws + (size_t)j * AFX_FE_LIMBS * count, 0, count, item));   // 1 / (e f g h)_j
    inv = fe_mul(inv, s.efgh);
    uint32_t w[8];
    c2x_finish(w, s, inv_j);
    enc_store(jobs[j].out_enc, item, w);
    if (jobs[j].reject_identity && is_identity_encoding(w)) atomicOr(&bad[item], AFX_BAD_IDENTITY);
  }
}

// k_negenc: the encodings of -P_j for decoded points P_j (the "-E1" of every proof of encryption an item carries,
// encryption.rs:185) with one field inversion per item instead of an inverse square root per point (ge.cuh negenc_*).  Same
// two-pass walk as k_compress2x; a point whose factor is zero (the identity, or the placeholder of a failed decode - the item
// is rejected either way) is left out of the product and encodes to zeros.
__global__ void __launch_bounds__(AFX_BLOCK, 2)
k_negenc(const afx_negenc_job* __restrict__ jobs, const afx_walk_row* __restrict__ rows, const afx_pass* __restrict__ passes) {
  const afx_walk_row row = rows[blockIdx.y];   // wave-uniform
  const afx_pass pass = passes[row.pass];
  const uint32_t count = pass.count;
  uint32_t* __restrict__ bad = pass.bad;
  const uint32_t item = blockIdx.x * blockDim.x + threadIdx.x;
  if (item >= count) return;
  const uint32_t njobs = row.n_jobs;
  jobs = reinterpret_cast<decltype(jobs)>(reinterpret_cast<const uint8_t*>(jobs) + row.job_off);
  int32_t* __restrict__ prefix_ws = row.prefix_ws;
  fe prod = fe_one();
#pragma unroll 1
  for (uint32_t j = 0; j < njobs; j++) {
    uint32_t w[8];
    enc_load(w, jobs[j].enc, item);
    fe den = negenc_den(fe_frombytes(w), var_load(jobs[j].var, count, item));
    fe_cmov(den, fe_one(), fe_is_zero(den));
    fe_store_soa(prefix_ws + (size_t)j * AFX_FE_LIMBS * count, 0, count, item, prod);
    prod = fe_mul(prod, den);
  }
  fe inv = fe_invert(prod);
#pragma unroll 1
  for (uint32_t jj = njobs; jj > 0; jj--) {
    const uint32_t j = jj - 1;
    uint32_t w[8];
    enc_load(w, jobs[j].enc, item);
    const fe s = fe_frombytes(w);
    const ge_p3 P = var_load(jobs[j].var, count, item);
    fe den = negenc_den(s, P);
    const bool zero = fe_is_zero(den);
    fe_cmov(den, fe_one(), zero);
    const fe inv_j = fe_mul(inv, fe_load_soa(prefix_ws + (size_t)j * AFX_FE_LIMBS * count, 0, count, item));
    inv = fe_mul(inv, den);
    uint32_t o[8];
    negenc_finish(o, s, P, inv_j);
#pragma unroll
    for (int i = 0; i < 8; i++) o[i] = zero ? 0u : o[i];
    enc_store(jobs[j].out_enc, item, o);
    if (jobs[j].reject_identity && is_identity_encoding(o)) atomicOr(&bad[item], AFX_BAD_IDENTITY);
  }
}

// k_pointsum: out = sum of the partial results of a job that Assembler::msm_split cut into one chain per term (+- addend)
__global__ void __launch_bounds__(AFX_BLOCK, 2) k_pointsum(const afx_pointsum_job* __restrict__ jobs, const afx_row* __restrict__ rows, const afx_pass* __restrict__ passes) {
  const afx_pointsum_job job = *row_job(jobs, rows);
  const afx_pass pass = passes[row_pass_index(rows)];   // wave-uniform: scalar loads
  const uint32_t count = pass.count;
  uint32_t* __restrict__ bad = pass.bad;
  const uint32_t item = blockIdx.x * blockDim.x + threadIdx.x;
  if (item >= count) return;
  ge_p3 acc = var_load(job.parts[0], count, item);
#pragma unroll 1
  for (uint32_t k = 1; k < job.n_parts; k++) acc = ge_p1p1_to_p3(ge_add_cached(acc, ge_p3_to_cached(var_load(job.parts[k], count, item)), false));
  if (job.addend) acc = ge_p1p1_to_p3(ge_add_cached(acc, ge_p3_to_cached(var_load(job.addend, count, item)), job.addend_negate != 0));
  if (job.out_var) var_store(job.out_var, count, item, acc);
  if (job.half_var) { var_store(job.half_var, count, item, acc); return; }   // encoded by k_compress2x
  if (job.out_enc) {
    uint32_t w[8];
    ristretto_encode(w, acc);
    enc_store(job.out_enc, item, w);
    if (job.reject_identity && is_identity_encoding(w)) atomicOr(&bad[item], AFX_BAD_IDENTITY);
  }
}

// ---------------------------------------------------------------------------------------------
// k_hash: STROBE-128 / merlin transcripts
// ---------------------------------------------------------------------------------------------
AFX_DEV uint64_t load_u64(const uint8_t* p) {
  const uint2 v = *reinterpret_cast<const uint2*>(p);
  return (uint64_t)v.x | ((uint64_t)v.y << 32);
}

__global__ void __launch_bounds__(AFX_BLOCK) k_hash(const afx_hash_program* __restrict__ progs, const afx_row* __restrict__ rows, const afx_pass* __restrict__ passes) {
  const afx_hash_program* prog = row_job(progs, rows);
  const afx_pass pass = passes[row_pass_index(rows)];   // wave-uniform: scalar loads
  const uint32_t count = pass.count;
  uint32_t* __restrict__ bad = pass.bad;
  const uint32_t item = blockIdx.x * blockDim.x + threadIdx.x;
  if (item >= count) return;
  uint64_t st[25];
  if (prog->load_state) {
#pragma unroll
    for (int i = 0; i < 25; i++) st[i] = prog->load_state[(size_t)i * count + item];
  } else {
#pragma unroll
    for (int i = 0; i < 25; i++) st[i] = prog->init_state[i];
  }
  const uint32_t nrec = prog->n_records;
#pragma unroll 1
  for (uint32_t r = 0; r < nrec; r++) {
    const afx_hash_record* rec = &prog->records[r];
#pragma unroll
    for (int w = 0; w < 21; w++) {
      const afx_hash_word hw = rec->w[w];
      uint64_t v = hw.c;
      if (hw.field >= 0) {
        const uint8_t* f = prog->fields[hw.field] + 32ull * item;
        const int q = hw.q;
        const uint64_t lo = (q >= 0) ? load_u64(f + 8 * q) : 0ull;
        const uint64_t hi = (q < 3) ? load_u64(f + 8 * (q + 1)) : 0ull;
        const uint32_t sh = 8u * hw.r;
        const uint64_t val = sh ? ((lo >> sh) | (hi << (64u - sh))) : lo;
        v ^= val & hw.fmask;
      }
      st[w] = (st[w] & hw.keep) ^ v;
    }
    keccak_f1600(st);
    if (rec->squeeze == AFX_SQ_WIDE_OUT) {   // uniform
      uint64_t* o = reinterpret_cast<uint64_t*>(prog->outs[rec->squeeze_out] + 64ull * item);
#pragma unroll
      for (int i = 0; i < 8; i++) o[i] = st[i];
    } else if (rec->squeeze != AFX_SQ_NONE) {
      uint32_t x[16];
#pragma unroll
      for (int i = 0; i < 8; i++) { x[2 * i] = (uint32_t)st[i]; x[2 * i + 1] = (uint32_t)(st[i] >> 32); }
      const sc c = sc_reduce512(x);
      if (rec->squeeze == AFX_SQ_CHALLENGE_COMPARE) {
        const sc want = sc_load_item(prog->challenge, 32, item);
        if (!sc_eq(c, want)) atomicOr(&bad[item], AFX_BAD_CHALLENGE);
        if (prog->trace) {
          uint32_t w8[8];
#pragma unroll
          for (int i = 0; i < 8; i++) w8[i] = c.v[i];
          enc_store(prog->trace, item, w8);
        }
      } else {
        uint32_t w8[8];
#pragma unroll
        for (int i = 0; i < 8; i++) w8[i] = c.v[i];
        enc_store(prog->outs[rec->squeeze_out], item, w8);
      }
    }
  }
  if (prog->save_state) {
#pragma unroll
    for (int i = 0; i < 25; i++) prog->save_state[(size_t)i * count + item] = st[i];
  }
}

// k_hash_coop: the same transcripts for SMALL passes, 32 lanes per (item, program).  One lane per item runs its 20-80
// permutations one after the other - 13.6 us each on an otherwise idle SIMD, 1.1 ms of the 2.9 ms a small issue call takes - and
// a sponge cannot be cut into independent pieces; what can be spread is the permutation itself.  Lane w < 25 of a group holds word
// w = x + 5y of the state: theta's column parities, the rho/pi move and chi's row neighbours are lane moves (round 4: 3 64-bit
// shuffles and 12 DPP moves a round, below; 9 shuffles before), every lane absorbs its own word of a rate block (afx_hash_word is per word already), lane 0 reduces what is squeezed.
// Only while the device has lanes to spare: the engine launches it for at most AFX_HASH_COOP_GROUPS (item, program) pairs (engine.cpp).
__device__ __constant__ const uint8_t KECCAK_RHO[25] = { 0, 1, 62, 28, 27, 36, 44, 6, 55, 20, 3, 10, 43, 25, 39, 41, 45, 15, 21, 8, 18, 2, 61, 56, 14 };
// word w of the state after rho+pi comes from word KECCAK_PI_SRC[w] (b[y + 5*((2x+3y)%5)] = rotl(a[x+5y]), keccak.cuh)
__device__ __constant__ const uint8_t KECCAK_PI_SRC[25] = { 0, 6, 12, 18, 24, 3, 9, 10, 16, 22, 1, 7, 13, 19, 20, 4, 5, 11, 17, 23, 2, 8, 14, 15, 21 };
AFX_DEV uint64_t shfl64(uint64_t v, uint32_t src) {
  const uint32_t lo = __shfl((uint32_t)v, (int)src, 32), hi = __shfl((uint32_t)(v >> 32), (int)src, 32);
  return (uint64_t)lo | ((uint64_t)hi << 32);
}
// The words of a row of five sit in five neighbouring lanes INSIDE one 16-lane DPP row (lane 15 of a group is skipped: words 0..14
// in lanes 0..14, words 15..24 in lanes 16..25), so that theta's and chi's row neighbours are data-parallel-primitive moves at
// VALU rate - a shift by the distance, or by the distance minus five where the row wraps - instead of trips through the LDS
// crossbar, of which a lone wave gets one per ~38 cycles; the column parities need two shuffles (their halves inside the two DPP rows are
// shifts as well) and pi one: 3 of the 9 a round.
AFX_DEV uint32_t kc_lane(uint32_t w) { return w + (w >= 15u ? 1u : 0u); }
template <int CTRL>
AFX_DEV uint64_t kc_row_shift(uint64_t v) {
  const uint32_t lo = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)v, CTRL, 0xf, 0xf, true), hi = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)(v >> 32), CTRL, 0xf, 0xf, true);
  return (uint64_t)lo | ((uint64_t)hi << 32);
}
template <int NEAR, int WRAP>
AFX_DEV uint64_t kc_row_fetch(uint64_t v, bool wraps) {
  const int lo = (int)(uint32_t)v, hi = (int)(uint32_t)(v >> 32);
  const uint32_t nl = (uint32_t)__builtin_amdgcn_update_dpp(0, lo, NEAR, 0xf, 0xf, true), nh = (uint32_t)__builtin_amdgcn_update_dpp(0, hi, NEAR, 0xf, 0xf, true);
  const uint32_t wl = (uint32_t)__builtin_amdgcn_update_dpp(0, lo, WRAP, 0xf, 0xf, true), wh = (uint32_t)__builtin_amdgcn_update_dpp(0, hi, WRAP, 0xf, 0xf, true);
  return (uint64_t)(wraps ? wl : nl) | ((uint64_t)(wraps ? wh : nh) << 32);
}
enum { KC_SHL1 = 0x101, KC_SHL2 = 0x102, KC_SHL4 = 0x104, KC_SHL5 = 0x105, KC_SHL10 = 0x10a, KC_SHR1 = 0x111, KC_SHR3 = 0x113, KC_SHR4 = 0x114 };   // row_shl:n = 0x100 + n (lane i reads lane i + n), row_shr:n = 0x110 + n
__global__ void __launch_bounds__(AFX_BLOCK) k_hash_coop(const afx_hash_program* __restrict__ progs, const afx_row* __restrict__ rows, const afx_pass* __restrict__ passes) {
  const afx_hash_program* prog = row_job(progs, rows);
  const afx_pass pass = passes[row_pass_index(rows)];   // wave-uniform: scalar loads
  const uint32_t count = pass.count;
  uint32_t* __restrict__ bad = pass.bad;
  if (blockIdx.x * (blockDim.x >> 5) + (((uint32_t)__builtin_amdgcn_readfirstlane((int)threadIdx.x) >> 6) << 1) >= count) return;   // a wave (two lane groups) past the end of this row's pass retires
  const uint32_t g = threadIdx.x & 31u;
  const uint32_t group = blockIdx.x * (blockDim.x >> 5) + (threadIdx.x >> 5);
  const bool live = group < count;                      // a whole group is live or not; dead groups shadow the last item and store nothing
  const uint32_t item = live ? group : count - 1;
  const bool holds = g < 15u || (g >= 16u && g <= 25u);   // lanes 15 and 26..31 take part in the shuffles only (never as a source)
  const uint32_t w = !holds ? 24u : g < 15u ? g : g - 1u, x = w % 5u;
  const uint32_t rho = KECCAK_RHO[w], pi_src = kc_lane(KECCAK_PI_SRC[w]);
  uint64_t st = prog->load_state ? prog->load_state[(size_t)w * count + item] : prog->init_state[w];
  const uint32_t nrec = prog->n_records;
  // A record's rate word and the 16 bytes of the field it takes from are requested while the record BEFORE it permutes: three
  // dependent loads (the word, the field's address, its bytes) that otherwise stand between two permutations, about as long as a
  // permutation's own 24 rounds.  (No record reads what a squeeze of its own program stored: SchnorrBuilder::make_program checks.)
  const uint32_t wl = w < 21u ? w : 0u;
  afx_hash_word hw_n = prog->records[0].w[wl];
  uint64_t lo_n = 0, hi_n = 0;
  if (w < 21u && hw_n.field >= 0) {
    const uint8_t* f = prog->fields[hw_n.field] + 32ull * item;
    lo_n = (hw_n.q >= 0) ? load_u64(f + 8 * hw_n.q) : 0ull;
    hi_n = (hw_n.q < 3) ? load_u64(f + 8 * (hw_n.q + 1)) : 0ull;
  }
#pragma unroll 1
  for (uint32_t r = 0; r < nrec; r++) {
    const afx_hash_record* rec = &prog->records[r];
    const afx_hash_word hw = hw_n;
    const uint64_t lo = lo_n, hi = hi_n;
    if (r + 1 < nrec) {   // uniform
      hw_n = prog->records[r + 1].w[wl];
      lo_n = 0; hi_n = 0;
      if (w < 21u && hw_n.field >= 0) {
        const uint8_t* f = prog->fields[hw_n.field] + 32ull * item;
        lo_n = (hw_n.q >= 0) ? load_u64(f + 8 * hw_n.q) : 0ull;
        hi_n = (hw_n.q < 3) ? load_u64(f + 8 * (hw_n.q + 1)) : 0ull;
      }
    }
    if (w < 21u) {
      uint64_t v = hw.c;
      if (hw.field >= 0) {
        const uint32_t sh = 8u * hw.r;
        const uint64_t val = sh ? ((lo >> sh) | (hi << (64u - sh))) : lo;
        v ^= val & hw.fmask;
      }
      st = (st & hw.keep) ^ v;
    }
#pragma unroll
    for (int round = 0; round < 24; round++) {   // unrolled: the round constants become literals (a scalar load and its wait a round otherwise)
      // theta: the parity of the column's words inside each DPP row lands on the row's first five lanes (words y = 0, 1, 2 on lanes
      // 0..4; y = 3, 4 on lanes 16..20) by two shifts, and every lane of column x fetches the two halves: 2 shuffles, not 4
      const uint64_t p = st ^ kc_row_shift<KC_SHL5>(st) ^ (g < 16u ? kc_row_shift<KC_SHL10>(st) : 0ull);
      const uint64_t c = shfl64(p, x) ^ shfl64(p, 16u + x);
      const uint64_t cr = kc_row_fetch<KC_SHL1, KC_SHR4>(c, x == 4u);   // column x + 1
      st ^= kc_row_fetch<KC_SHR1, KC_SHL4>(c, x == 0u) ^ ((cr << 1) | (cr >> 63));   // column x - 1
      // rho on the own word, pi as a shuffle
      const uint64_t rot = rho ? ((st << rho) | (st >> (64u - rho))) : st;
      const uint64_t b = shfl64(rot, pi_src);
      // chi along the row, iota on word 0
      st = b ^ (~kc_row_fetch<KC_SHL1, KC_SHR4>(b, x == 4u) & kc_row_fetch<KC_SHL2, KC_SHR3>(b, x >= 3u));
      if (w == 0u) st ^= KECCAK_RC[round];
    }
    if (rec->squeeze == AFX_SQ_WIDE_OUT) {   // uniform: the words go out as they are, a lane each
      if (holds && w < 8u && live) *reinterpret_cast<uint64_t*>(prog->outs[rec->squeeze_out] + 64ull * item + 8u * w) = st;
    } else if (rec->squeeze != AFX_SQ_NONE) {   // uniform: a property of the record
      uint32_t xw[16];
#pragma unroll
      for (int i = 0; i < 8; i++) { const uint64_t t = shfl64(st, (uint32_t)i); xw[2 * i] = (uint32_t)t; xw[2 * i + 1] = (uint32_t)(t >> 32); }
      if (g == 0u && live) {
        const sc c = sc_reduce512(xw);
        uint32_t w8[8];
#pragma unroll
        for (int i = 0; i < 8; i++) w8[i] = c.v[i];
        if (rec->squeeze == AFX_SQ_CHALLENGE_COMPARE) {
          const sc want = sc_load_item(prog->challenge, 32, item);
          if (!sc_eq(c, want)) atomicOr(&bad[item], AFX_BAD_CHALLENGE);
          if (prog->trace) enc_store(prog->trace, item, w8);
        } else {
          enc_store(prog->outs[rec->squeeze_out], item, w8);
        }
      }
    }
  }
  if (prog->save_state && holds && live) prog->save_state[(size_t)w * count + item] = st;
}

// k_hash_coop64: the same with a whole WAVE per (item, program) - for the few transcripts of a very small pass, whose permutations in
// a row are what the call waits for (an issuance's rng squeezes 21 blindings out of one sponge: 48 permutations).  Lanes 0..31 hold
// the LOW 32-bit halves of the state's words in k_hash_coop's layout, lanes 32..63 the high halves: every move of a round is one
// 32-bit move instead of two - three ds_bpermute_b32 a round instead of six, ten DPP moves instead of twenty - and the two places a
// 64-bit rotation needs the other half (theta's rotation by one, rho) fetch it with v_permlane32_swap_b32 (gfx950: lanes 32..63 of
// one register against lanes 0..31 of another, at VALU rate).  (Theta's column parities across the two DPP rows by v_permlane16_swap_b32
// instead of the two shuffles: no faster - 253 against 248 us for an issuance's 48 records; the round is a chain of dependent moves.)  Same schedule, same bytes; afxk_hash_coop picks it while the launch's
// groups leave most of the device idle.
AFX_DEV uint32_t kc_other_half(uint32_t v, bool upper) {
  const auto r = __builtin_amdgcn_permlane32_swap(v, v, false, false);   // r[0]: lanes 32..63 now hold v's lanes 0..31; r[1]: lanes 0..31 hold v's 32..63
  return upper ? r[0] : r[1];
}
template <int CTRL>
AFX_DEV uint32_t kc_shift32(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xf, 0xf, true); }
template <int NEAR, int WRAP>
AFX_DEV uint32_t kc_fetch32(uint32_t v, bool wraps) {
  const uint32_t n = kc_shift32<NEAR>(v), w = kc_shift32<WRAP>(v);
  return wraps ? w : n;
}
__global__ void __launch_bounds__(AFX_BLOCK) k_hash_coop64(const afx_hash_program* __restrict__ progs, const afx_row* __restrict__ rows, const afx_pass* __restrict__ passes) {
  const afx_hash_program* prog = row_job(progs, rows);
  const afx_pass pass = passes[row_pass_index(rows)];   // wave-uniform: scalar loads
  const uint32_t count = pass.count;
  uint32_t* __restrict__ bad = pass.bad;
  const uint32_t group = blockIdx.x * (blockDim.x >> 6) + ((uint32_t)__builtin_amdgcn_readfirstlane((int)threadIdx.x) >> 6);   // a wave per (item, program)
  if (group >= count) return;   // wave-uniform
  const uint32_t item = group;
  const uint32_t L = threadIdx.x & 63u, g = L & 31u;
  const bool upper = L >= 32u;
  const bool holds = g < 15u || (g >= 16u && g <= 25u);   // lanes 15 and 26..31 of each half take part in the moves only (never as a source)
  const uint32_t w = !holds ? 24u : g < 15u ? g : g - 1u, x = w % 5u;
  const uint32_t rho = KECCAK_RHO[w], rs = rho & 31u;
  const bool rho_swaps = rho >= 32u;
  const int half_base = (int)(L & 32u);
  const int a_c0 = 4 * (half_base + (int)x), a_c1 = 4 * (half_base + 16 + (int)x), a_pi = 4 * (half_base + (int)kc_lane(KECCAK_PI_SRC[w]));
  auto half_of = [&](uint64_t v) { return upper ? (uint32_t)(v >> 32) : (uint32_t)v; };
  uint32_t st = half_of(prog->load_state ? prog->load_state[(size_t)w * count + item] : prog->init_state[w]);
  const uint32_t nrec = prog->n_records;
  const uint32_t wl = w < 21u ? w : 0u;
  afx_hash_word hw_n = prog->records[0].w[wl];
  uint64_t lo_n = 0, hi_n = 0;
  if (w < 21u && hw_n.field >= 0) {
    const uint8_t* f = prog->fields[hw_n.field] + 32ull * item;
    lo_n = (hw_n.q >= 0) ? load_u64(f + 8 * hw_n.q) : 0ull;
    hi_n = (hw_n.q < 3) ? load_u64(f + 8 * (hw_n.q + 1)) : 0ull;
  }
#pragma unroll 1
  for (uint32_t r = 0; r < nrec; r++) {
    const afx_hash_record* rec = &prog->records[r];
    const afx_hash_word hw = hw_n;
    const uint64_t lo = lo_n, hi = hi_n;
    if (r + 1 < nrec) {   // uniform: the next record's words, requested while this one permutes
      hw_n = prog->records[r + 1].w[wl];
      lo_n = 0; hi_n = 0;
      if (w < 21u && hw_n.field >= 0) {
        const uint8_t* f = prog->fields[hw_n.field] + 32ull * item;
        lo_n = (hw_n.q >= 0) ? load_u64(f + 8 * hw_n.q) : 0ull;
        hi_n = (hw_n.q < 3) ? load_u64(f + 8 * (hw_n.q + 1)) : 0ull;
      }
    }
    if (w < 21u) {
      uint64_t v = hw.c;
      if (hw.field >= 0) {
        const uint32_t sh = 8u * hw.r;
        const uint64_t val = sh ? ((lo >> sh) | (hi << (64u - sh))) : lo;
        v ^= val & hw.fmask;
      }
      st = (st & half_of(hw.keep)) ^ half_of(v);
    }
#pragma unroll
    for (int round = 0; round < 24; round++) {
      // theta: column parities as in k_hash_coop, one half each; the rotation by one takes the top bit of the OTHER half's parity
      const uint32_t p = st ^ kc_shift32<KC_SHL5>(st) ^ (g < 16u ? kc_shift32<KC_SHL10>(st) : 0u);
      const uint32_t c = (uint32_t)__builtin_amdgcn_ds_bpermute(a_c0, (int)p) ^ (uint32_t)__builtin_amdgcn_ds_bpermute(a_c1, (int)p);
      const uint32_t cr = kc_fetch32<KC_SHL1, KC_SHR4>(c, x == 4u);   // column x + 1, this half
      const uint32_t cro = kc_other_half(cr, upper);                   // ... and the other half
      st ^= kc_fetch32<KC_SHR1, KC_SHL4>(c, x == 0u) ^ ((cr << 1) | (cro >> 31));
      // rho: (A << s) | (B >> (32 - s)) with (A, B) = (mine, other) for a rotation under 32 bits, (other, mine) from 32 on; pi as a shuffle
      const uint32_t so = kc_other_half(st, upper);
      const uint32_t A = rho_swaps ? so : st, B = rho_swaps ? st : so;
      const uint32_t rot = rs ? ((A << rs) | (B >> (32u - rs))) : A;
      const uint32_t b = (uint32_t)__builtin_amdgcn_ds_bpermute(a_pi, (int)rot);
      // chi along the row, iota on word 0
      st = b ^ (~kc_fetch32<KC_SHL1, KC_SHR4>(b, x == 4u) & kc_fetch32<KC_SHL2, KC_SHR3>(b, x >= 3u));
      if (w == 0u) st ^= upper ? (uint32_t)(KECCAK_RC[round] >> 32) : (uint32_t)KECCAK_RC[round];
    }
    if (rec->squeeze == AFX_SQ_WIDE_OUT) {   // uniform: the half words go out as they are, a lane each
      if (holds && w < 8u) *reinterpret_cast<uint32_t*>(prog->outs[rec->squeeze_out] + 64ull * item + 8u * w + (upper ? 4u : 0u)) = st;
    } else if (rec->squeeze != AFX_SQ_NONE) {   // uniform: a property of the record
      uint32_t xw[16];
#pragma unroll
      for (int i = 0; i < 8; i++) {
        xw[2 * i] = (uint32_t)__builtin_amdgcn_ds_bpermute(4 * i, (int)st);
        xw[2 * i + 1] = (uint32_t)__builtin_amdgcn_ds_bpermute(4 * (32 + i), (int)st);
      }
      if (L == 0u) {
        const sc c = sc_reduce512(xw);
        uint32_t w8[8];
#pragma unroll
        for (int i = 0; i < 8; i++) w8[i] = c.v[i];
        if (rec->squeeze == AFX_SQ_CHALLENGE_COMPARE) {
          const sc want = sc_load_item(prog->challenge, 32, item);
          if (!sc_eq(c, want)) atomicOr(&bad[item], AFX_BAD_CHALLENGE);
          if (prog->trace) enc_store(prog->trace, item, w8);
        } else {
          enc_store(prog->outs[rec->squeeze_out], item, w8);
        }
      }
    }
  }
  if (prog->save_state && holds) {
    // (a 64-bit store per word: the low half's lane writes, with the high half fetched from its partner)
    const uint32_t other = kc_other_half(st, upper);
    if (!upper) prog->save_state[(size_t)w * count + item] = (uint64_t)st | ((uint64_t)other << 32);
  }
}

// ---------------------------------------------------------------------------------------------
// status / utilities
// ---------------------------------------------------------------------------------------------
// one grid row per job: the status bytes of one pass / one array to fill (several passes share a launch, engine.cpp)
__global__ void k_finish(const afx_finish_job* __restrict__ jobs, const afx_row* __restrict__ rows) {
  const afx_finish_job job = *row_job(jobs, rows);
  const uint32_t item = blockIdx.x * blockDim.x + threadIdx.x;
  if (item >= job.count) return;
  job.status[item] = job.bad[item] ? (uint8_t)job.fail_code : (uint8_t)0;
}
__global__ void k_fill_u32(const afx_fill_job* __restrict__ jobs, const afx_row* __restrict__ rows) {
  const afx_fill_job job = *row_job(jobs, rows);
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < job.n) job.p[i] = job.v;
}
__global__ void __launch_bounds__(AFX_BLOCK, 2) k_from_uniform(const uint8_t* __restrict__ wide, uint8_t* __restrict__ out_enc, int32_t* out_var, uint32_t count) {
  const uint32_t item = blockIdx.x * blockDim.x + threadIdx.x;
  if (item >= count) return;
  uint32_t w[16];
  enc_load(w, wide, 2 * item);
  enc_load(w + 8, wide, 2 * item + 1);
  const ge_p3 P = ristretto_from_uniform(w);
  if (out_var) var_store(out_var, count, item, P);
  if (out_enc) {
    uint32_t e[8];
    ristretto_encode(e, P);
    enc_store(out_enc, item, e);
  }
}
__global__ void __launch_bounds__(AFX_BLOCK) k_reduce_wide(const uint8_t* __restrict__ wide, uint8_t* __restrict__ out, uint32_t count) {
  const uint32_t item = blockIdx.x * blockDim.x + threadIdx.x;
  if (item >= count) return;
  uint32_t w[16];
  enc_load(w, wide, 2 * item);
  enc_load(w + 8, wide, 2 * item + 1);
  const sc r = sc_reduce512(w);
  uint32_t o[8];
#pragma unroll
  for (int i = 0; i < 8; i++) o[i] = r.v[i];
  enc_store(out, item, o);
}
// the same two for the launches of a plan: one job per grid row, the row's pass gives the item count
__global__ void __launch_bounds__(AFX_BLOCK, 2) k_from_uniform_jobs(const afx_uniform_job* __restrict__ jobs, const afx_row* __restrict__ rows, const afx_pass* __restrict__ passes) {
  const afx_uniform_job job = *row_job(jobs, rows);
  const afx_pass pass = passes[row_pass_index(rows)];
  const uint32_t count = pass.count;
  const uint32_t item = blockIdx.x * blockDim.x + threadIdx.x;
  if (item >= count) return;
  uint32_t w[16];
  enc_load(w, job.wide, 2 * item);
  enc_load(w + 8, job.wide, 2 * item + 1);
  const ge_p3 P = ristretto_from_uniform(w);
  if (job.out_var) var_store(job.out_var, count, item, P);
  if (job.out_enc) {
    uint32_t e[8];
    ristretto_encode(e, P);
    enc_store(job.out_enc, item, e);
  }
}
__global__ void __launch_bounds__(AFX_BLOCK) k_reduce_wide_jobs(const afx_reduce_job* __restrict__ jobs, const afx_row* __restrict__ rows, const afx_pass* __restrict__ passes) {
  const afx_reduce_job job = *row_job(jobs, rows);
  const afx_pass pass = passes[row_pass_index(rows)];
  const uint32_t item = blockIdx.x * blockDim.x + threadIdx.x;
  if (item >= pass.count) return;
  uint32_t w[16];
  enc_load(w, job.wide, 2 * item);
  enc_load(w + 8, job.wide, 2 * item + 1);
  const sc r = sc_reduce512(w);
  uint32_t o[8];
#pragma unroll
  for (int i = 0; i < 8; i++) o[i] = r.v[i];
  enc_store(job.out, item, o);
}
// decode -> ok flag -> re-encode (round-trip parity test of decode+encode)
__global__ void __launch_bounds__(AFX_BLOCK, 2) k_validate(const uint8_t* __restrict__ enc, uint8_t* __restrict__ ok, uint8_t* __restrict__ reenc, uint32_t count) {
  const uint32_t item = blockIdx.x * blockDim.x + threadIdx.x;
  if (item >= count) return;
  uint32_t w[8];
  enc_load(w, enc, item);
  ge_p3 P;
  const bool good = ristretto_decode(P, w);
  ok[item] = good ? 1 : 0;
  if (reenc) {
    uint32_t e[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (good) ristretto_encode(e, P);
    enc_store(reenc, item, e);
  }
}

// wire records (array of structs, `cells` 32-byte cells per item) -> struct-of-arrays rows [row][count][32];
// row_of_cell maps a record cell to its SoA row (revealed attribute values land on their attribute position)
__global__ void __launch_bounds__(AFX_BLOCK) k_aos_to_soa(const uint8_t* __restrict__ rec, uint8_t* __restrict__ soa,
                                                          const uint32_t* __restrict__ row_of_cell, uint32_t cells, uint32_t count) {
  const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;   // t = cell * count + item: writes are coalesced
  if (t >= (uint64_t)cells * count) return;
  const uint32_t cell = (uint32_t)(t / count), item = (uint32_t)(t % count);
  const uint4* src = reinterpret_cast<const uint4*>(rec + ((uint64_t)item * cells + cell) * 32);
  uint4* dst = reinterpret_cast<uint4*>(soa + ((uint64_t)row_of_cell[cell] * count + item) * 32);
  const uint4 a = src[0], b = src[1];
  dst[0] = a;
  dst[1] = b;
}

// ---------------------------------------------------------------------------------------------
// host-callable launch wrappers (engine.cpp is plain C++ and never sees a kernel symbol)
// ---------------------------------------------------------------------------------------------
#include "kernels.h"
// Block size of a plan launch: AFX_BLOCK, but one or two waves when no pass of the launch has more items than that - a launch merged
// from many small passes (16 items each) then holds one live wave per row instead of a block of four of which three retire at once,
// and a compute unit keeps twelve such rows in flight instead of three.
static inline uint32_t block_for(uint32_t max_count) { return max_count <= 64 ? 64u : max_count <= 128 ? 128u : (uint32_t)AFX_BLOCK; }
static inline dim3 grid_for(uint32_t count, uint32_t njobs) { const uint32_t b = block_for(count); return dim3((count + b - 1) / b, njobs, 1); }

hipError_t afxk_setup_generators(hipStream_t s, const uint8_t* enc, uint32_t ngen, int32_t* ext, uint8_t* neg_enc, uint32_t* ok) {
  hipLaunchKernelGGL(k_setup_generators, dim3((ngen + 63) / 64), dim3(64), 0, s, enc, ngen, ext, neg_enc, ok);
  return hipGetLastError();
}
// `max_count`: the largest item count among the passes of the launch (sizes the grid; a row's lanes past its own pass's count retire)
hipError_t afxk_decode(hipStream_t s, const afx_decode_job* jobs, uint32_t njobs, const afx_row* rows, const afx_pass* passes, uint32_t max_count, int mixed) {
  if (mixed) hipLaunchKernelGGL(k_decode_mixed, grid_for(max_count, njobs), dim3(block_for(max_count)), 0, s, jobs, rows, passes);
  else hipLaunchKernelGGL(k_decode, grid_for(max_count, njobs), dim3(block_for(max_count)), 0, s, jobs, rows, passes);
  return hipGetLastError();
}
hipError_t afxk_sccheck(hipStream_t s, const afx_sccheck_job* jobs, uint32_t njobs, const afx_row* rows, const afx_pass* passes, uint32_t max_count) {
  hipLaunchKernelGGL(k_sccheck, grid_for(max_count, njobs), dim3(block_for(max_count)), 0, s, jobs, rows, passes);
  return hipGetLastError();
}
hipError_t afxk_pointop(hipStream_t s, const afx_pointop_job* jobs, uint32_t njobs, const afx_row* rows, const afx_pass* passes, uint32_t max_count) {
  hipLaunchKernelGGL(k_pointop, grid_for(max_count, njobs), dim3(block_for(max_count)), 0, s, jobs, rows, passes);
  return hipGetLastError();
}
hipError_t afxk_scalarop(hipStream_t s, const afx_scalarop_job* jobs, uint32_t njobs, const afx_row* rows, const afx_pass* passes, uint32_t max_count) {
  hipLaunchKernelGGL(k_scalarop, grid_for(max_count, njobs), dim3(block_for(max_count)), 0, s, jobs, rows, passes);
  return hipGetLastError();
}
// base: scratch for ngen * windows extended points (AFX_VAR_DWORDS each); secret != 0: the 6-bit tables for secret scalars (AFX_SEC_*: a window's 32 multiples, fetched one per lane and exchanged)
hipError_t afxk_setup_postables(hipStream_t s, const int32_t* ext, uint32_t ngen, int32_t* base, int32_t* postab, int secret) {
  const uint32_t bits = secret ? AFX_SEC_BITS : AFX_POS_BITS, windows = secret ? AFX_SEC_WINDOWS : AFX_POS_WINDOWS;
  const uint32_t entries = secret ? AFX_SEC_ENTRIES : AFX_POS_ENTRIES, wd = secret ? AFX_SEC_WINDOW_DWORDS : AFX_POS_WINDOW_DWORDS;
  const uint32_t chunks = (entries + AFX_POS_CHUNK - 1) / AFX_POS_CHUNK;
  hipLaunchKernelGGL(k_setup_posbase, dim3((ngen * windows + 63) / 64), dim3(64), 0, s, ext, ngen, base, bits, windows);
  hipLaunchKernelGGL(k_setup_postables, dim3((ngen * windows * chunks + 63) / 64), dim3(64), 0, s, base, ngen, postab, windows, entries, wd);
  return hipGetLastError();
}
template <int KIND>
static void launch_msm(hipStream_t s, int encodes, int secret, dim3 grid, dim3 block, const afx_msm_djob* jobs, const int32_t* pos_tables, const int32_t* sec_tables,
                       const afx_pass& P, unsigned long long* clock_probe) {
  if (secret) {
    if (encodes) hipLaunchKernelGGL((k_msm<KIND, true, true>), grid, block, 0, s, jobs, pos_tables, sec_tables, P.table_ws, P.digit_ws, P.bad, P.count, clock_probe);
    else hipLaunchKernelGGL((k_msm<KIND, false, true>), grid, block, 0, s, jobs, pos_tables, sec_tables, P.table_ws, P.digit_ws, P.bad, P.count, clock_probe);
  } else {
    if (encodes) hipLaunchKernelGGL((k_msm<KIND, true, false>), grid, block, 0, s, jobs, pos_tables, sec_tables, P.table_ws, P.digit_ws, P.bad, P.count, clock_probe);
    else hipLaunchKernelGGL((k_msm<KIND, false, false>), grid, block, 0, s, jobs, pos_tables, sec_tables, P.table_ws, P.digit_ws, P.bad, P.count, clock_probe);
  }
}
template <int KIND>
static void launch_msm_rows(hipStream_t s, int encodes, int secret, dim3 grid, dim3 block, const uint8_t* blob, const int32_t* pos_tables, const int32_t* sec_tables,
                            const afx_row* rows, const afx_pass* passes, unsigned long long* clock_probe) {
  if (secret) {
    if (encodes) hipLaunchKernelGGL((k_msm_rows<KIND, true, true>), grid, block, 0, s, blob, pos_tables, sec_tables, rows, passes, clock_probe);
    else hipLaunchKernelGGL((k_msm_rows<KIND, false, true>), grid, block, 0, s, blob, pos_tables, sec_tables, rows, passes, clock_probe);
  } else {
    if (encodes) hipLaunchKernelGGL((k_msm_rows<KIND, true, false>), grid, block, 0, s, blob, pos_tables, sec_tables, rows, passes, clock_probe);
    else hipLaunchKernelGGL((k_msm_rows<KIND, false, false>), grid, block, 0, s, blob, pos_tables, sec_tables, rows, passes, clock_probe);
  }
}
// rows == null: a plan's own launch; `pass_host` is the HOST copy of its pass (the fields travel as kernel arguments).  Otherwise a
// merged launch: `jobs` is the blob's base, rows / passes are device tables.
hipError_t afxk_msm(hipStream_t s, int kind, int encodes, int secret, const afx_msm_djob* jobs, uint32_t njobs, const int32_t* pos_tables,
                    const int32_t* sec_tables, const afx_row* rows, const afx_pass* passes, const afx_pass* pass_host, uint32_t max_count, unsigned long long* clock_probe, uint32_t variants) {
  if (secret && !sec_tables) return hipErrorInvalidValue;
  // secret & 2: some job's narrow tables hold cached entries (afx_msm_job.narrow == 2): only the four-wave chains read those, whatever
  // the launch's size (such jobs come from small prover passes; a merged launch wide enough to matter is a request of dozens of shapes)
  const bool only_quad = (secret & 2) != 0;
  secret = secret != 0;
  // a launch that leaves the device idle - windowed or fixed-base jobs without secret terms or in-kernel encodings, at most two
  // blocks of four waves per compute unit in all - runs four waves per item chain (k_msm_quad).  AFX_KV_ONE_WAVE_CHAINS switches it
  // off (tests: the two kernels give the same bytes).
  const bool quad_on = !(variants & AFX_KV_ONE_WAVE_CHAINS);
  const uint32_t quad_blocks = (max_count + 63) / 64;
  // (a plan on its own up to 1024 blocks: 512-item calls gain 7 %; merged launches up to 512: at 1024 a 16-shape request loses 14 %)
  if (only_quad && (encodes || kind == MSM_NAF || !max_count)) return hipErrorInvalidValue;
  if ((only_quad || quad_on) && !encodes && kind != MSM_NAF && max_count && (only_quad || (uint64_t)quad_blocks * njobs <= (rows ? 512u : 1024u))) {
    const dim3 qgrid(quad_blocks, njobs);
    if (!rows) {
      if (!pass_host) return hipErrorInvalidValue;
      if (secret) hipLaunchKernelGGL(k_msm_quad<true>, qgrid, dim3(256), 0, s, jobs, pos_tables, sec_tables, pass_host->table_ws, pass_host->digit_ws, pass_host->bad, pass_host->count);
      else hipLaunchKernelGGL(k_msm_quad<false>, qgrid, dim3(256), 0, s, jobs, pos_tables, sec_tables, pass_host->table_ws, pass_host->digit_ws, pass_host->bad, pass_host->count);
    } else {
      if (secret) hipLaunchKernelGGL(k_msm_quad_rows<true>, qgrid, dim3(256), 0, s, (const uint8_t*)jobs, pos_tables, sec_tables, rows, passes);
      else hipLaunchKernelGGL(k_msm_quad_rows<false>, qgrid, dim3(256), 0, s, (const uint8_t*)jobs, pos_tables, sec_tables, rows, passes);
    }
    return hipGetLastError();
  }
  const dim3 grid = grid_for(max_count, njobs), block(block_for(max_count));
  if (!rows) {
    if (!pass_host) return hipErrorInvalidValue;
    switch (kind) {
      case MSM_FIXED: launch_msm<MSM_FIXED>(s, encodes, secret, grid, block, jobs, pos_tables, sec_tables, *pass_host, clock_probe); break;
      case MSM_WINDOW: launch_msm<MSM_WINDOW>(s, encodes, secret, grid, block, jobs, pos_tables, sec_tables, *pass_host, clock_probe); break;
      case MSM_NAF: launch_msm<MSM_NAF>(s, encodes, secret, grid, block, jobs, pos_tables, sec_tables, *pass_host, clock_probe); break;
      default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
  }
  switch (kind) {
    case MSM_FIXED: launch_msm_rows<MSM_FIXED>(s, encodes, secret, grid, block, (const uint8_t*)jobs, pos_tables, sec_tables, rows, passes, clock_probe); break;
    case MSM_WINDOW: launch_msm_rows<MSM_WINDOW>(s, encodes, secret, grid, block, (const uint8_t*)jobs, pos_tables, sec_tables, rows, passes, clock_probe); break;
    default: return hipErrorInvalidValue;   // NAF schedules belong to the plans of large passes, which are never merged
  }
  return hipGetLastError();
}
hipError_t afxk_msm_tables(hipStream_t s, int kind, const afx_table_job* jobs, uint32_t nrows, const afx_row* rows, const afx_pass* passes, uint32_t max_count) {
  switch (kind) {
    case TABLE_WINDOW: hipLaunchKernelGGL(k_msm_tables<TABLE_WINDOW>, grid_for(max_count, nrows), dim3(block_for(max_count)), 0, s, jobs, rows, passes); break;
    case TABLE_ODD: hipLaunchKernelGGL(k_msm_tables<TABLE_ODD>, grid_for(max_count, nrows), dim3(block_for(max_count)), 0, s, jobs, rows, passes); break;
    case TABLE_NARROW: hipLaunchKernelGGL(k_msm_tables<TABLE_NARROW>, grid_for(max_count, nrows), dim3(block_for(max_count)), 0, s, jobs, rows, passes); break;
    case TABLE_NARROW_CACHED: hipLaunchKernelGGL(k_msm_tables<TABLE_NARROW_CACHED>, grid_for(max_count, nrows), dim3(block_for(max_count)), 0, s, jobs, rows, passes); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}
hipError_t afxk_compress2x(hipStream_t s, const afx_compress_job* jobs, const afx_walk_row* rows, uint32_t nrows, const afx_pass* passes, uint32_t max_count) {
  hipLaunchKernelGGL(k_compress2x, grid_for(max_count, nrows), dim3(block_for(max_count)), 0, s, jobs, rows, passes);
  return hipGetLastError();
}
hipError_t afxk_table_affine(hipStream_t s, const afx_table_job* jobs, const afx_walk_row* rows, uint32_t nrows, const afx_pass* passes, uint32_t max_count) {
  hipLaunchKernelGGL(k_table_affine, grid_for(max_count, nrows), dim3(block_for(max_count)), 0, s, jobs, rows, passes);
  return hipGetLastError();
}
hipError_t afxk_negenc(hipStream_t s, const afx_negenc_job* jobs, const afx_walk_row* rows, uint32_t nrows, const afx_pass* passes, uint32_t max_count) {
  hipLaunchKernelGGL(k_negenc, grid_for(max_count, nrows), dim3(block_for(max_count)), 0, s, jobs, rows, passes);
  return hipGetLastError();
}
hipError_t afxk_pointsum(hipStream_t s, const afx_pointsum_job* jobs, uint32_t njobs, const afx_row* rows, const afx_pass* passes, uint32_t max_count, int many_parts, uint32_t variants) {
  // the idle-device form, under the same rule as k_msm_quad (and the same switch)
  const bool quad_on = !(variants & AFX_KV_ONE_WAVE_CHAINS);
  // jobs of many parts over few items: a lane per part (AFX_KV_NO_POINTSUM_TREE switches it off: tests, same bytes)
  const bool tree_on = !(variants & AFX_KV_NO_POINTSUM_TREE);
  if (quad_on && tree_on && many_parts && max_count && (uint64_t)max_count * njobs <= 1024) {
    hipLaunchKernelGGL(k_pointsum_tree, dim3(max_count, njobs), dim3(256), 0, s, jobs, rows, passes);
    return hipGetLastError();
  }
  if (quad_on && max_count && (uint64_t)((max_count + 63) / 64) * njobs <= 512) {
    hipLaunchKernelGGL(k_pointsum_quad, dim3((max_count + 63) / 64, njobs), dim3(256), 0, s, jobs, rows, passes);
    return hipGetLastError();
  }
  hipLaunchKernelGGL(k_pointsum, grid_for(max_count, njobs), dim3(block_for(max_count)), 0, s, jobs, rows, passes);
  return hipGetLastError();
}
hipError_t afxk_powers(hipStream_t s, const afx_powers_job* jobs, uint32_t njobs, const afx_row* rows, const afx_pass* passes, uint32_t max_count) {
  hipLaunchKernelGGL(k_powers_quad, dim3((max_count + 63) / 64, njobs), dim3(256), 0, s, jobs, rows, passes);
  return hipGetLastError();
}
hipError_t afxk_hash_coop(hipStream_t s, const afx_hash_program* progs, uint32_t nprogs, const afx_row* rows, const afx_pass* passes, uint32_t max_count, uint32_t variants) {
  // a wave per (item, program) while that leaves the device mostly idle (AFX_KV_HASH_HALF_WAVE: tests, the 32-lane groups always)
  const bool wave_on = !(variants & AFX_KV_HASH_HALF_WAVE);
  if (wave_on && (uint64_t)max_count * nprogs <= 2048) {
    const uint32_t b64 = max_count <= 1 ? 64u : max_count <= 2 ? 128u : (uint32_t)AFX_BLOCK, per64 = b64 / 64;
    hipLaunchKernelGGL(k_hash_coop64, dim3((max_count + per64 - 1) / per64, nprogs), dim3(b64), 0, s, progs, rows, passes);
    return hipGetLastError();
  }
  const uint32_t block = max_count <= 2 ? 64u : max_count <= 4 ? 128u : (uint32_t)AFX_BLOCK, per_block = block / 32;   // 32 lanes per item
  hipLaunchKernelGGL(k_hash_coop, dim3((max_count + per_block - 1) / per_block, nprogs), dim3(block), 0, s, progs, rows, passes);
  return hipGetLastError();
}
hipError_t afxk_hash(hipStream_t s, const afx_hash_program* progs, uint32_t nprogs, const afx_row* rows, const afx_pass* passes, uint32_t max_count) {
  hipLaunchKernelGGL(k_hash, grid_for(max_count, nprogs), dim3(block_for(max_count)), 0, s, progs, rows, passes);
  return hipGetLastError();
}
hipError_t afxk_finish(hipStream_t s, const afx_finish_job* jobs, uint32_t njobs, const afx_row* rows, uint32_t max_count) {
  hipLaunchKernelGGL(k_finish, dim3((max_count + 255) / 256, njobs), dim3(256), 0, s, jobs, rows);
  return hipGetLastError();
}
hipError_t afxk_fill_u32(hipStream_t s, const afx_fill_job* jobs, uint32_t njobs, const afx_row* rows, uint32_t max_n) {
  hipLaunchKernelGGL(k_fill_u32, dim3((max_n + 255) / 256, njobs), dim3(256), 0, s, jobs, rows);
  return hipGetLastError();
}
hipError_t afxk_from_uniform_jobs(hipStream_t s, const afx_uniform_job* jobs, uint32_t njobs, const afx_row* rows, const afx_pass* passes, uint32_t max_count) {
  hipLaunchKernelGGL(k_from_uniform_jobs, grid_for(max_count, njobs), dim3(block_for(max_count)), 0, s, jobs, rows, passes);
  return hipGetLastError();
}
hipError_t afxk_reduce_wide_jobs(hipStream_t s, const afx_reduce_job* jobs, uint32_t njobs, const afx_row* rows, const afx_pass* passes, uint32_t max_count) {
  hipLaunchKernelGGL(k_reduce_wide_jobs, grid_for(max_count, njobs), dim3(block_for(max_count)), 0, s, jobs, rows, passes);
  return hipGetLastError();
}
hipError_t afxk_from_uniform(hipStream_t s, const uint8_t* wide, uint8_t* out_enc, int32_t* out_var, uint32_t count) {
  hipLaunchKernelGGL(k_from_uniform, dim3((count + AFX_BLOCK - 1) / AFX_BLOCK), dim3(AFX_BLOCK), 0, s, wide, out_enc, out_var, count);
  return hipGetLastError();
}
hipError_t afxk_reduce_wide(hipStream_t s, const uint8_t* wide, uint8_t* out, uint32_t count) {
  hipLaunchKernelGGL(k_reduce_wide, dim3((count + AFX_BLOCK - 1) / AFX_BLOCK), dim3(AFX_BLOCK), 0, s, wide, out, count);
  return hipGetLastError();
}
hipError_t afxk_validate(hipStream_t s, const uint8_t* enc, uint8_t* ok, uint8_t* reenc, uint32_t count) {
  hipLaunchKernelGGL(k_validate, dim3((count + AFX_BLOCK - 1) / AFX_BLOCK), dim3(AFX_BLOCK), 0, s, enc, ok, reenc, count);
  return hipGetLastError();
}
hipError_t afxk_aos_to_soa(hipStream_t s, const uint8_t* rec, uint8_t* soa, const uint32_t* row_of_cell, uint32_t cells, uint32_t count) {
  const uint64_t n = (uint64_t)cells * count;
  hipLaunchKernelGGL(k_aos_to_soa, dim3((uint32_t)((n + AFX_BLOCK - 1) / AFX_BLOCK)), dim3(AFX_BLOCK), 0, s, rec, soa, row_of_cell, cells, count);
  return hipGetLastError();
}
