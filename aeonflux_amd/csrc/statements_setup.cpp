// Cold-path utilities of the crate, batch form.  SHA-512 runs on the host (byte hashing);
// every field / group / scalar operation still runs in the HIP kernels.
//   SystemParameters::hash_and_pray   /root/reference/src/parameters.rs:196-326
//   Plaintext::from(&[u8; 30])        /root/reference/src/symmetric.rs:135-143, encode_to_group src/encoding.rs:56-70
//   Keypair::derive                   /root/reference/src/symmetric.rs:197-215
//   Keypair::encrypt / decrypt        /root/reference/src/symmetric.rs:252-261, 273-289 (decode_from_group src/encoding.rs:75-82)
#include "sha512_host.hpp"
#include "statements.hpp"

static afx_scalarop_job mk_sop(const uint8_t* a, uint32_t as_, const uint8_t* b, uint32_t bs, const uint8_t* c, uint32_t cs, uint8_t* out) {
  afx_scalarop_job o;
  memset(&o, 0, sizeof o);
  o.a = a; o.a_stride = as_; o.b = b; o.b_stride = bs; o.c = c; o.c_stride = cs; o.out = out;
  return o;
}
static afx_msm_job mk_msm(const std::vector<afx_msm_term>& terms, const int32_t* addend, int32_t* out_var, uint8_t* out_enc) {
  afx_msm_job j;
  memset(&j, 0, sizeof j);
  set_terms(j, terms);
  j.addend = addend; j.out_var = out_var; j.out_enc = out_enc;
  return j;
}
// zeroes a host buffer that held secret-derived bytes when it goes out of scope (whatever the exit path)
struct WipeOnExit {
  std::vector<uint8_t>& v;
  explicit WipeOnExit(std::vector<uint8_t>& b) : v(b) {}
  ~WipeOnExit() { volatile uint8_t* q = v.data(); for (size_t i = 0; i < v.size(); i++) q[i] = 0; }
};
static int sync_fetch(afx_ctx* c, void* dst, const uint8_t* src, size_t n) {
  AFX_HIP(hipMemcpyAsync(dst, src, n, hipMemcpyDeviceToHost, c->stream));
  AFX_HIP(hipStreamSynchronize(c->stream));
  return AFX_OK;
}

// a context without parameters: stream + staging only (enough for the validity kernel)
static int bare_ctx(afx_ctx** out, int device) {
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device < 0 || device >= ndev) {
    set_error("no usable HIP device (this engine has no CPU fallback)");
    return AFX_E_NO_DEVICE;
  }
  afx_ctx* c = new afx_ctx();
  c->device = device;
  if (hipSetDevice(device) != hipSuccess || hipStreamCreateWithFlags(&c->lane[0].stream, hipStreamNonBlocking) != hipSuccess) {
    afx_ctx_destroy(c);
    set_error("hipStreamCreate failed");
    return AFX_E_HIP;
  }
  c->stream = c->lane[0].stream;
  *out = c;
  return AFX_OK;
}

extern "C" int afx_system_parameters_generate(int device, uint32_t n, const uint8_t* rng_stream, size_t stream_len, uint8_t* params_out,
                                              size_t params_cap, size_t* consumed_out) try {
  if (!rng_stream || !params_out || !consumed_out) { set_error("null argument"); return AFX_E_BAD_ARGS; }
  if (n == 0 || n > AFX_MAX_ATTRIBUTES) { set_error("number of attributes out of range"); return AFX_E_BAD_ARGS; }
  const uint32_t g = n < 3 ? 3 : n, total = 4 + g + n + 4;
  const size_t need = 4 + 32 * (size_t)(1 + total);
  if (params_cap < need) { set_error("output buffer too small"); return AFX_E_BAD_ARGS; }
  const size_t nblocks = stream_len / 32;
  if (nblocks == 0) { set_error("empty rng stream"); return AFX_E_BAD_ARGS; }
  afx_ctx* c = nullptr;
  int rc = bare_ctx(&c, device);
  if (rc) return rc;
  // every 32-byte draw of the stream is tested on the GPU at once; the host then replays the reference's
  // sequential "draw until it decompresses" loops over the flags
  std::vector<uint8_t> ok(nblocks);
  {
    Stager st(c);
    const size_t o_in = st.add(rng_stream, 32 * nblocks), o_ok = st.add(nullptr, nblocks);
    if (!(rc = st.upload())) {
      hipError_t e = afxk_validate(c->stream, st.dev(o_in), st.dev(o_ok), nullptr, (uint32_t)nblocks);
      if (e != hipSuccess) { set_error(hipGetErrorString(e)); rc = AFX_E_HIP; }
      else rc = sync_fetch(c, ok.data(), st.dev(o_ok), nblocks);
    }
  }
  // the ristretto basepoint's encoding: G (parameters.rs:283, RISTRETTO_BASEPOINT_POINT)
  static const uint8_t BASE[32] = { 0xe2, 0xf2, 0xae, 0x0a, 0x6a, 0xbc, 0x4e, 0x71, 0xa8, 0x84, 0xa9, 0x61, 0xc5, 0x00, 0x51, 0x5f,
                                    0x58, 0xe3, 0x0b, 0x6a, 0xa5, 0x82, 0xdd, 0x8d, 0xb6, 0xa6, 0x59, 0x45, 0xe0, 0x8d, 0x2d, 0x76 };
  afx_ctx_destroy(c);
  if (rc) return rc;
  std::vector<const uint8_t*> gen(total);
  size_t pos = 0;
  for (uint32_t k = 0; k < total; k++) {
    for (;;) {
      if (pos >= nblocks) { set_error("rng stream exhausted before every generator was found"); return AFX_E_BAD_ARGS; }
      const bool good = ok[pos] != 0;
      gen[k] = rng_stream + 32 * pos;
      pos++;
      if (good) break;
    }
  }
  // order drawn: G_w, G_w', G_x0, G_x1, G_y[g], G_m[n], G_V, G_a, G_a0, G_a1.  Uniqueness / non-identity check restated
  // literally (parameters.rs:297-323: only the first n G_y are listed, and the inner loop skips the last element)
  std::vector<const uint8_t*> list;
  static const uint8_t ZERO[32] = { 0 };
  list.push_back(ZERO); list.push_back(BASE);
  list.push_back(gen[0]); list.push_back(gen[1]); list.push_back(gen[2]); list.push_back(gen[3]);
  for (uint32_t k = 0; k < 4; k++) list.push_back(gen[4 + g + n + k]);
  for (uint32_t i = 0; i < n; i++) { list.push_back(gen[4 + i]); list.push_back(gen[4 + g + i]); }
  while (list.size() >= 2) {
    const uint8_t* x = list.back();
    list.pop_back();
    for (size_t i = 0; i + 1 < list.size(); i++)
      if (memcmp(x, list[i], 32) == 0) { set_error("generators are not unique (CredentialError::NoSystemParameters)"); return AFX_E_BAD_PARAMS; }
  }
  uint8_t* p = params_out;
  p[0] = (uint8_t)n; p[1] = (uint8_t)(n >> 8); p[2] = (uint8_t)(n >> 16); p[3] = (uint8_t)(n >> 24);
  p += 4;
  memcpy(p, BASE, 32); p += 32;
  for (uint32_t k = 0; k < total; k++) { memcpy(p, gen[k], 32); p += 32; }
  *consumed_out = 32 * pos;
  return AFX_OK;
} catch (...) { return afx::exception_rc(); }

extern "C" int afx_plaintexts_from_bytes(afx_ctx* ctx, const uint8_t* msgs, size_t count, uint8_t* M1, uint8_t* M2, uint8_t* m3, uint32_t* counters) try {
  CtxLock lock__(ctx);
  if (!ctx || !msgs || !M1 || !M2 || !m3) { set_error("null argument"); return AFX_E_BAD_ARGS; }
  if (count == 0) return AFX_OK;
  AFX_HIP(hipSetDevice(ctx->device));
  int rc;
  // M2 = HashToG(m), m3 = HashToZZq(m): SHA-512 on the host, Elligator / reduction on the GPU
  {
    std::vector<uint8_t> wide(64 * count);
    for (size_t i = 0; i < count; i++) sha512(wide.data() + 64 * i, msgs + 30 * i, 30);
    Stager st(ctx);
    const size_t o_w = st.add(wide.data(), wide.size()), o_M2 = st.add(nullptr, 32 * count), o_m3 = st.add(nullptr, 32 * count);
    if ((rc = st.upload())) return rc;
    AFX_HIP(afxk_from_uniform(ctx->stream, st.dev(o_w), st.dev(o_M2), nullptr, (uint32_t)count));
    AFX_HIP(afxk_reduce_wide(ctx->stream, st.dev(o_w), st.dev(o_m3), (uint32_t)count));
    AFX_HIP(hipMemcpyAsync(M2, st.dev(o_M2), 32 * count, hipMemcpyDeviceToHost, ctx->stream));
    if ((rc = sync_fetch(ctx, m3, st.dev(o_m3), 32 * count))) return rc;
  }
  // M1 = EncodeToG(m): candidates i || m || j, counter order i fastest (encoding.rs:56-70); 16 candidates per message per round
  std::vector<size_t> pending(count);
  for (size_t i = 0; i < count; i++) pending[i] = i;
  const uint32_t PER = 16;
  for (uint32_t base = 0; base < 128 * 64 && !pending.empty(); base += PER) {
    std::vector<uint8_t> cand(32 * PER * pending.size());
    for (size_t k = 0; k < pending.size(); k++)
      for (uint32_t t = 0; t < PER; t++) {
        uint8_t* b = cand.data() + 32 * (k * PER + t);
        const uint32_t ctr = base + t;
        b[0] = (uint8_t)(2 * (ctr % 128));
        memcpy(b + 1, msgs + 30 * pending[k], 30);
        b[31] = (uint8_t)(ctr / 128);
      }
    std::vector<uint8_t> ok(PER * pending.size());
    Stager st(ctx);
    const size_t o_in = st.add(cand.data(), cand.size()), o_ok = st.add(nullptr, ok.size());
    if ((rc = st.upload())) return rc;
    AFX_HIP(afxk_validate(ctx->stream, st.dev(o_in), st.dev(o_ok), nullptr, (uint32_t)ok.size()));
    if ((rc = sync_fetch(ctx, ok.data(), st.dev(o_ok), ok.size()))) return rc;
    std::vector<size_t> still;
    for (size_t k = 0; k < pending.size(); k++) {
      uint32_t t = 0;
      while (t < PER && !ok[k * PER + t]) t++;
      if (t == PER) { still.push_back(pending[k]); continue; }
      memcpy(M1 + 32 * pending[k], cand.data() + 32 * (k * PER + t), 32);
      if (counters) counters[pending[k]] = base + t;
    }
    pending.swap(still);
  }
  if (!pending.empty()) { set_error("encode_to_group found no representative (the reference panics)"); return AFX_E_BAD_ARGS; }
  return AFX_OK;
} catch (...) { return afx::exception_rc(); }

extern "C" int afx_keypairs_derive(afx_ctx* ctx, const uint8_t* master_secrets, size_t count, uint8_t* a, uint8_t* a0, uint8_t* a1, uint8_t* pk) try {
  CtxLock lock__(ctx);
  if (!ctx || !master_secrets || !a || !a0 || !a1 || !pk) { set_error("null argument"); return AFX_E_BAD_ARGS; }
  if (count == 0) return AFX_OK;
  AFX_HIP(hipSetDevice(ctx->device));
  int rc;
  // a = H(master), a0 = H(a), a1 = H(a0)  (symmetric.rs:202-204)
  std::vector<uint8_t> wide(64 * count);   // SHA-512 outputs that reduce to the secret keys
  WipeOnExit wipe_wide(wide);
  uint8_t* outs[3] = { a, a0, a1 };
  for (int round = 0; round < 3; round++) {
    for (size_t i = 0; i < count; i++) {
      if (round == 0) sha512(wide.data() + 64 * i, master_secrets + 64 * i, 64);
      else sha512(wide.data() + 64 * i, outs[round - 1] + 32 * i, 32);
    }
    Stager st(ctx);
    const size_t o_w = st.add(wide.data(), wide.size()), o_s = st.add(nullptr, 32 * count);
    if ((rc = st.upload())) return rc;
    AFX_HIP(afxk_reduce_wide(ctx->stream, st.dev(o_w), st.dev(o_s), (uint32_t)count));
    if ((rc = sync_fetch(ctx, outs[round], st.dev(o_s), 32 * count))) return rc;
  }
  // pk = G_a*a + G_a0*a0 + G_a1*a1  (symmetric.rs:206-209)
  Stager st(ctx);
  const size_t o_a = st.add(a, 32 * count), o_a0 = st.add(a0, 32 * count), o_a1 = st.add(a1, 32 * count), o_pk = st.add(nullptr, 32 * count), o_st = st.add(nullptr, count);
  if ((rc = st.upload())) return rc;
  rc = run_chunked(ctx, count, [&](Assembler& as, size_t off, uint32_t) {
    as.secret_scalars = true;   // the user's symmetric key: afx_ctx_set_secret_independent_addressing
    afx_ctx* c = as.ctx;
    as.msm({ mk_msm({ mk_term(st.dev(o_a) + 32 * off, 32, nullptr, (int32_t)c->id_Ga(), false), mk_term(st.dev(o_a0) + 32 * off, 32, nullptr, (int32_t)c->id_Ga0(), false),
                      mk_term(st.dev(o_a1) + 32 * off, 32, nullptr, (int32_t)c->id_Ga1(), false) }, nullptr, nullptr, st.dev(o_pk) + 32 * off) });
    as.finish(st.dev(o_st) + off, 1);
  });
  if (rc) return rc;
  return sync_fetch(ctx, pk, st.dev(o_pk), 32 * count);
} catch (...) { return afx::exception_rc(); }

extern "C" int afx_encrypt(afx_ctx* ctx, const afx_keypairs_soa* kp, const uint8_t* M1, const uint8_t* M2, const uint8_t* m3, size_t count,
                           uint8_t* E1, uint8_t* E2, uint8_t* status) try {
  CtxLock lock__(ctx);
  if (!ctx || !kp || !kp->a || !kp->a0 || !kp->a1 || !M1 || !M2 || !m3 || !E1 || !E2 || !status) { set_error("null argument"); return AFX_E_BAD_ARGS; }
  if (count == 0) return AFX_OK;
  AFX_HIP(hipSetDevice(ctx->device));
  Stager st(ctx);
  const size_t row = 32 * count;
  const size_t o_a = st.add(kp->a, row), o_a0 = st.add(kp->a0, row), o_a1 = st.add(kp->a1, row), o_M1 = st.add(M1, row), o_M2 = st.add(M2, row),
               o_m3 = st.add(m3, row), o_E1 = st.add(nullptr, row), o_E2 = st.add(nullptr, row), o_st = st.add(nullptr, count);
  int rc = st.upload();
  if (rc) return rc;
  rc = run_chunked(ctx, count, [&](Assembler& as, size_t off, uint32_t) {
    as.secret_scalars = true;   // the user's symmetric key: afx_ctx_set_secret_independent_addressing
    auto at = [&](size_t o) { return st.dev(o) + 32 * off; };
    int32_t *v_M1 = as.new_var(), *v_M2 = as.new_var(), *v_E1 = as.new_var();
    uint8_t* k = as.new_enc();
    as.sccheck({ { at(o_a) }, { at(o_a0) }, { at(o_a1) }, { at(o_m3) } });
    as.decode({ { at(o_M1), v_M1, 0 }, { at(o_M2), v_M2, 0 } });
    as.scalarop({ mk_sop(at(o_a1), 32, at(o_m3), 32, at(o_a0), 32, k) });                                   // a0 + a1*m3
    afx_msm_job jE1 = mk_msm({ mk_term(k, 32, v_M2, -1, false) }, nullptr, v_E1, at(o_E1));                   // E1 = M2*(a0 + a1*m3)
    jE1.leave_half = 1;   // an output and the base of E2's term: plan.h afx_msm_job.leave_half
    as.msm({ jE1 });
    as.msm({ mk_msm({ mk_term(at(o_a), 32, v_E1, -1, false) }, v_M1, nullptr, at(o_E2)) });                   // E2 = E1*a + M1
    as.finish(st.dev(o_st) + off, AFX_ST_VERIFICATION_FAILURE);
  });
  if (rc) return rc;
  AFX_HIP(hipMemcpyAsync(E1, st.dev(o_E1), row, hipMemcpyDeviceToHost, ctx->stream));
  AFX_HIP(hipMemcpyAsync(E2, st.dev(o_E2), row, hipMemcpyDeviceToHost, ctx->stream));
  return sync_fetch(ctx, status, st.dev(o_st), count);
} catch (...) { return afx::exception_rc(); }

extern "C" int afx_decrypt(afx_ctx* ctx, const afx_keypairs_soa* kp, const uint8_t* E1, const uint8_t* E2, size_t count, uint8_t* M1, uint8_t* M2,
                           uint8_t* m3, uint8_t* messages, uint8_t* status) try {
  CtxLock lock__(ctx);
  if (!ctx || !kp || !kp->a || !kp->a0 || !kp->a1 || !E1 || !E2 || !M1 || !M2 || !m3 || !status) { set_error("null argument"); return AFX_E_BAD_ARGS; }
  if (count == 0) return AFX_OK;
  AFX_HIP(hipSetDevice(ctx->device));
  const size_t row = 32 * count;
  int rc;
  std::vector<uint8_t> bad1(count), bad2(count), e1p(row), wide(64 * count);
  WipeOnExit wipe_wide(wide), wipe_e1p(e1p);   // hashes of the recovered plaintext, candidate E1
  // M1' = E2 - E1*a  (symmetric.rs:278)
  {
    Stager st(ctx);
    const size_t o_a = st.add(kp->a, row), o_E1 = st.add(E1, row), o_E2 = st.add(E2, row), o_M1 = st.add(nullptr, row), o_st = st.add(nullptr, count);
    if ((rc = st.upload())) return rc;
    rc = run_chunked(ctx, count, [&](Assembler& as, size_t off, uint32_t) {
    as.secret_scalars = true;   // the user's symmetric key: afx_ctx_set_secret_independent_addressing
      auto at = [&](size_t o) { return st.dev(o) + 32 * off; };
      int32_t *v_E1 = as.new_var(), *v_E2 = as.new_var();
      as.sccheck({ { at(o_a) } });
      as.decode({ { at(o_E1), v_E1, 0 }, { at(o_E2), v_E2, 0 } });
      afx_msm_job j = mk_msm({ mk_term(at(o_a), 32, v_E1, -1, true) }, v_E2, nullptr, at(o_M1));
      as.msm({ j });
      as.finish(st.dev(o_st) + off, 1);
    });
    if (rc) return rc;
    AFX_HIP(hipMemcpyAsync(M1, st.dev(o_M1), row, hipMemcpyDeviceToHost, ctx->stream));
    if ((rc = sync_fetch(ctx, bad1.data(), st.dev(o_st), count))) return rc;
  }
  // m' = decode_from_group(M1'); m3' = HashToZZq(m'), M2' = HashToG(m'); E1' = M2'*(a0 + a1*m3')  (symmetric.rs:279-283)
  for (size_t i = 0; i < count; i++) {
    const uint8_t* m = M1 + 32 * i + 1;
    if (messages) memcpy(messages + 30 * i, m, 30);
    sha512(wide.data() + 64 * i, m, 30);
  }
  {
    Stager st(ctx);
    const size_t o_w = st.add(wide.data(), wide.size()), o_a0 = st.add(kp->a0, row), o_a1 = st.add(kp->a1, row), o_M2 = st.add(nullptr, row),
                 o_m3 = st.add(nullptr, row), o_E1p = st.add(nullptr, row), o_st = st.add(nullptr, count);
    if ((rc = st.upload())) return rc;
    rc = run_chunked(ctx, count, [&](Assembler& as, size_t off, uint32_t) {
    as.secret_scalars = true;   // the user's symmetric key: afx_ctx_set_secret_independent_addressing
      auto at = [&](size_t o) { return st.dev(o) + 32 * off; };
      int32_t* v_M2 = as.new_var();
      uint8_t* k = as.new_enc();
      as.sccheck({ { at(o_a0) }, { at(o_a1) } });
      as.reduce_wide(st.dev(o_w) + 64 * off, at(o_m3));
      as.from_uniform(st.dev(o_w) + 64 * off, at(o_M2), v_M2);
      as.scalarop({ mk_sop(at(o_a1), 32, at(o_m3), 32, at(o_a0), 32, k) });
      as.msm({ mk_msm({ mk_term(k, 32, v_M2, -1, false) }, nullptr, nullptr, at(o_E1p)) });
      as.finish(st.dev(o_st) + off, 1);
    });
    if (rc) return rc;
    AFX_HIP(hipMemcpyAsync(M2, st.dev(o_M2), row, hipMemcpyDeviceToHost, ctx->stream));
    AFX_HIP(hipMemcpyAsync(m3, st.dev(o_m3), row, hipMemcpyDeviceToHost, ctx->stream));
    AFX_HIP(hipMemcpyAsync(e1p.data(), st.dev(o_E1p), row, hipMemcpyDeviceToHost, ctx->stream));
    if ((rc = sync_fetch(ctx, bad2.data(), st.dev(o_st), count))) return rc;
  }
  // ciphertext.E1 == E1' : canonical encodings are equal iff the group elements are (symmetric.rs:285-288)
  for (size_t i = 0; i < count; i++)
    status[i] = (!bad1[i] && !bad2[i] && memcmp(e1p.data() + 32 * i, E1 + 32 * i, 32) == 0) ? AFX_ST_OK : AFX_ST_UNDECRYPTABLE;
  return AFX_OK;
} catch (...) { return afx::exception_rc(); }
