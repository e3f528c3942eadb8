// Issuance (Issuer::issue, CredentialIssuance::verify) and presentation (AnonymousCredential::show) in batch
// form.  Restates, as GPU launch lists:
//   /root/reference/src/issuer.rs:111-124  Issuer::issue  = Amac::tag (src/amacs.rs:276-294, compute_V :256-272,
//                                           Messages::from_attributes :225-243) + ProofOfIssuance::prove
//                                           (src/nizk/issuance.rs:40-129)
//   /root/reference/src/issuer.rs:48-57    CredentialIssuance::verify -> src/nizk/issuance.rs:132-218
//   /root/reference/src/credential.rs:37-46 show -> src/nizk/presentation.rs:139-321 + src/nizk/encryption.rs:58-142
//                                           + Keypair::encrypt (src/symmetric.rs:252-261)
//   /root/reference/src/parameters.rs:349-362, src/amacs.rs:104  IssuerParameters::generate, W = w*G_w
#include "statements.hpp"

static bool is_scalar_kind(uint8_t k) { return k == AFX_ATTR_PUBLIC_SCALAR || k == AFX_ATTR_SECRET_SCALAR; }

static afx_scalarop_job mk_scalarop(const uint8_t* a, uint32_t as_, const uint8_t* b, uint32_t bs, const uint8_t* c, uint32_t cs, bool neg, uint8_t* out) {
  afx_scalarop_job o;
  memset(&o, 0, sizeof o);
  o.a = a; o.a_stride = as_; o.b = b; o.b_stride = bs; o.c = c; o.c_stride = cs; o.negate = neg ? 1u : 0u; o.out = out;
  return o;
}
static afx_msm_job mk_job(const std::vector<afx_msm_term>& terms, const int32_t* addend, int32_t* out_var, uint8_t* out_enc, bool reject_identity) {
  afx_msm_job j;
  memset(&j, 0, sizeof j);
  set_terms(j, terms);
  j.addend = addend;
  j.out_var = out_var;
  j.out_enc = out_enc;
  j.reject_identity = reject_identity ? 1u : 0u;
  return j;
}
// a job whose result is encoded AND is the base of later multiscalar terms, and nothing else: it leaves its half (plan.h
// afx_msm_job.leave_half): no inverse square root for the encoding, the later terms double their scalars
static afx_msm_job leaving_half(afx_msm_job j) { j.leave_half = 1; return j; }
static ScalarVar sv_item(const uint8_t* dev) { ScalarVar s; s.dev = dev; s.stride = 32; return s; }
static ScalarVar sv_uniform(const uint8_t* dev, const Enc& host) { ScalarVar s; s.dev = dev; s.stride = 0; s.host = host; return s; }

// the transcript / constraint part shared by ProofOfIssuance::prove (:48-126) and ::verify (:142-215)
struct IssuanceVars {
  ScalarVar w, wp, x0, x1, y[AFX_MAX_ATTRIBUTES], one;
  PointVar U, V, tU, M[AFX_MAX_ATTRIBUTES];
  uint32_t n_messages;
};
static void issuance_statement(SchnorrBuilder& b, afx_ctx* c, const IssuanceVars& iv) {
  const uint32_t n = c->n, g = c->g;
  const int w = b.allocate_scalar("w", iv.w);
  const int w_prime = b.allocate_scalar("w'", iv.wp);
  const int x_0 = b.allocate_scalar("x_0", iv.x0);
  const int x_1 = b.allocate_scalar("x_1", iv.x1);
  int y[AFX_MAX_ATTRIBUTES];
  for (uint32_t i = 0; i < n; i++) y[i] = b.allocate_scalar("y", iv.y[i]);
  const int one = b.allocate_scalar("1", iv.one);
  const int G_V = b.allocate_point("G_V", PointVar::Const(c->id_GV()));
  const int G_w = b.allocate_point("G_w", PointVar::Const(c->id_Gw()));
  const int G_w_prime = b.allocate_point("G_w_prime", PointVar::Const(c->id_Gwp()));
  const int neg_G_x_0 = b.allocate_point("-G_x_0", PointVar::Const(c->id_Gx0(), true));
  const int neg_G_x_1 = b.allocate_point("-G_x_1", PointVar::Const(c->id_Gx1(), true));
  int neg_G_y[AFX_MAX_ATTRIBUTES];
  for (uint32_t i = 0; i < g; i++) neg_G_y[i] = b.allocate_point("-G_y", PointVar::Const(c->id_Gy(i), true));
  const int C_W = b.allocate_point("C_W", PointVar::Const(c->id_CW()));
  const int I = b.allocate_point("I", PointVar::Const(c->id_I()));
  const int U = b.allocate_point("U", iv.U);
  const int V = b.allocate_point("V", iv.V);
  const int tU = b.allocate_point("tU", iv.tU);
  int M[AFX_MAX_ATTRIBUTES];
  for (uint32_t i = 0; i < iv.n_messages; i++) M[i] = b.allocate_point("M", iv.M[i]);
  b.constrain(C_W, { { w, G_w }, { w_prime, G_w_prime } });
  std::vector<std::pair<int, int>> rhs = { { one, G_V }, { x_0, neg_G_x_0 }, { x_1, neg_G_x_1 } };
  for (uint32_t i = 0; i < n; i++) rhs.push_back({ y[i], neg_G_y[i] });   // y.zip(neg_G_y) stops at n (:114)
  b.constrain(I, rhs);
  rhs = { { w, G_w }, { x_0, U }, { x_1, tU } };
  for (uint32_t i = 0; i < n && i < iv.n_messages; i++) rhs.push_back({ y[i], M[i] });
  b.constrain(V, rhs);
}

// Messages::from_attributes (src/amacs.rs:225-243): scalar kinds -> m*G_m[i] (one fixed-base job each),
// point kinds -> the decoded point.  Returns per-attribute PointVars.
static void messages_from_attributes(Assembler& as, const afx_attributes_soa& a, size_t total, size_t off, bool reject_identity,
                                     std::vector<afx_sccheck_job>& sccheck, std::vector<afx_decode_job>& decode,
                                     std::vector<afx_msm_job>& msm, PointVar M[AFX_MAX_ATTRIBUTES]) {
  afx_ctx* c = as.ctx;
  for (uint32_t i = 0; i < a.n_attributes; i++) {
    const uint8_t* val = a.values + (i * total + off) * 32;
    if (is_scalar_kind(a.kinds[i])) {
      // only the ENCODING of M_i = m_i*G_m_i is needed (it is hashed): every term on M_i runs as a fixed-base term on G_m_i
      // (has_alt), so the job keeps no coordinates and its encoding comes from k_compress2x
      sccheck.push_back({ val });
      uint8_t* e = as.new_enc();
      msm.push_back(mk_job({ mk_term(val, 32, nullptr, (int32_t)c->id_Gm(i), false) }, nullptr, nullptr, e, reject_identity));
      M[i] = PointVar::Var(nullptr, e);
      M[i].has_alt = true; M[i].alt_gen = c->id_Gm(i); M[i].alt_scalar = val;   // M_i = m_i * G_m_i
    } else {
      int32_t* v = as.new_var();
      decode.push_back({ val, v, reject_identity ? 1u : 0u });
      M[i] = PointVar::Var(v, val);
    }
  }
}

// ------------------------------------------------------------------------------------------------
// CredentialIssuance::verify
// ------------------------------------------------------------------------------------------------
extern "C" int afx_verify_issuances_dev(afx_ctx* ctx, const afx_attributes_soa* attrs, const afx_issuance_soa* iss, uint32_t n_responses,
                                        size_t count, uint8_t* status_dev) try {
  CtxLock lock__(ctx);
  if (!ctx || !attrs || !iss || !status_dev) { set_error("null argument"); return AFX_E_BAD_ARGS; }
  if (count == 0) return AFX_OK;
  const afx_attributes_soa a = *attrs;
  const afx_issuance_soa s = *iss;
  // the arrays a well-formed request reads (a request every item fails on reads none): a null one is a bad call, not a GPU fault
  if (a.n_attributes <= ctx->n && n_responses == ctx->n + 5 &&
      (!s.t || !s.U || !s.V || !s.challenge || !s.responses || (a.n_attributes && !a.values))) { set_error("null batch array"); return AFX_E_BAD_ARGS; }
  struct { uint32_t n, nr; uint8_t kinds[AFX_MAX_ATTRIBUTES]; } kd__;
  memset(&kd__, 0, sizeof kd__);
  kd__.n = a.n_attributes; kd__.nr = n_responses; memcpy(kd__.kinds, a.kinds, std::min<size_t>(a.n_attributes, AFX_MAX_ATTRIBUTES));
  const PlanKey key__ = ctx->trace ? PlanKey() : plan_key("verify_issuances", &kd__, sizeof kd__, mode_flags(ctx));
  return run_chunked(ctx, count, [&](Assembler& as, size_t off, uint32_t) {
    afx_ctx* c = as.ctx;
    JobSets js;
    auto row = [&](const uint8_t* base, size_t k) { return base + (k * count + off) * 32; };
    bool bad_kind = false;
    for (uint32_t i = 0; i < a.n_attributes && i < AFX_MAX_ATTRIBUTES; i++) bad_kind |= a.kinds[i] > AFX_ATTR_SECRET_POINT;
    // more attributes than generators: Messages::from_attributes indexes G_m[i] (panic); wrong response count: zkp rejects
    if (a.n_attributes > c->n || n_responses != c->n + 5 || bad_kind) as.fail_all = true;
    if (as.fail_all) { emit(as, js, status_dev + off, AFX_ST_VERIFICATION_FAILURE); return; }
    js.sccheck.push_back({ row(s.t, 0) });
    js.sccheck.push_back({ row(s.challenge, 0) });
    for (uint32_t r = 0; r < n_responses; r++) js.sccheck.push_back({ row(s.responses, r) });
    IssuanceVars iv;
    int32_t *v_U = as.new_var(), *v_V = as.new_var(), *v_tU = as.new_var();
    uint8_t* e_tU = as.new_enc();
    js.decode.push_back({ row(s.U, 0), v_U, 1 });
    js.decode.push_back({ row(s.V, 0), v_V, 1 });
    messages_from_attributes(as, a, count, off, true, js.sccheck, js.decode, js.msm1, iv.M);
    js.msm1.push_back(leaving_half(mk_job({ mk_term(row(s.t, 0), 32, v_U, -1, false) }, nullptr, v_tU, e_tU, true)));   // t*U (:189): hashed, and x_1's base
    iv.n_messages = a.n_attributes;
    iv.w = sv_item(row(s.responses, 0)); iv.wp = sv_item(row(s.responses, 1)); iv.x0 = sv_item(row(s.responses, 2)); iv.x1 = sv_item(row(s.responses, 3));
    for (uint32_t i = 0; i < c->n; i++) iv.y[i] = sv_item(row(s.responses, 4 + i));
    iv.one = sv_item(row(s.responses, 4 + c->n));
    iv.U = PointVar::Var(v_U, row(s.U, 0));
    iv.V = PointVar::Var(v_V, row(s.V, 0));
    iv.tU = PointVar::Var(v_tU, e_tU);
    SchnorrBuilder v(as, "2019/1416 anonymous credential", "2019/1416 issuance proof");
    issuance_statement(v, c, iv);
    v.verify_compact(row(s.challenge, 0), 0, count, off, js.msm2, js.hash, &js.scalarop);   // (resp_y * m_i) * G_m_i products: inputs only
    emit(as, js, status_dev + off, AFX_ST_VERIFICATION_FAILURE);
  }, key__);
} catch (...) { return afx::exception_rc(); }

// ------------------------------------------------------------------------------------------------
// Issuer::issue
// ------------------------------------------------------------------------------------------------
extern "C" int afx_issue_dev(afx_ctx* ctx, const afx_attributes_soa* requests, const afx_issue_randomness* rnd, size_t count,
                             const afx_issuance_soa* out, uint8_t* status_dev) try {
  CtxLock lock__(ctx);
  if (!ctx || !requests || !rnd || !out || !status_dev) { set_error("null argument"); return AFX_E_BAD_ARGS; }
  if (!ctx->has_key) { set_error("Issuer::issue needs the issuer key"); return AFX_E_NO_KEY; }
  if (count == 0) return AFX_OK;
  const afx_attributes_soa a = *requests;
  const afx_issue_randomness r = *rnd;
  const afx_issuance_soa o = *out;
  if (a.n_attributes == ctx->n && (!a.values || !r.t_wide || !r.U_wide || !r.rng_seed || !o.t || !o.U || !o.V || !o.challenge || !o.responses)) {
    set_error("null batch array");
    return AFX_E_BAD_ARGS;
  }
  struct { uint32_t n; uint8_t kinds[AFX_MAX_ATTRIBUTES]; } kd__;
  memset(&kd__, 0, sizeof kd__);
  kd__.n = a.n_attributes; memcpy(kd__.kinds, a.kinds, std::min<size_t>(a.n_attributes, AFX_MAX_ATTRIBUTES));
  const PlanKey key__ = plan_key("issue", &kd__, sizeof kd__, mode_flags(ctx));
  return run_chunked(ctx, count, [&](Assembler& as, size_t off, uint32_t) {
    afx_ctx* c = as.ctx;
    as.secret_scalars = true;   // the key, t, the proof's blindings: afx_ctx_set_secret_independent_addressing
    const uint32_t n = c->n;
    auto row = [&](const uint8_t* base, size_t k) { return base + (k * count + off) * 32; };
    auto orow = [&](uint8_t* base, size_t k) { return base + (k * count + off) * 32; };
    bool bad_kind = false;
    for (uint32_t i = 0; i < a.n_attributes && i < AFX_MAX_ATTRIBUTES; i++) bad_kind |= a.kinds[i] > AFX_ATTR_SECRET_POINT;
    // Amac::tag: attributes.len() != NUMBER_OF_ATTRIBUTES -> MacError::MessageLengthError -> MacCreation (amacs.rs:285-287)
    if (a.n_attributes != n || bad_kind) { as.fail_all = true; as.finish(status_dev + off, AFX_ST_MAC_CREATION); return; }
    std::vector<afx_sccheck_job> sccheck;
    std::vector<afx_decode_job> decode;
    std::vector<afx_msm_job> msm1;
    std::vector<afx_scalarop_job> sc1;
    // t = Scalar::random, U = RistrettoPoint::random (amacs.rs:289-290)
    int32_t* v_U = as.new_var();
    as.reduce_wide(r.t_wide + off * 64, orow(o.t, 0));
    as.from_uniform(r.U_wide + off * 64, orow(o.U, 0), v_U);
    IssuanceVars iv;
    messages_from_attributes(as, a, count, off, false, sccheck, decode, msm1, iv.M);
    // V = W + (x0 + x1*t)*U + sum y_i*M_i (amacs.rs:267-270); scalar attributes fold into fixed-base terms (y_i*m_i)*G_m[i]
    uint8_t* k_xt = as.new_enc();
    sc1.push_back(mk_scalarop(c->key_x1(), 0, orow(o.t, 0), 32, c->key_x0(), 0, false, k_xt));
    std::vector<afx_msm_term> vterms = { mk_term(k_xt, 32, v_U, -1, false) };
    for (uint32_t i = 0; i < n; i++) {
      if (is_scalar_kind(a.kinds[i])) {
        uint8_t* ym = as.new_enc();
        sc1.push_back(mk_scalarop(c->key_y(i), 0, row(a.values, i), 32, nullptr, 0, false, ym));
        vterms.push_back(mk_term(ym, 32, nullptr, (int32_t)c->id_Gm(i), false));
      } else {
        vterms.push_back(mk_term(c->key_y(i), 0, iv.M[i].var, -1, false));
      }
    }
    // W joins as the fixed-base term 1*W: V is then a pure multiscalar sum of which only the encoding is needed (the prover
    // never multiplies V), so it is encoded by k_compress2x like the commitments
    vterms.push_back(mk_term(c->const_one(), 0, nullptr, (int32_t)c->id_W(), false));
    int32_t* v_tU = as.new_var();
    uint8_t* e_tU = as.new_enc();
    msm1.push_back(mk_job(vterms, nullptr, nullptr, orow(o.V, 0), false));
    msm1.push_back(leaving_half(mk_job({ mk_term(orow(o.t, 0), 32, v_U, -1, false) }, nullptr, v_tU, e_tU, false)));   // t*U (issuance.rs:91): hashed, and x_1's base
    as.sccheck(sccheck);
    as.decode(decode);
    as.scalarop(sc1);
    as.msm(msm1);
    // ProofOfIssuance::prove
    iv.n_messages = n;
    Enc one{};
    one[0] = 1;
    iv.w = sv_uniform(c->key_w(), c->host_key[0]); iv.wp = sv_uniform(c->key_wp(), c->host_key[1]);
    iv.x0 = sv_uniform(c->key_x0(), c->host_key[2]); iv.x1 = sv_uniform(c->key_x1(), c->host_key[3]);
    for (uint32_t i = 0; i < n; i++) iv.y[i] = sv_uniform(c->key_y(i), c->host_key[4 + i]);
    iv.one = sv_uniform(c->const_one(), one);
    iv.U = PointVar::Var(v_U, orow(o.U, 0));
    iv.V = PointVar::Var(nullptr, orow(o.V, 0));   // left-hand side only: the prover needs its encoding, not its coordinates
    iv.tU = PointVar::Var(v_tU, e_tU);
    iv.tU.parts.push_back({ orow(o.t, 0), 32, false, v_U, -1 });   // tU = t * U (a segmenting pass multiplies by U instead: SchnorrBuilder::prove_compact)
    SchnorrBuilder p(as, "2019/1416 anonymous credential", "2019/1416 issuance proof");
    issuance_statement(p, c, iv);
    std::vector<afx_hash_program> rng_hash, chal_hash;
    std::vector<afx_msm_job> commit;
    std::vector<afx_scalarop_job> resp, blind_products;
    p.prove_compact(r.rng_seed + off * 32, orow(o.challenge, 0), orow(o.responses, 0), 32 * count, rng_hash, commit, chal_hash, resp, &blind_products);
    as.hash(rng_hash);
    as.scalarop(blind_products);   // (blinding_y * m_i) for the scalar attributes' fixed-base terms
    as.msm(commit);
    as.hash(chal_hash);
    as.scalarop(resp);
    as.finish(status_dev + off, AFX_ST_MAC_CREATION);
  }, key__);
} catch (...) { return afx::exception_rc(); }

// ------------------------------------------------------------------------------------------------
// AnonymousCredential::show
// ------------------------------------------------------------------------------------------------
extern "C" int afx_show_dev(afx_ctx* ctx, const afx_credentials_soa* creds, const afx_keypairs_soa* keypairs, const afx_show_randomness* rnd,
                            size_t count, const afx_presentation_out* out, afx_shape* shape_out, uint8_t* status_dev) try {
  CtxLock lock__(ctx);
  if (!ctx || !creds || !rnd || !out || !shape_out || !status_dev) { set_error("null argument"); return AFX_E_BAD_ARGS; }
  const afx_credentials_soa cr = *creds;
  const uint32_t na = cr.n_attributes;
  if (na == 0 || na > ctx->n) { set_error("credential attribute count does not fit the system parameters"); return AFX_E_BAD_ARGS; }
  for (uint32_t i = 0; i < na; i++)
    if (cr.kinds[i] > AFX_ATTR_SECRET_POINT) { set_error("unknown attribute kind"); return AFX_E_BAD_ARGS; }
  // the presentation's shape (encrypted_attributes kinds, hidden_scalar_indices, proofs_of_encryption indices; :293-320)
  afx_shape sh;
  memset(&sh, 0, sizeof sh);
  sh.n_attributes = na;
  uint32_t hs = 0, nsp = 0;
  for (uint32_t i = 0; i < na; i++) {
    switch (cr.kinds[i]) {
      case AFX_ATTR_PUBLIC_SCALAR: sh.kinds[i] = AFX_ENC_PUBLIC_SCALAR; break;
      case AFX_ATTR_SECRET_SCALAR: sh.kinds[i] = AFX_ENC_SECRET_SCALAR; sh.hidden_scalar_indices[hs++] = (uint16_t)i; break;
      case AFX_ATTR_SECRET_POINT: sh.kinds[i] = AFX_ENC_SECRET_POINT; sh.enc_indices[nsp++] = (uint16_t)i; break;
      default: sh.kinds[i] = AFX_ENC_PUBLIC_POINT; break;
    }
  }
  sh.n_hidden_scalars = hs;
  sh.n_responses = 3 + hs;
  sh.n_enc_proofs = nsp;
  *shape_out = sh;
  if (count == 0) return AFX_OK;
  if (nsp && (!cr.M2 || !cr.m3 || !out->enc)) { set_error("hidden group elements need M2, m3 and enc outputs"); return AFX_E_BAD_ARGS; }
  const bool no_key = nsp && !keypairs;   // CredentialError::NoSymmetricKey (:150-157)
  if (!no_key) {
    bool missing = !cr.values || !cr.t || !cr.U || !cr.V || !rnd->z_wide || !rnd->rng_seed || !out->challenge || !out->responses || !out->C_x_0 ||
                   !out->C_x_1 || !out->C_V || !out->C_y;
    if (nsp) {
      missing |= !rnd->enc_seeds || !keypairs->a || !keypairs->a0 || !keypairs->a1 || !keypairs->pk;
      for (uint32_t e = 0; e < nsp; e++) {
        const afx_encproof_out& q = out->enc[e];
        missing |= !q.challenge || !q.responses || !q.pk || !q.E1 || !q.E2 || !q.C_y_1 || !q.C_y_2 || !q.C_y_3 || !q.C_y_2p;
      }
    }
    if (missing) { set_error("null batch array"); return AFX_E_BAD_ARGS; }
  }
  const afx_keypairs_soa kp = keypairs ? *keypairs : afx_keypairs_soa{ nullptr, nullptr, nullptr, nullptr };
  const afx_show_randomness r = *rnd;
  const afx_presentation_out o = *out;
  std::vector<afx_encproof_out> eo;
  if (nsp) eo.assign(o.enc, o.enc + nsp);
  struct { uint32_t n; uint8_t kinds[AFX_MAX_ATTRIBUTES]; } kd__;
  memset(&kd__, 0, sizeof kd__);
  kd__.n = na; memcpy(kd__.kinds, cr.kinds, std::min<size_t>(na, AFX_MAX_ATTRIBUTES));
  const PlanKey key__ = plan_key("show", &kd__, sizeof kd__, mode_flags(ctx) | (no_key ? 4u : 0u));
  return run_chunked(ctx, count, [&](Assembler& as, size_t off, uint32_t cc) {
    afx_ctx* c = as.ctx;
    as.secret_scalars = true;   // z, the hidden attributes, the symmetric key, every blinding: afx_ctx_set_secret_independent_addressing
    auto row = [&](const uint8_t* base, size_t k) { return base + (k * count + off) * 32; };
    auto orow = [&](uint8_t* base, size_t k) { return base + (k * count + off) * 32; };
    if (no_key) { as.fail_all = true; as.finish(status_dev + off, AFX_ST_NO_SYMMETRIC_KEY); return; }
    std::vector<afx_sccheck_job> sccheck;
    std::vector<afx_decode_job> decode;
    std::vector<afx_scalarop_job> sc1, sc2;
    std::vector<afx_msm_job> msm1, msm1b;
    std::vector<afx_pointop_job> pops;
    uint8_t *z = as.new_enc(), *z_0 = as.new_enc();
    as.reduce_wide(r.z_wide + off * 64, z);                                       // z = Scalar::random (:162)
    sc1.push_back(mk_scalarop(row(cr.t, 0), 32, z, 32, nullptr, 0, true, z_0));   // z_0 = -t*z (:163)
    sccheck.push_back({ row(cr.t, 0) });
    int32_t *v_U = as.new_var(), *v_V = as.new_var();
    decode.push_back({ row(cr.U, 0), v_U, 0 });
    decode.push_back({ row(cr.V, 0), v_V, 0 });
    // commitments (:169-184)
    int32_t* v_Cy[AFX_MAX_ATTRIBUTES];
    int32_t* v_M1[AFX_MAX_ATTRIBUTES] = { nullptr };
    for (uint32_t i = 0; i < na; i++) {
      // a commitment that is only a left-hand side of the statement (every C_y of a kind other than a hidden group element)
      // needs no coordinates: the prover never multiplies it.  Without out_var its job is encoded by k_compress2x.
      v_Cy[i] = cr.kinds[i] == AFX_ATTR_SECRET_POINT ? as.new_var() : nullptr;
      std::vector<afx_msm_term> t = { mk_term(z, 32, nullptr, (int32_t)c->id_Gy(i), false) };
      const int32_t* addend = nullptr;
      if (is_scalar_kind(cr.kinds[i])) {
        sccheck.push_back({ row(cr.values, i) });
        if (cr.kinds[i] == AFX_ATTR_SECRET_SCALAR) t.push_back(mk_term(row(cr.values, i), 32, nullptr, (int32_t)c->id_Gm(i), false));
      } else {
        v_M1[i] = as.new_var();
        decode.push_back({ row(cr.values, i), v_M1[i], 0 });
        if (cr.kinds[i] == AFX_ATTR_SECRET_POINT) addend = v_M1[i];
      }
      msm1.push_back(mk_job(t, addend, v_Cy[i], orow(o.C_y, i), false));
      // revealed values travel with the presentation (:298-302)
      if (cr.kinds[i] != AFX_ATTR_SECRET_SCALAR && cr.kinds[i] != AFX_ATTR_SECRET_POINT && o.attr_values)
        as.copy(orow(o.attr_values, i), row(cr.values, i), 32 * (size_t)cc);
    }
    int32_t *v_Cx0 = as.new_var(), *v_Cx1 = nullptr, *v_Z = nullptr;   // C_x_1 and Z are left-hand sides only
    uint8_t* e_Z = as.new_enc();
    msm1.push_back(mk_job({ mk_term(z, 32, nullptr, (int32_t)c->id_Gx0(), false) }, v_U, v_Cx0, orow(o.C_x_0, 0), false));
    msm1.push_back(mk_job({ mk_term(z, 32, nullptr, (int32_t)c->id_Gx1(), false), mk_term(row(cr.t, 0), 32, v_U, -1, false) }, nullptr, v_Cx1, orow(o.C_x_1, 0), false));
    msm1.push_back(mk_job({ mk_term(z, 32, nullptr, (int32_t)c->id_GV(), false) }, v_V, nullptr, orow(o.C_V, 0), false));
    msm1.push_back(mk_job({ mk_term(z, 32, nullptr, (int32_t)c->id_I(), false) }, nullptr, v_Z, e_Z, false));

    // the presentation proof's statement (:187-273); witnesses z, z_0, t, hidden scalars
    SchnorrBuilder p(as, "2019/1416 anonymous credential", "2019/1416 presentation proof");
    const int zv = p.allocate_scalar("z", sv_item(z));
    const int z_0v = p.allocate_scalar("z_0", sv_item(z_0));
    const int tv = p.allocate_scalar("t", sv_item(row(cr.t, 0)));
    int H_s[AFX_MAX_ATTRIBUTES];
    for (uint32_t j = 0; j < hs; j++) H_s[j] = p.allocate_scalar("m", sv_item(row(cr.values, sh.hidden_scalar_indices[j])));
    const int I = p.allocate_point("I", PointVar::Const(c->id_I()));
    const int C_x_1 = p.allocate_point("C_x_1", PointVar::Var(v_Cx1, orow(o.C_x_1, 0)));
    PointVar pCx0 = PointVar::Var(v_Cx0, orow(o.C_x_0, 0));
    pCx0.parts.push_back({ z, 32, false, nullptr, (int32_t)c->id_Gx0() });   // C_x_0 = z*G_x_0 + U (:171)
    pCx0.parts.push_back({ nullptr, 0, false, v_U, -1 });
    const int C_x_0 = p.allocate_point("C_x_0", pCx0);
    const int G_x_0 = p.allocate_point("G_x_0", PointVar::Const(c->id_Gx0()));
    const int G_x_1 = p.allocate_point("G_x_1", PointVar::Const(c->id_Gx1()));
    int C_y[AFX_MAX_ATTRIBUTES], G_y[AFX_MAX_ATTRIBUTES], G_m[AFX_MAX_ATTRIBUTES];
    uint32_t k = 0, keep[AFX_MAX_ATTRIBUTES];
    for (uint32_t i = 0; i < na; i++)
      if (cr.kinds[i] != AFX_ATTR_SECRET_POINT) { keep[k] = i; C_y[k++] = p.allocate_point("C_y", PointVar::Var(v_Cy[i], orow(o.C_y, i))); }
    for (uint32_t i = 0; i < c->g; i++) G_y[i] = p.allocate_point("G_y", PointVar::Const(c->id_Gy(i)));
    for (uint32_t j = 0; j < hs; j++) G_m[j] = p.allocate_point("G_m", PointVar::Const(c->id_Gm(sh.hidden_scalar_indices[j])));
    // strict mode: C_y[i] - C_y_1 = z*G_y[i] + z*(-G_y[0]) for every hidden group element i != 0 and the C_y_1 of its proof of
    // encryption (the DLEQ of README.md:121-122; the oracle states the reason).  The prover knows the difference in closed form.
    int D[AFX_MAX_ATTRIBUTES], D_pos[AFX_MAX_ATTRIBUTES], nD = 0, neg_G_y_1 = -1;
    if (c->strict)
      for (uint32_t e = 0; e < nsp; e++) {
        const uint32_t i = sh.enc_indices[e];
        if (i == 0) continue;   // the difference is the identity by construction
        if (neg_G_y_1 < 0) neg_G_y_1 = p.allocate_point("-G_y_1", PointVar::Const(c->id_Gy(0), true));
        int32_t* v_D = as.new_var();
        uint8_t* e_D = as.new_enc();
        msm1.push_back(mk_job({ mk_term(z, 32, nullptr, (int32_t)c->id_Gy(i), false), mk_term(z, 32, nullptr, (int32_t)c->id_Gy(0), true) }, nullptr, v_D, e_D, false));
        D[nD] = p.allocate_point("C_y-C_y_1", PointVar::Var(v_D, e_D));
        D_pos[nD++] = (int)i;
      }
    const int Z = p.allocate_point("Z", PointVar::Var(v_Z, e_Z));
    p.constrain(Z, { { zv, I } });
    p.constrain(C_x_1, { { tv, C_x_0 }, { z_0v, G_x_0 }, { zv, G_x_1 } });
    for (uint32_t j = 0; j < k; j++) {
      // the reference uses the compact index j as an original position, literally (:267-273); strict mode uses the
      // commitment's own position (afx_ctx_set_strict)
      const uint32_t q = c->strict ? keep[j] : j;
      if (cr.kinds[q] == AFX_ATTR_SECRET_POINT) continue;
      if (cr.kinds[q] == AFX_ATTR_SECRET_SCALAR) {
        int slot = -1;
        for (uint32_t h = 0; h < hs; h++) if (sh.hidden_scalar_indices[h] == q) slot = (int)h;
        p.constrain(C_y[j], { { zv, G_y[q] }, { H_s[slot], G_m[slot] } });
      } else {
        p.constrain(C_y[j], { { zv, G_y[q] } });
      }
    }
    for (int d = 0; d < nD; d++) p.constrain(D[d], { { zv, G_y[D_pos[d]] }, { zv, neg_G_y_1 } });
    std::vector<afx_hash_program> rng_hash, chal_hash;
    std::vector<afx_msm_job> commit;
    std::vector<afx_scalarop_job> resp, blind_products;   // (blinding * coefficient) of the bases a segmenting pass multiplies part by part
    p.prove_compact(r.rng_seed + off * 32, orow(o.challenge, 0), orow(o.responses, 0), 32 * count, rng_hash, commit, chal_hash, resp, &blind_products);

    // proofs of encryption, one per hidden group element, in attribute order (:293-309 -> encryption.rs:58-142)
    for (uint32_t e = 0; e < nsp; e++) {
      const uint32_t i = sh.enc_indices[e];
      const afx_encproof_out& q = eo[e];
      const uint8_t *a = row(kp.a, 0), *a0 = row(kp.a0, 0), *a1 = row(kp.a1, 0), *m3 = row(cr.m3, i);
      sccheck.push_back({ a }); sccheck.push_back({ a0 }); sccheck.push_back({ a1 }); sccheck.push_back({ m3 });
      int32_t *v_M2 = as.new_var(), *v_pk = as.new_var();
      decode.push_back({ row(cr.M2, i), v_M2, 0 });
      decode.push_back({ row(kp.pk, 0), v_pk, 0 });
      as.copy(orow(q.pk, 0), row(kp.pk, 0), 32 * (size_t)cc);
      uint8_t *kk = as.new_enc(), *z1 = as.new_enc();
      sc1.push_back(mk_scalarop(a1, 32, m3, 32, a0, 32, false, kk));        // a0 + a1*m3
      sc2.push_back(mk_scalarop(z, 32, kk, 32, nullptr, 0, true, z1));      // z1 = -z(a0 + a1*m3) (encryption.rs:78)
      int32_t *v_E1 = as.new_var(), *v_E2 = as.new_var(), *v_C1 = as.new_var(), *v_C2 = as.new_var(), *v_C3 = nullptr /* left-hand side only */,
              *v_C2p = as.new_var(), *v_D1 = as.new_var();
      uint8_t *e_D1 = as.new_enc(), *e_D2 = as.new_enc();
      // Keypair::encrypt (symmetric.rs:252-261): E1 = M2*(a0 + a1*m3), E2 = E1*a + M1
      // E1 and C_y_2' are outputs AND bases of later terms (E2 = a*E1 + M1, the proof's a*(-E1) and m3*C_y_2'): they leave their halves
      msm1.push_back(leaving_half(mk_job({ mk_term(kk, 32, v_M2, -1, false) }, nullptr, v_E1, orow(q.E1, 0), false)));
      // Small passes: a chain that waits for a chain is what a call waits for, and both second-stage results are sums the prover
      // knows term by term: E2 = a*E1 + M1 = (a*kk)*M2 + M1 and C_y_2' = a1*C_y_2 = (a1*z)*G_y_2 + a1*M2 - chains on decoded
      // inputs, beside the first stage's.  (Same group elements, hence the same encodings.)
      const bool flat = as.small();
      uint8_t* a1z_of_flat = nullptr;
      if (flat) {
        uint8_t *akk = as.new_enc(), *a1z = as.new_enc();
        a1z_of_flat = a1z;
        sc2.push_back(mk_scalarop(a, 32, kk, 32, nullptr, 0, false, akk));
        sc1.push_back(mk_scalarop(a1, 32, z, 32, nullptr, 0, false, a1z));
        msm1.push_back(mk_job({ mk_term(akk, 32, v_M2, -1, false) }, v_M1[i], v_E2, orow(q.E2, 0), false));
        msm1.push_back(leaving_half(mk_job({ mk_term(a1z, 32, nullptr, (int32_t)c->id_Gy(1), false), mk_term(a1, 32, v_M2, -1, false) }, nullptr, v_C2p,
                                           orow(q.C_y_2p, 0), false)));
      } else {
        msm1b.push_back(mk_job({ mk_term(a, 32, v_E1, -1, false) }, v_M1[i], v_E2, orow(q.E2, 0), false));
      }
      // C_y_1..3, C_y_2' (encryption.rs:70-75)
      msm1.push_back(mk_job({ mk_term(z, 32, nullptr, (int32_t)c->id_Gy(0), false) }, v_M1[i], v_C1, orow(q.C_y_1, 0), false));
      msm1.push_back(mk_job({ mk_term(z, 32, nullptr, (int32_t)c->id_Gy(1), false) }, v_M2, v_C2, orow(q.C_y_2, 0), false));
      msm1.push_back(mk_job({ mk_term(z, 32, nullptr, (int32_t)c->id_Gy(2), false), mk_term(m3, 32, nullptr, (int32_t)c->id_Gm(i), false) }, nullptr, v_C3, orow(q.C_y_3, 0), false));
      if (!flat) msm1b.push_back(leaving_half(mk_job({ mk_term(a1, 32, v_C2, -1, false) }, nullptr, v_C2p, orow(q.C_y_2p, 0), false)));
      afx_pointop_job d1 = { v_C1, v_E2, nullptr, +1, -1, v_D1, e_D1, 0 };
      pops.push_back(d1);
      as.compress_also(v_E1, e_D2, true, 0);   // only the encoding of -E1 is needed: from E1's half, with msm1's other commitments
      SchnorrBuilder ep(as, "2019/1416 anonymous credentials", "2019/1416 proof of encryption");
      const int sa = ep.allocate_scalar("a", sv_item(a));
      const int sa0 = ep.allocate_scalar("a0", sv_item(a0));
      const int sa1 = ep.allocate_scalar("a1", sv_item(a1));
      const int sm3 = ep.allocate_scalar("m3", sv_item(m3));
      const int sz = ep.allocate_scalar("z", sv_item(z));
      const int sz1 = ep.allocate_scalar("z1", sv_item(z1));
      const int pk = ep.allocate_point("pk", PointVar::Var(v_pk, row(kp.pk, 0)));
      const int G_a = ep.allocate_point("G_a", PointVar::Const(c->id_Ga()));
      const int G_a_0 = ep.allocate_point("G_a_0", PointVar::Const(c->id_Ga0()));
      const int G_a_1 = ep.allocate_point("G_a_1", PointVar::Const(c->id_Ga1()));
      const int G_y_1 = ep.allocate_point("G_y_1", PointVar::Const(c->id_Gy(0)));
      const int G_y_2 = ep.allocate_point("G_y_2", PointVar::Const(c->id_Gy(1)));
      const int G_y_3 = ep.allocate_point("G_y_3", PointVar::Const(c->id_Gy(2)));
      const int G_m_3 = ep.allocate_point("G_m_3", PointVar::Const(c->id_Gm(i)));
      PointVar pC2 = PointVar::Var(v_C2, orow(q.C_y_2, 0)), pC2p = PointVar::Var(v_C2p, orow(q.C_y_2p, 0)), pNegE1 = PointVar::NegOf(v_E1, e_D2);
      if (flat) {
        // what the prover knows these points to be, on the pass's inputs (SchnorrBuilder::prove_compact, segmenting passes):
        // C_y_2 = z*G_y_2 + M2, C_y_2' = a1*C_y_2 = (a1*z)*G_y_2 + a1*M2, -E1 = -(a0 + a1*m3)*M2
        pC2.parts.push_back({ z, 32, false, nullptr, (int32_t)c->id_Gy(1) });
        pC2.parts.push_back({ nullptr, 0, false, v_M2, -1 });
        pC2p.parts.push_back({ a1z_of_flat, 32, false, nullptr, (int32_t)c->id_Gy(1) });
        pC2p.parts.push_back({ a1, 32, false, v_M2, -1 });
        pNegE1.parts.push_back({ kk, 32, true, v_M2, -1 });
      }
      const int C_y_2 = ep.allocate_point("C_y_2", pC2);
      const int C_y_3 = ep.allocate_point("C_y_3", PointVar::Var(v_C3, orow(q.C_y_3, 0)));
      const int C_y_2p = ep.allocate_point("C_y_2'", pC2p);
      const int C_y_1_minus_E2 = ep.allocate_point("C_y_1-E2", PointVar::Var(v_D1, e_D1));
      const int E1 = ep.allocate_point("E1", PointVar::Var(v_E1, orow(q.E1, 0)));
      const int minus_E1 = ep.allocate_point("-E1", pNegE1);   // b_a*(-E1) runs as -(b_a*E1) on E1's window table
      ep.constrain(pk, { { sa, G_a }, { sa0, G_a_0 }, { sa1, G_a_1 } });
      ep.constrain(C_y_1_minus_E2, { { sz, G_y_1 }, { sa, minus_E1 } });
      ep.constrain(C_y_2p, { { sa1, C_y_2 } });
      ep.constrain(E1, { { sa0, C_y_2 }, { sm3, C_y_2p }, { sz1, G_y_2 } });
      ep.constrain(C_y_3, { { sz, G_y_3 }, { sm3, G_m_3 } });
      ep.prove_compact(r.enc_seeds + (e * count + off) * 32, orow(q.challenge, 0), orow(q.responses, 0), 32 * count, rng_hash, commit, chal_hash, resp, &blind_products);
    }
    as.sccheck(sccheck);
    as.decode(decode);
    as.scalarop(sc1);
    as.scalarop(sc2);
    as.msm(msm1);
    as.msm(msm1b);
    as.pointop(pops);
    as.hash(rng_hash);
    as.scalarop(blind_products);
    as.msm(commit);
    as.hash(chal_hash);
    as.scalarop(resp);
    as.finish(status_dev + off, AFX_ST_VERIFICATION_FAILURE);
  }, key__);
} catch (...) { return afx::exception_rc(); }

// ------------------------------------------------------------------------------------------------
// host-pointer front ends
// ------------------------------------------------------------------------------------------------
static int fetch(afx_ctx* ctx, void* dst, const uint8_t* src_dev, size_t n) {
  if (!dst || !n) return AFX_OK;
  AFX_HIP(hipMemcpyAsync(dst, src_dev, n, hipMemcpyDeviceToHost, ctx->stream));
  return AFX_OK;
}

// Requests [first, first + n) of a batch of `total` held in host memory (outputs and status are indexed like the inputs:
// item i of the batch lands in element i of every output array).  Slices alternate between the two lanes (statements.hpp).
extern "C" int afx_issue_range(afx_ctx* ctx, const afx_attributes_soa* req, const afx_issue_randomness* rnd, size_t total, size_t first, size_t n,
                               const afx_issuance_soa* out, uint8_t* status) try {
  CtxLock lock__(ctx, true);
  if (!ctx || !req || !rnd || !out || !status) { set_error("null argument"); return AFX_E_BAD_ARGS; }
  if (!rnd->t_wide || !rnd->U_wide || !rnd->rng_seed || !out->t || !out->U || !out->V || !out->challenge || !out->responses || (req->n_attributes && !req->values)) {
    set_error("null batch array");
    return AFX_E_BAD_ARGS;
  }
  if (first > total || n > total - first) { set_error("range outside the batch"); return AFX_E_BAD_ARGS; }
  if (n == 0) return AFX_OK;
  AFX_HIP(hipSetDevice(ctx->device));
  if (!ctx->has_key) { set_error("Issuer::issue needs the issuer key"); return AFX_E_NO_KEY; }
  // wrong attribute count: every request is MacCreation (amacs.rs:285-287); the arrays' extents are not trusted then
  if (req->n_attributes != ctx->n) { memset(status + first, AFX_ST_MAC_CREATION, n); return AFX_OK; }
  const uint32_t na = req->n_attributes, nr = ctx->n + 5;
  struct { uint32_t n; uint8_t kinds[AFX_MAX_ATTRIBUTES]; } jd;   // what makes two calls one pass (statements.hpp host_pipe)
  memset(&jd, 0, sizeof jd);
  jd.n = na; memcpy(jd.kinds, req->kinds, std::min<size_t>(na, AFX_MAX_ATTRIBUTES));
  const PlanKey jkey = plan_key("I", &jd, sizeof jd, mode_flags(ctx));
  return host_pipe(ctx, n, [&](Stager& st, size_t off, size_t sn) -> int {
    const size_t f0 = first + off;
    const size_t dn = st.dev_items(sn);   // the pass's size on the device (small calls: padded to the size their plan is kept for)
    const size_t o_val = st.add_rows(req->values, na, 32, total, f0, sn, dn), o_tw = st.add_rows(rnd->t_wide, 1, 64, total, f0, sn, dn),
                 o_uw = st.add_rows(rnd->U_wide, 1, 64, total, f0, sn, dn), o_seed = st.add_rows(rnd->rng_seed, 1, 32, total, f0, sn, dn),
                 o_t = st.add(nullptr, 32 * dn), o_U = st.add(nullptr, 32 * dn), o_V = st.add(nullptr, 32 * dn), o_ch = st.add(nullptr, 32 * dn),
                 o_rs = st.add(nullptr, 32 * dn * nr), o_st = st.add(nullptr, dn);
    st.plan_fetch(out->t, o_t, 1, 32, total, f0, sn, dn);
    st.plan_fetch(out->U, o_U, 1, 32, total, f0, sn, dn);
    st.plan_fetch(out->V, o_V, 1, 32, total, f0, sn, dn);
    st.plan_fetch(out->challenge, o_ch, 1, 32, total, f0, sn, dn);
    st.plan_fetch(out->responses, o_rs, nr, 32, total, f0, sn, dn);
    st.plan_fetch(status, o_st, 1, 1, total, f0, sn, dn);
    int rc = st.upload();
    if (rc) return rc;
    afx_attributes_soa da = *req;
    da.values = st.dev(o_val);
    afx_issue_randomness dr = { st.dev(o_tw), st.dev(o_uw), st.dev(o_seed) };
    afx_issuance_soa dout = { st.dev(o_t), st.dev(o_U), st.dev(o_V), st.dev(o_ch), st.dev(o_rs) };
    if ((rc = afx_issue_dev(ctx, &da, &dr, dn, &dout, st.dev(o_st)))) return rc;
    return st.fetch_all();
  }, jkey);
} catch (...) { return afx::exception_rc(); }
extern "C" int afx_issue(afx_ctx* ctx, const afx_attributes_soa* req, const afx_issue_randomness* rnd, size_t count,
                         const afx_issuance_soa* out, uint8_t* status) try {
  return afx_issue_range(ctx, req, rnd, count, 0, count, out, status);
} catch (...) { return afx::exception_rc(); }

// Issuances [first, first + n) of a host batch of `total` (user side, CredentialIssuance::verify, /root/reference/src/issuer.rs:48-57)
extern "C" int afx_verify_issuances_range(afx_ctx* ctx, const afx_attributes_soa* attrs, const afx_issuance_soa* iss, uint32_t n_responses,
                                          size_t total, size_t first, size_t n, uint8_t* status) try {
  CtxLock lock__(ctx, true);
  if (!ctx || !attrs || !iss || !status) { set_error("null argument"); return AFX_E_BAD_ARGS; }
  if (!iss->t || !iss->U || !iss->V || !iss->challenge || (n_responses && !iss->responses) || (attrs->n_attributes && !attrs->values)) {
    set_error("null batch array");
    return AFX_E_BAD_ARGS;
  }
  if (first > total || n > total - first) { set_error("range outside the batch"); return AFX_E_BAD_ARGS; }
  if (n == 0) return AFX_OK;
  AFX_HIP(hipSetDevice(ctx->device));
  // shapes every item fails on (zkp: responses.len() != num_scalars; G_m[i] out of range): answer without reading the arrays
  if (attrs->n_attributes > ctx->n || n_responses != ctx->n + 5) { memset(status + first, AFX_ST_VERIFICATION_FAILURE, n); return AFX_OK; }
  const uint32_t na = attrs->n_attributes, nr = n_responses;
  struct { uint32_t n, nr; uint8_t kinds[AFX_MAX_ATTRIBUTES]; } jd;
  memset(&jd, 0, sizeof jd);
  jd.n = na; jd.nr = nr; memcpy(jd.kinds, attrs->kinds, std::min<size_t>(na, AFX_MAX_ATTRIBUTES));
  const PlanKey jkey = plan_key("VI", &jd, sizeof jd, mode_flags(ctx));
  return host_pipe(ctx, n, [&](Stager& st, size_t off, size_t sn) -> int {
    const size_t f0 = first + off;
    const size_t dn = st.dev_items(sn);
    const size_t o_val = st.add_rows(attrs->values, na, 32, total, f0, sn, dn), o_t = st.add_rows(iss->t, 1, 32, total, f0, sn, dn),
                 o_U = st.add_rows(iss->U, 1, 32, total, f0, sn, dn), o_V = st.add_rows(iss->V, 1, 32, total, f0, sn, dn),
                 o_ch = st.add_rows(iss->challenge, 1, 32, total, f0, sn, dn), o_rs = st.add_rows(iss->responses, nr, 32, total, f0, sn, dn),
                 o_st = st.add(nullptr, dn);
    st.plan_fetch(status, o_st, 1, 1, total, f0, sn, dn);
    int rc = st.upload();
    if (rc) return rc;
    afx_attributes_soa da = *attrs;
    da.values = st.dev(o_val);
    afx_issuance_soa di = { st.dev(o_t), st.dev(o_U), st.dev(o_V), st.dev(o_ch), st.dev(o_rs) };
    if ((rc = afx_verify_issuances_dev(ctx, &da, &di, n_responses, dn, st.dev(o_st)))) return rc;
    return st.fetch_all();
  }, jkey);
} catch (...) { return afx::exception_rc(); }
extern "C" int afx_verify_issuances(afx_ctx* ctx, const afx_attributes_soa* attrs, const afx_issuance_soa* iss, uint32_t n_responses,
                                    size_t count, uint8_t* status) try {
  return afx_verify_issuances_range(ctx, attrs, iss, n_responses, count, 0, count, status);
} catch (...) { return afx::exception_rc(); }

// Credentials [first, first + n) of a host batch of `total` (AnonymousCredential::show, /root/reference/src/credential.rs:37-46);
// every output array is indexed like the inputs.  shape_out is the same for every range of one batch.
extern "C" int afx_show_range(afx_ctx* ctx, const afx_credentials_soa* creds, const afx_keypairs_soa* keypairs, const afx_show_randomness* rnd,
                              size_t total, size_t first, size_t n, const afx_presentation_out* out, afx_shape* shape_out, uint8_t* status) try {
  CtxLock lock__(ctx, true);
  if (!ctx || !creds || !rnd || !out || !shape_out || !status) { set_error("null argument"); return AFX_E_BAD_ARGS; }
  const uint32_t na = creds->n_attributes;
  if (na == 0 || na > ctx->n) { set_error("credential attribute count does not fit the system parameters"); return AFX_E_BAD_ARGS; }
  uint32_t hs = 0, nsp = 0;
  for (uint32_t i = 0; i < na; i++) { hs += creds->kinds[i] == AFX_ATTR_SECRET_SCALAR; nsp += creds->kinds[i] == AFX_ATTR_SECRET_POINT; }
  if (!creds->values || !creds->t || !creds->U || !creds->V || !rnd->z_wide || !rnd->rng_seed || !out->challenge || !out->responses ||
      !out->C_x_0 || !out->C_x_1 || !out->C_V || !out->C_y || (nsp && (!rnd->enc_seeds || !out->enc || !creds->M2 || !creds->m3))) {
    set_error("null batch array");
    return AFX_E_BAD_ARGS;
  }
  if (keypairs && nsp && (!keypairs->a || !keypairs->a0 || !keypairs->a1 || !keypairs->pk)) { set_error("null keypair array"); return AFX_E_BAD_ARGS; }
  if (first > total || n > total - first) { set_error("range outside the batch"); return AFX_E_BAD_ARGS; }
  AFX_HIP(hipSetDevice(ctx->device));
  const bool kp = keypairs && nsp;
  if (n == 0) {   // the shape is still reported (an empty batch has one)
    afx_credentials_soa dc = *creds;
    afx_show_randomness dr = *rnd;
    return afx_show_dev(ctx, &dc, kp ? keypairs : nullptr, &dr, 0, out, shape_out, status);
  }
  struct { uint32_t n; uint8_t kinds[AFX_MAX_ATTRIBUTES]; } jd;
  memset(&jd, 0, sizeof jd);
  jd.n = na; memcpy(jd.kinds, creds->kinds, std::min<size_t>(na, AFX_MAX_ATTRIBUTES));
  const PlanKey jkey = plan_key("S", &jd, sizeof jd, mode_flags(ctx) | (kp ? (uint64_t)1 << 63 : 0) | (out->attr_values ? (uint64_t)1 << 62 : 0));
  return host_pipe(ctx, n, [&](Stager& st, size_t off, size_t sn) -> int {
    const size_t f0 = first + off;
    const size_t dn = st.dev_items(sn);
    auto in = [&](const uint8_t* p, size_t rows, size_t elem) { return (p && rows) ? st.add_rows(p, rows, elem, total, f0, sn, dn) : st.reserve(0); };
    auto res = [&](uint8_t* dst, size_t rows, size_t elem) {
      const size_t o = st.add(nullptr, rows * dn * elem);
      st.plan_fetch(dst, o, rows, elem, total, f0, sn, dn);
      return o;
    };
    const size_t o_val = in(creds->values, na, 32), o_M2 = in(nsp ? creds->M2 : nullptr, nsp ? na : 0, 32), o_m3 = in(nsp ? creds->m3 : nullptr, nsp ? na : 0, 32),
                 o_t = in(creds->t, 1, 32), o_U = in(creds->U, 1, 32), o_V = in(creds->V, 1, 32), o_zw = in(rnd->z_wide, 1, 64),
                 o_seed = in(rnd->rng_seed, 1, 32), o_es = in(nsp ? rnd->enc_seeds : nullptr, nsp, 32);
    size_t o_kp[4] = { 0, 0, 0, 0 };
    if (kp) { o_kp[0] = in(keypairs->a, 1, 32); o_kp[1] = in(keypairs->a0, 1, 32); o_kp[2] = in(keypairs->a1, 1, 32); o_kp[3] = in(keypairs->pk, 1, 32); }
    const size_t o_ch = res(out->challenge, 1, 32), o_rs = res(out->responses, 3 + hs, 32), o_x0 = res(out->C_x_0, 1, 32), o_x1 = res(out->C_x_1, 1, 32),
                 o_cv = res(out->C_V, 1, 32), o_cy = res(out->C_y, na, 32), o_av = res(out->attr_values, na, 32), o_st = res(status, 1, 1);
    std::vector<std::array<size_t, 9>> oe(nsp);
    for (uint32_t e = 0; e < nsp; e++) {
      uint8_t* dst[9] = { out->enc[e].challenge, out->enc[e].responses, out->enc[e].pk, out->enc[e].E1, out->enc[e].E2,
                          out->enc[e].C_y_1, out->enc[e].C_y_2, out->enc[e].C_y_3, out->enc[e].C_y_2p };
      for (int f = 0; f < 9; f++) oe[e][f] = res(dst[f], f == 1 ? 6 : 1, 32);
    }
    int rc = st.upload();
    if (rc) return rc;
    afx_credentials_soa dc = *creds;
    dc.values = st.dev(o_val); dc.M2 = nsp ? st.dev(o_M2) : nullptr; dc.m3 = nsp ? st.dev(o_m3) : nullptr;
    dc.t = st.dev(o_t); dc.U = st.dev(o_U); dc.V = st.dev(o_V);
    afx_keypairs_soa dk = { st.dev(o_kp[0]), st.dev(o_kp[1]), st.dev(o_kp[2]), st.dev(o_kp[3]) };
    afx_show_randomness dr = { st.dev(o_zw), st.dev(o_seed), st.dev(o_es) };
    std::vector<afx_encproof_out> de(nsp);
    for (uint32_t e = 0; e < nsp; e++)
      de[e] = { st.dev(oe[e][0]), st.dev(oe[e][1]), st.dev(oe[e][2]), st.dev(oe[e][3]), st.dev(oe[e][4]), st.dev(oe[e][5]), st.dev(oe[e][6]), st.dev(oe[e][7]), st.dev(oe[e][8]) };
    afx_presentation_out dout = { st.dev(o_ch), st.dev(o_rs), st.dev(o_x0), st.dev(o_x1), st.dev(o_cv), st.dev(o_cy), st.dev(o_av), de.data() };
    if ((rc = afx_show_dev(ctx, &dc, kp ? &dk : nullptr, &dr, dn, &dout, shape_out, st.dev(o_st)))) return rc;
    return st.fetch_all();
  }, jkey);
} catch (...) { return afx::exception_rc(); }
extern "C" int afx_show(afx_ctx* ctx, const afx_credentials_soa* creds, const afx_keypairs_soa* keypairs, const afx_show_randomness* rnd,
                        size_t count, const afx_presentation_out* out, afx_shape* shape_out, uint8_t* status) try {
  return afx_show_range(ctx, creds, keypairs, rnd, count, 0, count, out, shape_out, status);
} catch (...) { return afx::exception_rc(); }

// ------------------------------------------------------------------------------------------------
// IssuerParameters::generate + W = w*G_w
// ------------------------------------------------------------------------------------------------
extern "C" int afx_issuer_keygen(int device, const uint8_t* sysparams, size_t sysparams_len, const uint8_t* key_scalars, size_t key_scalars_len,
                                 uint8_t W_out[32], uint8_t issuer_params_out[64]) try {
  if (!sysparams || !key_scalars || !W_out || !issuer_params_out) { set_error("null argument"); return AFX_E_BAD_ARGS; }
  if (sysparams_len < 4) { set_error("SystemParameters too short"); return AFX_E_BAD_PARAMS; }
  const uint32_t n = (uint32_t)sysparams[0] | ((uint32_t)sysparams[1] << 8) | ((uint32_t)sysparams[2] << 16) | ((uint32_t)sysparams[3] << 24);
  if (n == 0 || n > AFX_MAX_ATTRIBUTES || key_scalars_len != 4 + 32 * (size_t)(4 + n) || memcmp(key_scalars, sysparams, 4) != 0) {
    set_error("key scalar block length / attribute count");
    return AFX_E_BAD_PARAMS;
  }
  afx_ctx* c = nullptr;
  int rc = afx_ctx_create_impl(&c, device, sysparams, sysparams_len, nullptr, 0, key_scalars, nullptr);
  if (rc) return rc;
  uint8_t buf[96];
  {
  // the Stager refers to the context: it must be gone before afx_ctx_destroy(c)
  Stager st(c);
  const size_t o_out = st.add(nullptr, 96), o_st = st.add(nullptr, 1);
  rc = st.upload();
  if (!rc) rc = run_chunked(c, 1, [&](Assembler& as, size_t, uint32_t) {
    std::vector<afx_msm_job> jobs;
    jobs.push_back(mk_job({ mk_term(c->key_w(), 0, nullptr, (int32_t)c->id_Gw(), false) }, nullptr, nullptr, st.dev(o_out), false));   // W (amacs.rs:104)
    jobs.push_back(mk_job({ mk_term(c->key_w(), 0, nullptr, (int32_t)c->id_Gw(), false), mk_term(c->key_wp(), 0, nullptr, (int32_t)c->id_Gwp(), false) },
                          nullptr, nullptr, st.dev(o_out) + 32, false));                                                               // C_W (parameters.rs:350-351)
    std::vector<afx_msm_term> it = { mk_term(c->const_one(), 0, nullptr, (int32_t)c->id_GV(), false), mk_term(c->key_x0(), 0, nullptr, (int32_t)c->id_Gx0(), true),
                                     mk_term(c->key_x1(), 0, nullptr, (int32_t)c->id_Gx1(), true) };
    for (uint32_t i = 0; i < c->n; i++) it.push_back(mk_term(c->key_y(i), 0, nullptr, (int32_t)c->id_Gy(i), true));
    jobs.push_back(mk_job(it, nullptr, nullptr, st.dev(o_out) + 64, false));                                                           // I (parameters.rs:353-359)
    as.msm(jobs);
    as.finish(st.dev(o_st), 1);
  });
  if (!rc) {
    hipError_t e = hipMemcpyAsync(buf, st.dev(o_out), 96, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e != hipSuccess) { set_error(hipGetErrorString(e)); rc = AFX_E_HIP; }
  }
  }
  if (!rc) {
    memcpy(W_out, buf, 32);
    memcpy(issuer_params_out, buf + 32, 64);
  }
  afx_ctx_destroy(c);
  return rc;
} catch (...) { return afx::exception_rc(); }
