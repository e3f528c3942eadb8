// Host side of the wire formats (include/aeonflux_gpu.h "wire format"): writing AFXP / AFXI batches from the struct-of-arrays
// a prover call returns.  Bytes only - the reference defines no serialisation for these messages
// (/root/reference/src/nizk/presentation.rs:117-127, src/issuer.rs:42-45).
#include <string.h>
#include "statements.hpp"

namespace {
void wr32(uint8_t* p, uint32_t v) { p[0] = (uint8_t)v; p[1] = (uint8_t)(v >> 8); p[2] = (uint8_t)(v >> 16); p[3] = (uint8_t)(v >> 24); }
}

extern "C" int afx_wire_pack_presentations(const afx_shape* shape, const afx_presentation_soa* batch, size_t count, uint8_t* blob, size_t blob_cap,
                                           size_t* len_out) try {
  if (!shape || !batch || !len_out) { set_error("null argument"); return AFX_E_BAD_ARGS; }
  const size_t hdr = afx_wire_header_bytes(shape);
  const uint32_t cells = afx_wire_cells_per_record(shape);
  if (hdr == 0 || cells == 0 || count > 0xffffffffu) { set_error("shape out of range"); return AFX_E_BAD_ARGS; }
  const size_t len = hdr + count * cells * 32;
  *len_out = len;
  if (!blob) return AFX_OK;   // size query
  if (blob_cap < len) { set_error("blob buffer too small"); return AFX_E_BAD_ARGS; }
  const afx_shape& sh = *shape;
  const afx_presentation_soa& b = *batch;
  bool missing = count && (!b.challenge || !b.C_x_0 || !b.C_x_1 || !b.C_V || (sh.n_attributes && !b.C_y) || (sh.n_responses && !b.responses) || (sh.n_enc_proofs && !b.enc));
  for (uint32_t i = 0; i < sh.n_attributes; i++)
    if ((sh.kinds[i] == AFX_ENC_PUBLIC_SCALAR || sh.kinds[i] == AFX_ENC_PUBLIC_POINT) && count && !b.attr_values) missing = true;
  for (uint32_t e = 0; e < sh.n_enc_proofs && !missing && count; e++) {
    const afx_encproof_soa& q = b.enc[e];
    missing |= !q.challenge || !q.responses || !q.pk || !q.E1 || !q.E2 || !q.C_y_1 || !q.C_y_2 || !q.C_y_3 || !q.C_y_2p;
  }
  if (missing) { set_error("null batch array"); return AFX_E_BAD_ARGS; }
  memset(blob, 0, hdr);
  memcpy(blob, "AFXP", 4);
  wr32(blob + 4, 1); wr32(blob + 8, (uint32_t)count); wr32(blob + 12, cells);
  wr32(blob + 16, sh.n_attributes); wr32(blob + 20, sh.n_responses); wr32(blob + 24, sh.n_hidden_scalars); wr32(blob + 28, sh.n_enc_proofs);
  uint8_t* p = blob + 32;
  for (uint32_t i = 0; i < sh.n_attributes; i++) *p++ = sh.kinds[i];
  for (uint32_t i = 0; i < sh.n_hidden_scalars; i++) { *p++ = (uint8_t)sh.hidden_scalar_indices[i]; *p++ = (uint8_t)(sh.hidden_scalar_indices[i] >> 8); }
  for (uint32_t i = 0; i < sh.n_enc_proofs; i++) { *p++ = (uint8_t)sh.enc_indices[i]; *p++ = (uint8_t)(sh.enc_indices[i] >> 8); }
  // the record's cells, in order: where each comes from in the struct-of-arrays ([row][count][32])
  std::vector<const uint8_t*> col;
  auto rows = [&](const uint8_t* base, uint32_t k) { for (uint32_t r = 0; r < k; r++) col.push_back(base + (size_t)r * count * 32); };
  rows(b.challenge, 1); rows(b.responses, sh.n_responses); rows(b.C_x_0, 1); rows(b.C_x_1, 1); rows(b.C_V, 1); rows(b.C_y, sh.n_attributes);
  for (uint32_t i = 0; i < sh.n_attributes; i++)
    if (sh.kinds[i] == AFX_ENC_PUBLIC_SCALAR || sh.kinds[i] == AFX_ENC_PUBLIC_POINT) col.push_back(b.attr_values + (size_t)i * count * 32);
  for (uint32_t e = 0; e < sh.n_enc_proofs; e++) {
    const afx_encproof_soa& q = b.enc[e];
    rows(q.challenge, 1); rows(q.responses, 6); rows(q.pk, 1); rows(q.E1, 1); rows(q.E2, 1); rows(q.C_y_1, 1); rows(q.C_y_2, 1); rows(q.C_y_3, 1); rows(q.C_y_2p, 1);
  }
  if (col.size() != cells) { set_error("internal: wire cell list"); return AFX_E_BAD_ARGS; }
  uint8_t* rec = blob + hdr;
  for (size_t i = 0; i < count; i++)
    for (uint32_t c = 0; c < cells; c++, rec += 32) memcpy(rec, col[c] + i * 32, 32);
  return AFX_OK;
} catch (...) { return afx::exception_rc(); }

extern "C" int afx_issuance_wire_pack(const afx_attributes_soa* attrs, const afx_issuance_soa* issuances, uint32_t n_responses, size_t count,
                                      uint8_t* blob, size_t blob_cap, size_t* len_out) try {
  if (!attrs || !issuances || !len_out) { set_error("null argument"); return AFX_E_BAD_ARGS; }
  const uint32_t n = attrs->n_attributes;
  const size_t hdr = afx_issuance_wire_header_bytes(n);
  if (hdr == 0 || n_responses > AFX_MAX_ATTRIBUTES + 5 || count > 0xffffffffu) { set_error("layout out of range"); return AFX_E_BAD_ARGS; }
  const uint32_t cells = 4 + n_responses + n;
  const size_t len = hdr + count * cells * 32;
  *len_out = len;
  if (!blob) return AFX_OK;   // size query
  if (blob_cap < len) { set_error("blob buffer too small"); return AFX_E_BAD_ARGS; }
  const afx_issuance_soa& s = *issuances;
  if (count && (!s.t || !s.U || !s.V || !s.challenge || (n_responses && !s.responses) || (n && !attrs->values))) { set_error("null batch array"); return AFX_E_BAD_ARGS; }
  for (uint32_t i = 0; i < n; i++)
    if (attrs->kinds[i] > AFX_ATTR_SECRET_POINT) { set_error("attribute kind out of range"); return AFX_E_BAD_ARGS; }
  memset(blob, 0, hdr);
  memcpy(blob, "AFXI", 4);
  wr32(blob + 4, 1); wr32(blob + 8, (uint32_t)count); wr32(blob + 12, cells); wr32(blob + 16, n); wr32(blob + 20, n_responses);
  memcpy(blob + 24, attrs->kinds, n);
  std::vector<const uint8_t*> col = { s.t, s.U, s.V, s.challenge };
  for (uint32_t r = 0; r < n_responses; r++) col.push_back(s.responses + (size_t)r * count * 32);
  for (uint32_t i = 0; i < n; i++) col.push_back(attrs->values + (size_t)i * count * 32);
  uint8_t* rec = blob + hdr;
  for (size_t i = 0; i < count; i++)
    for (uint32_t c = 0; c < cells; c++, rec += 32) memcpy(rec, col[c] + i * 32, 32);
  return AFX_OK;
} catch (...) { return afx::exception_rc(); }
