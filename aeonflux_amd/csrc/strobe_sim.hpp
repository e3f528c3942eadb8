// Host-side compiler from merlin/STROBE-128 transcript operations to the device byte schedule
// (afx_hash_record).  It never touches secret or per-item data: per-item 32-byte values are HOLES that
// the k_hash kernel fills from struct-of-arrays batches.  Every offset of the transcripts aeonflux builds
// (/root/reference/src/nizk/presentation.rs:355-435, encryption.rs:160-209, issuance.rs:142-217 through zkp's
// TranscriptProtocol [3P]) is data-independent once the statement shape is fixed,
// which is what makes this compilation possible.
#pragma once
#include <stdint.h>
#include <string.h>
#include <stdexcept>
#include <string>
#include <vector>
#include "plan.h"

namespace afx {

inline uint64_t rotl64h(uint64_t x, int n) { return (x << n) | (x >> (64 - n)); }

// Keccak-f[1600] (FIPS 202) on the host: used only for the constant STROBE initialisation state.
inline void keccak_f1600_host(uint64_t a[25]) {
  static const uint64_t RC[24] = {
    0x0000000000000001ULL, 0x0000000000008082ULL, 0x800000000000808aULL, 0x8000000080008000ULL, 0x000000000000808bULL,
    0x0000000080000001ULL, 0x8000000080008081ULL, 0x8000000000008009ULL, 0x000000000000008aULL, 0x0000000000000088ULL,
    0x0000000080008009ULL, 0x000000008000000aULL, 0x000000008000808bULL, 0x800000000000008bULL, 0x8000000000008089ULL,
    0x8000000000008003ULL, 0x8000000000008002ULL, 0x8000000000000080ULL, 0x000000000000800aULL, 0x800000008000000aULL,
    0x8000000080008081ULL, 0x8000000000008080ULL, 0x0000000080000001ULL, 0x8000000080008008ULL };
  for (int round = 0; round < 24; round++) {
    uint64_t c[5], b[25];
    for (int x = 0; x < 5; x++) c[x] = a[x] ^ a[x + 5] ^ a[x + 10] ^ a[x + 15] ^ a[x + 20];
    for (int x = 0; x < 5; x++) {
      const uint64_t d = c[(x + 4) % 5] ^ rotl64h(c[(x + 1) % 5], 1);
      for (int y = 0; y < 25; y += 5) a[y + x] ^= d;
    }
    int x = 1, y = 0;
    b[0] = a[0];
    for (int t = 0; t < 24; t++) {
      const int r = ((t + 1) * (t + 2) / 2) % 64;
      const int nx = y, ny = (2 * x + 3 * y) % 5;
      b[nx + 5 * ny] = r ? rotl64h(a[x + 5 * y], r) : a[x + 5 * y];
      x = nx; y = ny;
    }
    for (int yy = 0; yy < 25; yy += 5)
      for (int xx = 0; xx < 5; xx++) a[yy + xx] = b[yy + xx] ^ (~b[yy + (xx + 1) % 5] & b[yy + (xx + 2) % 5]);
    a[0] ^= RC[round];
  }
}

struct SymByte {
  uint8_t c = 0;         // constant part, XOR-ed (or stored when overwrite)
  int32_t field = -1;    // hole: index into the program's field table
  uint8_t fbyte = 0;     // which byte of the 32-byte field
  bool overwrite = false;
};

struct SimRecord {
  SymByte b[168];
  uint32_t squeeze = AFX_SQ_NONE;
  uint32_t squeeze_out = 0;
};

class StrobeSim {
 public:
  static constexpr int R = 166;
  enum : uint8_t { FLAG_I = 1, FLAG_A = 2, FLAG_C = 4, FLAG_T = 8, FLAG_M = 16, FLAG_K = 32 };

  uint64_t init_state[25];          // concrete state the first record applies to
  std::vector<SimRecord> records;   // completed blocks (each ends with a permutation)
  SimRecord cur;                    // block being filled
  int pos = 0, pos_begin = 0;

  // Strobe128::new(label) then merlin Transcript::new(label): init_state is concrete, the rest is recorded.
  explicit StrobeSim(const char* merlin_label) : StrobeSim((const uint8_t*)merlin_label, strlen(merlin_label)) {}
  StrobeSim(const uint8_t* merlin_label, size_t label_len) {
    // the state after Strobe128::new's permutation is one constant: computed once per process
    struct Init {
      uint64_t w[25];
      Init() {
        uint8_t st[200];
        memset(st, 0, sizeof st);
        const uint8_t hdr[6] = { 1, R + 2, 1, 0, 1, 96 };
        memcpy(st, hdr, 6);
        memcpy(st + 6, "STROBEv1.0.2", 12);
        for (int i = 0; i < 25; i++) {
          w[i] = 0;
          for (int j = 0; j < 8; j++) w[i] |= (uint64_t)st[8 * i + j] << (8 * j);
        }
        keccak_f1600_host(w);
      }
    };
    static const Init init;
    memcpy(init_state, init.w, sizeof init_state);
    meta_ad_const((const uint8_t*)"Merlin v1.0", 11, false);
    append_message_const("dom-sep", merlin_label, label_len);
  }

  // ---- merlin level ----
  void append_message_const(const char* label, const uint8_t* msg, size_t len) { append_message_const((const uint8_t*)label, strlen(label), msg, len); }
  void append_message_const(const uint8_t* label, size_t llen, const uint8_t* msg, size_t len) {
    uint8_t l4[4] = { (uint8_t)len, (uint8_t)(len >> 8), (uint8_t)(len >> 16), (uint8_t)(len >> 24) };
    meta_ad_const(label, llen, false);
    meta_ad_const(l4, 4, true);
    begin_op(FLAG_A, false);
    absorb_const(msg, len);
  }
  void append_message_hole32(const char* label, int field) { append_message_hole32((const uint8_t*)label, strlen(label), field); }
  void append_message_hole32(const uint8_t* label, size_t llen, int field) {
    const uint8_t l4[4] = { 32, 0, 0, 0 };
    meta_ad_const(label, llen, false);
    meta_ad_const(l4, 4, true);
    begin_op(FLAG_A, false);
    absorb_hole32(field);
  }
  // challenge_bytes(label, n bytes), n <= 64, as the LAST operation of a program: the device hands out st[0..64) of the block the
  // forced permutation closes, of which the first n are the challenge (what the sponge would do to the state afterwards - zero the n
  // bytes, go on at position n - no longer matters).  afx_merlin_challenges (statements.cpp): third-party transcript vectors on the GPU.
  void challenge_final(const uint8_t* label, size_t llen, uint32_t n, uint32_t squeeze_kind, uint32_t squeeze_out) {
    if (n > 64) throw std::length_error("a final challenge of more than 64 bytes");
    const uint8_t l4[4] = { (uint8_t)n, 0, 0, 0 };
    meta_ad_const(label, llen, false);
    meta_ad_const(l4, 4, true);
    prf64(squeeze_kind, squeeze_out);
  }
  // challenge_bytes(label, 64 bytes); the squeeze action lands on the record that the forced permutation closes
  void challenge64(const char* label, uint32_t squeeze_kind, uint32_t squeeze_out) {
    const uint8_t l4[4] = { 64, 0, 0, 0 };
    meta_ad_const((const uint8_t*)label, strlen(label), false);
    meta_ad_const(l4, 4, true);
    prf64(squeeze_kind, squeeze_out);
  }

  // challenge_bytes(label, n bytes) somewhere BEFORE the end of a program: nothing is handed out, but the operation acts on the state
  // like any other - its label and length are absorbed, the sponge is permuted, the n bytes it squeezes are zeroed and the position
  // moves behind them (a transcript whose earlier challenges are known - they come back as fields - replayed up to a later one)
  void challenge_discard(const uint8_t* label, size_t llen, uint32_t n) {
    if (n > 64) throw std::length_error("a challenge of more than 64 bytes");
    const uint8_t l4[4] = { (uint8_t)n, 0, 0, 0 };
    meta_ad_const(label, llen, false);
    meta_ad_const(l4, 4, true);
    begin_op(FLAG_I | FLAG_A | FLAG_C, false);
    if (pos != 0) throw std::logic_error("prf not at a block boundary");
    for (uint32_t i = 0; i < n; i++) { cur.b[i].overwrite = true; cur.b[i].c = 0; cur.b[i].field = -1; }
    pos = (int)n;
  }

  // ---- strobe level ----
  void meta_ad_const(const uint8_t* d, size_t n, bool more) { begin_op(FLAG_M | FLAG_A, more); absorb_const(d, n); }
  void key_const(const uint8_t* d, size_t n) { begin_op(FLAG_A | FLAG_C, false); for (size_t i = 0; i < n; i++) overwrite_sym(d[i], -1, 0); }
  void key_hole32(int field) { begin_op(FLAG_A | FLAG_C, false); for (int i = 0; i < 32; i++) overwrite_sym(0, field, (uint8_t)i); }
  void prf64(uint32_t squeeze_kind, uint32_t squeeze_out) {
    begin_op(FLAG_I | FLAG_A | FLAG_C, false);
    if (pos != 0 || records.empty()) throw std::logic_error("prf not at a block boundary");
    records.back().squeeze = squeeze_kind;
    records.back().squeeze_out = squeeze_out;
    // squeeze() hands out st[0..64) and zeroes it
    for (int i = 0; i < 64; i++) { cur.b[i].overwrite = true; cur.b[i].c = 0; cur.b[i].field = -1; }
    pos = 64;
  }
  void absorb_const(const uint8_t* d, size_t n) { for (size_t i = 0; i < n; i++) absorb_sym(d[i], -1, 0); }
  void absorb_hole32(int field) { for (int i = 0; i < 32; i++) absorb_sym(0, field, (uint8_t)i); }

  // translate completed records [from, end) (and nothing else) to device form
  void emit(std::vector<afx_hash_record>& out, size_t from = 0) const {
    for (size_t i = from; i < records.size(); i++) out.push_back(to_device(records[i]));
  }
  // How many leading records act on the state with constants only (no per-item hole, nothing squeezed): labels, lengths, domain
  // separators and batch-constant points - the same for every item, so the host applies them once (fold_prefix) instead of every
  // lane permuting through them (an issuance transcript opens with ~1.8 KB of such bytes: 10 of its 79 permutations).
  size_t constant_prefix() const {
    size_t k = 0;
    for (; k < records.size(); k++) {
      const SimRecord& r = records[k];
      if (r.squeeze != AFX_SQ_NONE) break;
      bool hole = false;
      for (int t = 0; t < 168 && !hole; t++) hole = r.b[t].field >= 0;
      if (hole) break;
    }
    return k;
  }
  // the bytes that determine fold_prefix's result: the key of the context's cache of folded states
  void prefix_key(size_t k, std::string& key) const {
    key.assign((const char*)init_state, sizeof init_state);
    for (size_t i = 0; i < k; i++)
      for (int t = 0; t < 168; t++) { key.push_back((char)records[i].b[t].c); key.push_back(records[i].b[t].overwrite ? 1 : 0); }
  }
  // state after the first k records, exactly as k_hash would compute it: st = (st & keep) ^ c per byte, then Keccak-f
  void fold_prefix(size_t k, uint64_t st[25]) const {
    memcpy(st, init_state, sizeof init_state);
    for (size_t i = 0; i < k; i++) {
      for (int t = 0; t < 168; t++) {
        const SymByte& s = records[i].b[t];
        const int sh = 8 * (t & 7);
        if (s.overwrite) st[t >> 3] &= ~(0xffULL << sh);
        st[t >> 3] ^= (uint64_t)s.c << sh;
      }
      keccak_f1600_host(st);
    }
  }

 private:
  void begin_op(uint8_t flags, bool more) {
    if (more) return;
    const uint8_t old_begin = (uint8_t)pos_begin;
    pos_begin = pos + 1;
    absorb_sym(old_begin, -1, 0);
    absorb_sym(flags, -1, 0);
    const bool force_f = (flags & (FLAG_C | FLAG_K)) != 0;
    if (force_f && pos != 0) run_f();
  }
  void absorb_sym(uint8_t c, int field, uint8_t fbyte) {
    SymByte& s = cur.b[pos];
    if (field >= 0) {
      if (s.field >= 0) throw std::logic_error("two holes on one byte");
      s.field = field; s.fbyte = fbyte;
    } else {
      s.c ^= c;
    }
    if (++pos == R) run_f();
  }
  void overwrite_sym(uint8_t c, int field, uint8_t fbyte) {
    SymByte& s = cur.b[pos];
    s.overwrite = true; s.c = c; s.field = field; s.fbyte = fbyte;
    if (++pos == R) run_f();
  }
  void run_f() {
    cur.b[pos].c ^= (uint8_t)pos_begin;
    cur.b[pos + 1].c ^= 0x04;
    cur.b[R + 1].c ^= 0x80;
    records.push_back(cur);
    cur = SimRecord();
    pos = 0;
    pos_begin = 0;
  }
  static afx_hash_record to_device(const SimRecord& r) {
    afx_hash_record d;
    memset(&d, 0, sizeof d);
    d.squeeze = r.squeeze;
    d.squeeze_out = r.squeeze_out;
    for (int w = 0; w < 21; w++) {
      afx_hash_word& hw = d.w[w];
      hw.c = 0; hw.keep = 0; hw.fmask = 0; hw.field = -1; hw.q = 0; hw.r = 0;
      int delta = 0;
      for (int t = 0; t < 8; t++) {
        const SymByte& s = r.b[8 * w + t];
        hw.c |= (uint64_t)s.c << (8 * t);
        if (!s.overwrite) hw.keep |= 0xffULL << (8 * t);
        if (s.field >= 0) {
          const int dl = (int)s.fbyte - t;
          if (hw.field >= 0 && (hw.field != s.field || dl != delta)) throw std::logic_error("word spans two fields");
          hw.field = s.field;
          delta = dl;
          hw.fmask |= 0xffULL << (8 * t);
        }
      }
      if (hw.field >= 0) {
        // field byte index of stream byte t is delta + t; q = floor(delta / 8), r = delta mod 8
        int q = delta >= 0 ? delta / 8 : -((-delta + 7) / 8);
        int rr = delta - 8 * q;
        hw.q = (int8_t)q;
        hw.r = (uint8_t)rr;
      }
    }
    return d;
  }
};

}  // namespace afx
