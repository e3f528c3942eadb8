// TEST INFRASTRUCTURE (CPU only): mutation fuzzer for the entry points that take network bytes, linked with the engine's host
// sources and the fake HIP runtime (fake_hip.cpp) under AddressSanitizer + UBSan by tests/test_wire_fuzz.py.
//   wire_fuzz <dir> <mutations>
// <dir> holds params.bin, key.bin, ip.bin and the valid streams a.afxp, b.afxp, mixed.afxp, i.afxi written by the test from
// the Python packers.  Every mutated stream goes through afx_wire_parse, afx_wire_section_bytes, afx_verify_presentations_wire,
// afx_verify_presentations_mixed_wire, afx_issuance_wire_parse and afx_verify_issuances_wire in an EXACT-size heap buffer (an
// over-read of one byte lands in a red zone); every call must return AFX_OK or AFX_E_BAD_ARGS.
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <string>
#include <vector>
#include "../../include/aeonflux_gpu.h"

typedef std::vector<uint8_t> Bytes;
static Bytes rd(const std::string& p) {
  FILE* f = fopen(p.c_str(), "rb");
  if (!f) { fprintf(stderr, "cannot read %s\n", p.c_str()); exit(2); }
  Bytes v;
  uint8_t buf[4096];
  size_t n;
  while ((n = fread(buf, 1, sizeof buf, f)) > 0) v.insert(v.end(), buf, buf + n);
  fclose(f);
  return v;
}
static uint64_t rng_state = 0x20261003ull;
static uint64_t rnd() { rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17; return rng_state; }
static afx_ctx* ctx;
static std::vector<uint8_t> status(1 << 16);
static unsigned long long calls = 0;

static void hit(const Bytes& b) {
  const size_t n = b.size();
  uint8_t* p = (uint8_t*)malloc(n ? n : 1);
  if (n) memcpy(p, b.data(), n);
  size_t cnt = 0, off = 0, sl = 0;
  afx_shape sh;
  uint32_t na = 0, nr = 0;
  uint8_t kinds[AFX_MAX_ATTRIBUTES];
  const int rcs[6] = {
    afx_wire_parse(p, n, &sh, &cnt, &off),
    afx_wire_section_bytes(p, n, &sl),
    afx_verify_presentations_wire(ctx, p, n, status.data(), status.size(), &cnt),
    afx_verify_presentations_mixed_wire(ctx, p, n, status.data(), status.size(), &cnt),
    afx_issuance_wire_parse(p, n, &na, kinds, &nr, &cnt, &off),
    afx_verify_issuances_wire(ctx, p, n, status.data(), status.size(), &cnt),
  };
  for (int i = 0; i < 6; i++)
    if (rcs[i] != AFX_OK && rcs[i] != AFX_E_BAD_ARGS) {
      fprintf(stderr, "entry point %d returned %d (%s) on a %zu-byte stream starting", i, rcs[i], afx_last_error(), n);
      for (size_t k = 0; k < n && k < 48; k++) fprintf(stderr, " %02x", p[k]);
      fprintf(stderr, "\n");
      exit(1);
    }
  free(p);
  calls++;
}
// a merlin transcript script (afx_merlin_challenges): the same rule - AFX_OK or AFX_E_BAD_ARGS, whatever the bytes say
static unsigned long long merlin_calls = 0, merlin_ok = 0;
static void hit_merlin(const Bytes& b) {
  static std::vector<uint8_t> f0(3 * 32, 0x11), f1(3 * 32, 0x22), out(3 * 64);
  const uint8_t* fields[2] = { f0.data(), f1.data() };
  const size_t n = b.size();
  uint8_t* p = (uint8_t*)malloc(n ? n : 1);
  if (n) memcpy(p, b.data(), n);
  const int rc = afx_merlin_challenges(ctx, p, n, fields, 2, 3, out.data());
  if (rc != AFX_OK && rc != AFX_E_BAD_ARGS) {
    fprintf(stderr, "afx_merlin_challenges returned %d (%s) on a %zu-byte script starting", rc, afx_last_error(), n);
    for (size_t k = 0; k < n && k < 48; k++) fprintf(stderr, " %02x", p[k]);
    fprintf(stderr, "\n");
    exit(1);
  }
  merlin_ok += rc == AFX_OK;
  free(p);
  merlin_calls++;
}
static void put32(Bytes& b, size_t at, uint32_t v) { for (int k = 0; k < 4 && at + k < b.size(); k++) b[at + k] = (uint8_t)(v >> (8 * k)); }

int main(int argc, char** argv) {
  if (argc < 3) return 2;
  const std::string dir = argv[1];
  const unsigned long long target = strtoull(argv[2], nullptr, 10);
  const Bytes params = rd(dir + "/params.bin"), key = rd(dir + "/key.bin"), ip = rd(dir + "/ip.bin");
  if (afx_ctx_create(&ctx, 0, params.data(), params.size(), key.data(), key.size(), ip.data())) { fprintf(stderr, "ctx: %s\n", afx_last_error()); return 2; }
  const Bytes A = rd(dir + "/a.afxp"), B = rd(dir + "/b.afxp"), M = rd(dir + "/mixed.afxp"), I = rd(dir + "/i.afxi");
  const std::vector<Bytes> seeds = { A, B, M, I };
  for (const Bytes& g : seeds) hit(g);
  static const uint32_t EDGE[] = { 0, 1, 2, 3, 31, 32, 33, 255, 256, 65535, 65536, 0x7fffffffu, 0x80000000u, 0xfffffffeu, 0xffffffffu, 0x04000000u, 0x08000001u };
  unsigned long long m = 0;
  // (1) every header word of every stream gets every edge value; (2) truncations at every 32-byte boundary and one byte either side
  for (const Bytes& g : seeds) {
    const size_t words = g.size() / 4 < 24 ? g.size() / 4 : 24;
    for (size_t w = 0; w < words; w++)
      for (uint32_t v : EDGE) { Bytes b = g; put32(b, 4 * w, v); hit(b); m++; }
    for (size_t cut = 0; cut <= g.size(); cut += 32)
      for (int d = -1; d <= 1; d++) {
        const long long at = (long long)cut + d;
        if (at < 0 || at > (long long)g.size()) continue;
        hit(Bytes(g.begin(), g.begin() + at)); m++;
      }
  }
  // (3) the second section's header inside the mixed stream, and splices: sections dropped, doubled, swapped, cut in the middle
  for (size_t w = 0; w < 12; w++)
    for (uint32_t v : EDGE) { Bytes b = M; put32(b, A.size() + 4 * w, v); hit(b); m++; }
  const std::vector<Bytes> parts = { A, B, I, Bytes(A.begin(), A.begin() + A.size() / 2), Bytes(B.begin(), B.begin() + 40), Bytes({ 'A', 'F', 'X', 'P' }),
                                     [] { Bytes x(32, 0); memcpy(x.data(), "AFXI", 4); return x; }() };
  for (const Bytes& x : parts)
    for (const Bytes& y : parts) {
      Bytes b = x; b.insert(b.end(), y.begin(), y.end()); hit(b); m++;
      for (size_t k = 0; k < parts.size(); k += 2) { Bytes c = b; c.insert(c.end(), parts[k].begin(), parts[k].end()); hit(c); m++; }
    }
  // (4) random damage until the target: a few bit flips in the first 96 bytes (header + the start of the records), now and then a
  // random truncation or a field copied from elsewhere in the stream
  while (m < target) {
    Bytes b = seeds[rnd() % seeds.size()];
    const size_t span = b.size() < 96 ? b.size() : 96;
    for (int k = 1 + (int)(rnd() % 3); k > 0; k--) { const size_t bit = rnd() % (8 * span); b[bit >> 3] ^= (uint8_t)(1u << (bit & 7)); }
    const unsigned r = (unsigned)(rnd() % 100);
    if (r < 15) b.resize(rnd() % (b.size() + 1));
    else if (r < 25 && b.size() > 8) { const size_t a = rnd() % (b.size() - 4), c = rnd() % (b.size() - 4); memmove(&b[a], &b[c], 4); }
    hit(b); m++;
  }
  // (5) transcript scripts: a valid one (every operation, a 400-byte message across two rate blocks, both fields), every length word
  // and operation byte at every edge value, truncations at every byte, random damage
  Bytes S;
  auto op = [&](uint8_t o, const char* label) { S.push_back(o); const uint32_t l = (uint32_t)strlen(label); for (int k = 0; k < 4; k++) S.push_back((uint8_t)(l >> (8 * k))); S.insert(S.end(), label, label + l); };
  auto u32 = [&](uint32_t v) { for (int k = 0; k < 4; k++) S.push_back((uint8_t)(v >> (8 * k))); };
  op(AFX_MERLIN_NEW, "fuzz protocol");
  op(AFX_MERLIN_APPEND, "msg"); u32(400); S.insert(S.end(), 400, 0x63);
  op(AFX_MERLIN_APPEND_FIELD, "f0"); u32(0);
  op(AFX_MERLIN_CHALLENGE, "mid"); u32(32);
  op(AFX_MERLIN_APPEND, ""); u32(0);
  op(AFX_MERLIN_APPEND_FIELD, "f1"); u32(1);
  op(AFX_MERLIN_CHALLENGE, "chal"); u32(64);
  hit_merlin(S);
  if (merlin_ok != 1) { fprintf(stderr, "the valid script was refused: %s\n", afx_last_error()); return 1; }
  for (size_t at = 0; at + 4 <= S.size(); at++) {
    if (at > 40 && at < 420) continue;   // (inside the 400-byte message: nothing is parsed there)
    for (uint32_t v : EDGE) { Bytes b = S; put32(b, at, v); hit_merlin(b); }
    for (int o = 0; o < 8; o++) { Bytes b = S; b[at] = (uint8_t)o; hit_merlin(b); }
  }
  for (size_t cut = 0; cut <= S.size(); cut++) hit_merlin(Bytes(S.begin(), S.begin() + cut));
  for (unsigned long long k = 0; k < target / 8; k++) {
    Bytes b = S;
    for (int j = 1 + (int)(rnd() % 3); j > 0; j--) { const size_t bit = rnd() % (8 * b.size()); b[bit >> 3] ^= (uint8_t)(1u << (bit & 7)); }
    if (rnd() % 100 < 15) b.resize(rnd() % (b.size() + 1));
    hit_merlin(b);
  }
  afx_ctx_destroy(ctx);
  printf("wire fuzz ok: %llu mutated streams, %llu x 6 entry-point calls; %llu transcript scripts (%llu accepted)\n", m, calls, merlin_calls, merlin_ok);
  return 0;
}
