// Native driver for tools/concurrent_small_calls.py: K host threads make small synchronous host-pointer calls through the C ABI -
// Issuer::verify / Issuer::issue / AnonymousCredential::show, `items` items a call - on ONE context (what a server behind the
// crate's `&self` methods does, /root/reference/src/issuer.rs:141-147) or on a context each, and every call's results are compared
// with the bytes the Python side computed for the whole batch in one call.  No Python in the timed loop (a GIL would serialise
// the callers).  Prints one line per run: calls/s, latency percentiles, and how the calls were collected.
//
//   g++ -O2 -std=c++17 -pthread -I include tools/coalesce_drive.cpp -o /tmp/coalesce_drive -L aeonflux_amd/lib -laeonflux_gpu
//   coalesce_drive <dump> <verify|issue|show> <threads> <calls per thread> <items per call> <one|each> [max_wait_us max_items]
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <atomic>
#include <chrono>
#include <map>
#include <string>
#include <thread>
#include <vector>
#include "aeonflux_gpu.h"

typedef std::vector<uint8_t> Bytes;
static std::map<std::string, Bytes> A;   // the dump: named arrays

static void load(const char* path) {
  FILE* f = fopen(path, "rb");
  if (!f) { perror(path); exit(2); }
  char magic[4];
  uint32_t n = 0;
  if (fread(magic, 1, 4, f) != 4 || memcmp(magic, "AFXD", 4) != 0 || fread(&n, 4, 1, f) != 1) { fprintf(stderr, "bad dump\n"); exit(2); }
  for (uint32_t i = 0; i < n; i++) {
    char name[33] = { 0 };
    uint64_t len = 0;
    if (fread(name, 1, 32, f) != 32 || fread(&len, 8, 1, f) != 1) { fprintf(stderr, "bad dump\n"); exit(2); }
    Bytes b(len);
    if (len && fread(b.data(), 1, len, f) != len) { fprintf(stderr, "short dump\n"); exit(2); }
    A[name] = std::move(b);
  }
  fclose(f);
}
static const uint8_t* arr(const std::string& k) {
  auto it = A.find(k);
  if (it == A.end()) { fprintf(stderr, "dump lacks %s\n", k.c_str()); exit(2); }
  return it->second.data();
}
static uint32_t u32(const std::string& k) { uint32_t v; memcpy(&v, arr(k), 4); return v; }

int main(int argc, char** argv) {
  if (argc < 7) { fprintf(stderr, "usage: coalesce_drive <dump> <verify|issue|show> <threads> <calls> <items> <one|each> [max_wait_us max_items]\n"); return 2; }
  load(argv[1]);
  const std::string op = argv[2];
  const int K = atoi(argv[3]), calls = atoi(argv[4]);
  const size_t items = (size_t)atoi(argv[5]);
  const bool one = std::string(argv[6]) == "one";
  const size_t total = u32("count");
  if ((size_t)K * items > total) { fprintf(stderr, "the dump has %zu items, %d threads x %zu need more\n", total, K, items); return 2; }
  const Bytes &params = A["params"], &key = A["key"], &ip = A["ip"];
  const bool user_side = op == "show";
  std::vector<afx_ctx*> ctxs(one ? 1 : K, nullptr);
  for (auto& c : ctxs) {
    const int rc = afx_ctx_create(&c, 0, params.data(), params.size(), user_side ? nullptr : key.data(), user_side ? 0 : key.size(), ip.data());
    if (rc) { fprintf(stderr, "afx_ctx_create: %d %s\n", rc, afx_last_error()); return 1; }
    if (argc >= 9) afx_ctx_set_coalescing(c, (uint32_t)atoi(argv[7]), (uint32_t)atoi(argv[8]));
  }
  // ---- the batch's structures: every thread calls the *_range form on its own items of the SAME arrays
  afx_shape shape;
  memset(&shape, 0, sizeof shape);
  std::vector<afx_encproof_soa> enc;
  afx_presentation_soa pres;
  memset(&pres, 0, sizeof pres);
  afx_attributes_soa req;
  memset(&req, 0, sizeof req);
  afx_issue_randomness irnd = { nullptr, nullptr, nullptr };
  afx_credentials_soa creds;
  memset(&creds, 0, sizeof creds);
  afx_keypairs_soa kp = { nullptr, nullptr, nullptr, nullptr };
  afx_show_randomness srnd = { nullptr, nullptr, nullptr };
  uint32_t nsp = 0, na = 0, nresp = 0;
  if (op == "verify") {
    memcpy(&shape, arr("shape"), sizeof shape);
    enc.resize(shape.n_enc_proofs);
    for (uint32_t e = 0; e < shape.n_enc_proofs; e++) {
      auto f = [&](const char* n) { return arr("enc" + std::to_string(e) + "_" + n); };
      enc[e] = { f("challenge"), f("responses"), f("pk"), f("E1"), f("E2"), f("C_y_1"), f("C_y_2"), f("C_y_3"), f("C_y_2p") };
    }
    pres = { arr("challenge"), arr("responses"), arr("C_x_0"), arr("C_x_1"), arr("C_V"), arr("C_y"), arr("attr_values"), enc.data() };
  } else if (op == "issue") {
    req.n_attributes = na = u32("n_attributes");
    memcpy(req.kinds, arr("kinds"), na);
    req.values = arr("values");
    irnd = { arr("t_wide"), arr("U_wide"), arr("rng_seed") };
    nresp = na + 5;
  } else if (op == "show") {
    creds.n_attributes = na = u32("n_attributes");
    memcpy(creds.kinds, arr("kinds"), na);
    creds.values = arr("values"); creds.M2 = arr("M2"); creds.m3 = arr("m3"); creds.t = arr("t"); creds.U = arr("U"); creds.V = arr("V");
    kp = { arr("a"), arr("a0"), arr("a1"), arr("pk") };
    srnd = { arr("z_wide"), arr("rng_seed"), arr("enc_seeds") };
    uint32_t hs = 0;
    for (uint32_t i = 0; i < na; i++) { nsp += creds.kinds[i] == AFX_ATTR_SECRET_POINT; hs += creds.kinds[i] == AFX_ATTR_SECRET_SCALAR; }
    nresp = 3 + hs;
  } else { fprintf(stderr, "unknown operation\n"); return 2; }

  std::vector<std::vector<double>> lat(K);
  std::atomic<int> bad{ 0 }, ready{ 0 };
  std::atomic<bool> go{ false };
  auto work = [&](int t) {
    afx_ctx* c = ctxs[one ? 0 : t];
    const size_t first = (size_t)t * items;
    std::vector<uint8_t> status(total, 0xee);
    // outputs: whole-batch arrays of this thread's own (it only looks at its items)
    std::map<std::string, Bytes> out;
    auto o = [&](const std::string& n, size_t rows, size_t elem = 32) { Bytes& b = out[n]; b.assign(rows * total * elem, 0xee); return b.data(); };
    afx_issuance_soa iout = { nullptr, nullptr, nullptr, nullptr, nullptr };
    std::vector<afx_encproof_out> eout(nsp);
    afx_presentation_out pout;
    memset(&pout, 0, sizeof pout);
    if (op == "issue") iout = { o("t", 1), o("U", 1), o("V", 1), o("challenge", 1), o("responses", nresp) };
    if (op == "show") {
      for (uint32_t e = 0; e < nsp; e++) {
        auto f = [&](const char* n, size_t rows) { return o("enc" + std::to_string(e) + "_" + n, rows); };
        eout[e] = { f("challenge", 1), f("responses", 6), f("pk", 1), f("E1", 1), f("E2", 1), f("C_y_1", 1), f("C_y_2", 1), f("C_y_3", 1), f("C_y_2p", 1) };
      }
      pout = { o("challenge", 1), o("responses", nresp), o("C_x_0", 1), o("C_x_1", 1), o("C_V", 1), o("C_y", na), o("attr_values", na), eout.data() };
    }
    afx_shape sh_out;
    auto call = [&]() -> int {
      if (op == "verify") return afx_verify_presentations_range(c, &shape, &pres, total, first, items, status.data());
      if (op == "issue") return afx_issue_range(c, &req, &irnd, total, first, items, &iout, status.data());
      return afx_show_range(c, &creds, &kp, &srnd, total, first, items, &pout, &sh_out, status.data());
    };
    auto check = [&]() {
      const uint8_t* want = arr("want_status");
      if (memcmp(status.data() + first, want + first, items) != 0) { bad++; return; }
      for (auto& kv : out) {
        const Bytes& w = A["want_" + kv.first];
        if (w.size() != kv.second.size()) { bad++; return; }
        const size_t rows = w.size() / total / 32;
        for (size_t r = 0; r < rows; r++)
          if (memcmp(kv.second.data() + (r * total + first) * 32, w.data() + (r * total + first) * 32, items * 32) != 0) { bad++; return; }
      }
    };
    for (int w = 0; w < 3; w++) { if (call()) { bad++; fprintf(stderr, "call: %s\n", afx_last_error()); } }
    check();
    ready++;
    while (!go.load()) std::this_thread::yield();
    lat[t].reserve(calls);
    for (int i = 0; i < calls; i++) {
      const auto t0 = std::chrono::steady_clock::now();
      const int rc = call();
      lat[t].push_back(std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
      if (rc) { bad++; fprintf(stderr, "call: %s\n", afx_last_error()); }
    }
    check();
  };
  std::vector<std::thread> ths;
  for (int t = 0; t < K; t++) ths.emplace_back(work, t);
  while (ready.load() < K) std::this_thread::yield();
  afx_coalescing_stats s0;
  memset(&s0, 0, sizeof s0);
  if (one) afx_ctx_get_coalescing_stats(ctxs[0], &s0);
  const auto t0 = std::chrono::steady_clock::now();
  go = true;
  for (auto& t : ths) t.join();
  const double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  afx_coalescing_stats s1;
  memset(&s1, 0, sizeof s1);
  if (one) afx_ctx_get_coalescing_stats(ctxs[0], &s1);
  std::vector<double> all;
  for (auto& v : lat) all.insert(all.end(), v.begin(), v.end());
  std::sort(all.begin(), all.end());
  auto pct = [&](double p) { return all.empty() ? 0.0 : all[std::min(all.size() - 1, (size_t)(p * all.size()))]; };
  const double n_calls = (double)K * calls;
  printf("{\"op\": \"%s\", \"threads\": %d, \"contexts\": %d, \"items_per_call\": %zu, \"calls_per_s\": %.0f, \"items_per_s\": %.0f, \"p50_ms\": %.3f, \"p99_ms\": %.3f, "
         "\"max_ms\": %.3f, \"launch_sets\": %llu, \"calls_per_launch_set\": %.1f, \"appended_calls\": %llu, \"staging_us_per_call\": %.2f, \"launch_us_per_set\": %.1f, \"wrong\": %d}\n",
         op.c_str(), K, one ? 1 : K, items, n_calls / secs, n_calls * items / secs, pct(0.50), pct(0.99), all.empty() ? 0.0 : all.back(),
         (unsigned long long)(s1.sessions - s0.sessions), s1.sessions > s0.sessions ? (double)(s1.calls - s0.calls) / (double)(s1.sessions - s0.sessions) : 0.0,
         (unsigned long long)(s1.appended_calls - s0.appended_calls),
         s1.calls > s0.calls ? 1e-3 * (double)(s1.staging_ns - s0.staging_ns) / (double)(s1.calls - s0.calls) : 0.0,
         s1.sessions > s0.sessions ? 1e-3 * (double)(s1.launch_ns - s0.launch_ns) / (double)(s1.sessions - s0.sessions) : 0.0, bad.load());
  for (auto c : ctxs) afx_ctx_destroy(c);
  return bad.load() ? 1 : 0;
}
