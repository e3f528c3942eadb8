/* A C host of the engine: the smallest complete caller of the drop-in boundary (include/aeonflux_gpu.h).
 * Verifies a batch of presentations held in host memory on every GPU of the node with one call, the way a C (or, through
 * the same ABI, Rust / Go / Java) server would behind Issuer::verify (/root/reference/src/issuer.rs:141-147).
 * Build: gcc -std=c99 -I include integration/example_verify.c -L aeonflux_amd/lib -laeonflux_gpu -o example_verify
 * (tests/test_abi_and_host.py compiles it; it needs a GPU to run). */
#include <stdio.h>
#include <stdlib.h>
#include "aeonflux_gpu.h"

/* params/key/issuer_params: the byte forms the crate already defines (SystemParameters::to_bytes, amacs::SecretKey::to_bytes,
 * C_W || I); shape + batch: same-shape presentations as struct-of-arrays, see the header. */
int verify_on_all_gpus(const unsigned char* params, size_t params_len, const unsigned char* key, size_t key_len,
                       const unsigned char issuer_params[64], const int* devices, unsigned n_devices, const afx_shape* shape,
                       const afx_presentation_soa* batch, size_t count, unsigned char* status) {
  afx_group* g = NULL;
  int rc = afx_group_create(&g, devices, n_devices, params, params_len, key, key_len, issuer_params);
  if (rc != AFX_OK) { fprintf(stderr, "afx_group_create: %d %s\n", rc, afx_last_error()); return rc; }
  rc = afx_group_verify_presentations(g, shape, batch, count, status);   /* status[i]: AFX_ST_OK or AFX_ST_VERIFICATION_FAILURE */
  if (rc != AFX_OK) fprintf(stderr, "afx_group_verify_presentations: %d %s\n", rc, afx_last_error());
  afx_group_destroy(g);   /* wipes every copy of the key */
  return rc;
}

/* the same with the caller's own threads: each thread takes a contiguous range of the batch on its own context */
int verify_my_range(afx_ctx* ctx, unsigned members, unsigned index, const afx_shape* shape, const afx_presentation_soa* batch,
                    size_t count, unsigned char* status) {
  size_t first = 0, n = 0;
  afx_shard_bounds(count, members, index, &first, &n);
  return afx_verify_presentations_range(ctx, shape, batch, count, first, n, status);
}

/* a request stream as it arrives: serialized presentations of whatever shapes, one AFXP section each (or one per same-shape
 * run), back to back in `stream`.  The library groups them by shape, runs one GPU batch per distinct shape and answers in
 * arrival order: status[i] belongs to the i-th presentation of the stream; *n = how many there were. */
int verify_request_stream(afx_ctx* ctx, const unsigned char* stream, size_t stream_len, unsigned char* status, size_t status_cap, size_t* n) {
  const int rc = afx_verify_presentations_mixed_wire(ctx, stream, stream_len, status, status_cap, n);
  if (rc != AFX_OK) fprintf(stderr, "afx_verify_presentations_mixed_wire: %d %s\n", rc, afx_last_error());
  return rc;
}

/* What a server does when the ACCELERATOR fails under a call (AFX_E_NO_DEVICE, AFX_E_HIP, AFX_E_NO_MEMORY: the device, not the data): it
 * does not answer "verification failed" - that would turn every honest user away for the length of a GPU reset - it hands the same
 * request to its CPU verifier (for the crate: the body of Issuer::verify, /root/reference/src/issuer.rs:146; the Rust shim's
 * `try_verify` / INTEGRATION.md section 1 do exactly this).  `cpu_verify` is that verifier (any implementation with this signature);
 * *fell_through counts the calls that took it.  Return codes about the caller's data (AFX_E_BAD_ARGS ...) go back as they are. */
typedef int (*afx_cpu_verify_fn)(void* cpu_issuer, const unsigned char* stream, size_t stream_len, unsigned char* status, size_t status_cap, size_t* n);
static int is_engine_fault(int rc) { return rc == AFX_E_NO_DEVICE || rc == AFX_E_HIP || rc == AFX_E_NO_MEMORY; }
int verify_request_stream_or_fall_through(afx_ctx* ctx, afx_cpu_verify_fn cpu_verify, void* cpu_issuer, const unsigned char* stream, size_t stream_len,
                                          unsigned char* status, size_t status_cap, size_t* n, unsigned long* fell_through) {
  int rc = ctx ? afx_verify_presentations_mixed_wire(ctx, stream, stream_len, status, status_cap, n) : AFX_E_NO_DEVICE;   /* (no engine at all: the CPU path) */
  if (!is_engine_fault(rc)) return rc;
  fprintf(stderr, "engine fault %d (%s): this request goes to the CPU verifier\n", rc, ctx ? afx_last_error() : "no context");
  if (fell_through) ++*fell_through;
  return cpu_verify ? cpu_verify(cpu_issuer, stream, stream_len, status, status_cap, n) : rc;
}
