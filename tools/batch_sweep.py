"""Time per call of Issuer::verify against batch size (C3 shape), for the two plans (afx_ctx_set_small_batch_items 0 = one chain
per job, 4096 = the default: one chain per term up to 2^12 items, the key job split up to 2^14):
  dev   afx_verify_presentations_dev, inputs resident in HBM, calls queued back to back on the context's stream
  host  afx_verify_presentations, host pointers, one synchronous call (staging, kernels, status back): what a caller of the
        reference's one-presentation Issuer::verify (src/issuer.rs:141-147) waits for
python tools/batch_sweep.py [max_log2]"""
import ctypes as C
import sys
import time
sys.path.insert(0, ".")
import numpy as np
import torch
import aeonflux_amd as afx
import bench
from aeonflux_amd import batch

top = int(sys.argv[1]) if len(sys.argv) > 1 else 17
params, key, ip = bench.load_fixture("c3_8attrs_SSPPeeee")
issuer, user = afx.Context(params, key, ip), afx.Context(params, None, ip)
N = 1 << top
parts = [bench.generate(afx, batch, issuer, user, params, 8, "SSPPEEEE", [4, 5, 6, 7], min(N, 1 << 16), 5 + o) for o in range(0, N, 1 << 16)]
shape = parts[0][1]
pres = {f: np.concatenate([p[0][f] for p in parts], axis=-2) for f in batch.PRES_FIELDS}
pres["enc"] = [{f: np.concatenate([p[0]["enc"][e][f] for p in parts], axis=-2) for f in batch.ENC_FIELDS} for e in range(shape.n_enc_proofs)]
dev = torch.device("cuda", 0)
print("%-10s %s" % ("items", "  ".join("%-26s" % ("small_batch_items=%d: dev / host ms" % t) for t in (0, 4096))))
for lg in range(0, top + 1):
    n = 1 << lg
    hsub = {f: np.ascontiguousarray(pres[f][..., :n, :]) for f in batch.PRES_FIELDS}
    hsub["enc"] = [{f: np.ascontiguousarray(d[f][..., :n, :]) for f in batch.ENC_FIELDS} for d in pres["enc"]]
    sub = {f: torch.from_numpy(hsub[f]).to(dev) for f in batch.PRES_FIELDS}
    sub["enc"] = [{f: torch.from_numpy(d[f]).to(dev) for f in batch.ENC_FIELDS} for d in hsub["enc"]]
    soa, keep = batch.presentation_soa(sub, ptr=lambda t: t.data_ptr())
    hsoa, hkeep = batch.presentation_soa(hsub)
    st = torch.zeros(n, dtype=torch.uint8, device=dev)
    hst = np.zeros(n, np.uint8)
    cols = []
    for thr in (0, 4096):
        issuer.set_small_batch_items(thr)
        call = lambda: afx.check(afx.lib().afx_verify_presentations_dev(issuer.h, C.byref(shape), C.byref(soa), n, st.data_ptr()))
        hcall = lambda: afx.check(afx.lib().afx_verify_presentations(issuer.h, C.byref(shape), C.byref(hsoa), n, hst.ctypes.data))
        call(); issuer.synchronize(); hcall()
        reps = max(3, min(100, (1 << 17) // n))
        t0 = time.perf_counter()
        for _ in range(reps):
            call()
        issuer.synchronize()
        dt = (time.perf_counter() - t0) / reps
        assert not st.cpu().numpy().any()
        hreps = max(3, min(30, (1 << 15) // n))
        t0 = time.perf_counter()
        for _ in range(hreps):
            hcall()
        hdt = (time.perf_counter() - t0) / hreps
        assert not hst.any()
        cols.append("%8.3f / %8.3f  (%6.3f M/s)" % (dt * 1e3, hdt * 1e3, n / dt / 1e6))
    print("2^%-8d %s" % (lg, "  ".join(cols)), flush=True)
issuer.close()
user.close()
