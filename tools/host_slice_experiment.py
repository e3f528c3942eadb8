"""Mid-size host-pointer calls: one slice (one lane) against two or four slices that alternate between the context's two lanes
(afx_ctx_set_chunk_items sets the slice size): python tools/host_slice_experiment.py"""
import sys, time
sys.path.insert(0, ".")
import numpy as np
import aeonflux_amd as afx, bench
from aeonflux_amd import batch
params, key, ip = bench.load_fixture("c3_8attrs_SSPPeeee")
issuer, user = afx.Context(params, key, ip), afx.Context(params, None, ip)
N = 1 << 17
parts = [bench.generate(afx, batch, issuer, user, params, 8, "SSPPEEEE", [4, 5, 6, 7], 1 << 16, 5 + o) for o in range(0, N, 1 << 16)]
shape = parts[0][1]
pres = {f: np.concatenate([p[0][f] for p in parts], axis=-2) for f in batch.PRES_FIELDS}
pres["enc"] = [{f: np.concatenate([p[0]["enc"][e][f] for p in parts], axis=-2) for f in batch.ENC_FIELDS} for e in range(shape.n_enc_proofs)]
print("%-8s %s" % ("items", "ms per host-pointer call with 1 / 2 / 4 / 8 slices"))
for lg in range(11, 18):
    n = 1 << lg
    sub = {f: np.ascontiguousarray(pres[f][..., :n, :]) for f in batch.PRES_FIELDS}
    sub["enc"] = [{f: np.ascontiguousarray(d[f][..., :n, :]) for f in batch.ENC_FIELDS} for d in pres["enc"]]
    out = []
    for k in (1, 2, 4, 8):
        issuer.set_chunk_items(max(256, n // k))
        assert not batch.verify_presentations(issuer, shape, sub).any()
        reps = max(3, min(20, (1 << 17) // n))
        t0 = time.perf_counter()
        for _ in range(reps):
            batch.verify_presentations(issuer, shape, sub)
        out.append((time.perf_counter() - t0) / reps * 1e3)
    issuer.set_chunk_items(0)
    print("2^%-6d %s" % (lg, "  ".join("%8.3f" % x for x in out)), flush=True)
