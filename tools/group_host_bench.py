import sys, time
sys.path.insert(0, ".")
import numpy as np
import aeonflux_amd as afx, bench
from aeonflux_amd import batch
params, key, ip = bench.load_fixture("c3_8attrs_SSPPeeee")
issuer = afx.Context(params, key, ip); user = afx.Context(params, None, ip)
count, chunk = 1 << 20, 1 << 16
parts = [bench.generate(afx, batch, issuer, user, params, 8, "SSPPEEEE", [4, 5, 6, 7], chunk, 100 + o) for o in range(0, count, chunk)]
shape = parts[0][1]
pres = {f: np.concatenate([p[0][f] for p in parts], axis=-2) for f in batch.PRES_FIELDS}
pres["enc"] = [{f: np.concatenate([p[0]["enc"][e][f] for p in parts], axis=-2) for f in batch.ENC_FIELDS} for e in range(4)]
want = bench.corrupt(pres, count, 3)
user.close()
for name, ctx in (("one context", issuer), ("group [0, 0]", afx.Group(params, key, ip, [0, 0]))):
    batch.verify_presentations(ctx, shape, pres)
    t0 = time.perf_counter(); st = batch.verify_presentations(ctx, shape, pres); dt = time.perf_counter() - t0
    assert np.array_equal(st, want)
    print("%-14s host-pointer verify of 2^20 C3 presentations: %.3f s = %.2f M/s" % (name, dt, count / dt / 1e6))
