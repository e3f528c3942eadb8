"""Small Issuer::verify calls from several host threads, each thread with a context of its own (a context serialises its calls;
contexts have their own streams, so their kernels share the device): calls per second and presentations per second against the
number of contexts, for calls of 64 and 1024 presentations (C3 shape, host pointers, synchronous).
python tools/concurrent_small_calls.py"""
import sys
import threading
import time
sys.path.insert(0, ".")
import numpy as np
import aeonflux_amd as afx
import bench
from aeonflux_amd import batch

params, key, ip = bench.load_fixture("c3_8attrs_SSPPeeee")
gen_i, gen_u = afx.Context(params, key, ip), afx.Context(params, None, ip)
pres, shape = bench.generate(afx, batch, gen_i, gen_u, params, 8, "SSPPEEEE", [4, 5, 6, 7], 1024, 5)
gen_u.close()
gen_i.close()
print("%-8s %-10s %-14s %-18s %-12s" % ("items", "contexts", "calls/s", "presentations/s", "ms per call"))
for n in (64, 1024):
    sub = {f: np.ascontiguousarray(pres[f][..., :n, :]) for f in batch.PRES_FIELDS}
    sub["enc"] = [{f: np.ascontiguousarray(d[f][..., :n, :]) for f in batch.ENC_FIELDS} for d in pres["enc"]]
    for k in (1, 2, 4, 8, 16):
        ctxs = [afx.Context(params, key, ip) for _ in range(k)]
        for c in ctxs:
            assert not batch.verify_presentations(c, shape, sub).any()
        reps, lat = 60, [0.0] * k

        def work(i):
            t0 = time.perf_counter()
            for _ in range(reps):
                batch.verify_presentations(ctxs[i], shape, sub)
            lat[i] = (time.perf_counter() - t0) / reps
        ths = [threading.Thread(target=work, args=(i,)) for i in range(k)]
        t0 = time.perf_counter()
        for t in ths:
            t.start()
        for t in ths:
            t.join()
        dt = time.perf_counter() - t0
        print("%-8d %-10d %-14.0f %-18.0f %-12.3f" % (n, k, k * reps / dt, k * reps * n / dt, 1e3 * sum(lat) / k), flush=True)
        for c in ctxs:
            c.close()
