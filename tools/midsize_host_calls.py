"""One synchronous host-pointer afx_verify_presentations call (C3 shape) at mid sizes, ms per call: the cost of bringing ~75 input
rows to the device (python tools/midsize_host_calls.py; up to 16 MB the rows are gathered into one pinned image and sent in one
copy instead of 75 copies from pageable memory: Stager::PACK_LIMIT, chosen with this tool in round 4 - profiles/r04_midsize_host_calls.txt)."""
import os, sys, time
sys.path.insert(0, ".")
import numpy as np
import aeonflux_amd as afx
import bench
from aeonflux_amd import batch
params, key, ip = bench.load_fixture("c3_8attrs_SSPPeeee")
issuer, user = afx.Context(params, key, ip), afx.Context(params, None, ip)
N = 1 << 15
parts = [bench.generate(afx, batch, issuer, user, params, 8, "SSPPEEEE", [4, 5, 6, 7], min(N, 1 << 15), 5, fast_tables=True)]
pres, shape = parts[0]
out = []
for n in (1 << 10, 1 << 11, 1 << 12, 1 << 13, 1 << 14, 1 << 15):
    p = {f: np.ascontiguousarray(pres[f][..., :n, :]) for f in batch.PRES_FIELDS}
    p["enc"] = [{f: np.ascontiguousarray(d[f][..., :n, :]) for f in batch.ENC_FIELDS} for d in pres["enc"]]
    soa, keep = batch.presentation_soa(p)
    st = np.zeros(n, np.uint8)
    import ctypes as C
    f = lambda: afx.check(afx.lib().afx_verify_presentations(issuer.h, C.byref(shape), C.byref(soa), n, st.ctypes.data))
    f(); f()
    t0 = time.perf_counter()
    for _ in range(10):
        f()
    out.append("%d: %.3f" % (n, (time.perf_counter() - t0) / 10 * 1e3))
print("ms per host-pointer call:  " + "   ".join(out))
