"""Where the plans cross (C3 shape, device-resident): time per call under each of the three plans of Assembler::msm.  python tools/plan_crossover.py"""
import ctypes as C, sys, time
sys.path.insert(0, ".")
import numpy as np, torch
import aeonflux_amd as afx, bench
from aeonflux_amd import batch
params, key, ip = bench.load_fixture("c3_8attrs_SSPPeeee")
issuer, user = afx.Context(params, key, ip), afx.Context(params, None, ip)
N = 1 << 16
pres, shape = bench.generate(afx, batch, issuer, user, params, 8, "SSPPEEEE", [4, 5, 6, 7], N, 5)
dev = torch.device("cuda", 0)
for lg in range(10, 17):
    n = 1 << lg
    sub = {f: torch.from_numpy(np.ascontiguousarray(pres[f][..., :n, :])).to(dev) for f in batch.PRES_FIELDS}
    sub["enc"] = [{f: torch.from_numpy(np.ascontiguousarray(d[f][..., :n, :])).to(dev) for f in batch.ENC_FIELDS} for d in pres["enc"]]
    soa, keep = batch.presentation_soa(sub, ptr=lambda t: t.data_ptr())
    st = torch.zeros(n, dtype=torch.uint8, device=dev)
    out = []
    for thr in (0, 65536, max(256, n // 2)):   # one chain per job / one chain per term / only the key job split (n <= 4 * thr)
        issuer.set_small_batch_items(thr)
        call = lambda: afx.check(afx.lib().afx_verify_presentations_dev(issuer.h, C.byref(shape), C.byref(soa), n, st.data_ptr()))
        call(); issuer.synchronize()
        reps = max(3, min(50, (1 << 18) // n))
        t0 = time.perf_counter()
        for _ in range(reps): call()
        issuer.synchronize()
        out.append((time.perf_counter() - t0) / reps * 1e3)
    print("2^%d  one chain per job %.3f ms   one chain per term %.3f ms   key job split %.3f ms" % (lg, out[0], out[1], out[2]), flush=True)
