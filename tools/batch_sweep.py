"""Throughput of afx_verify_presentations_dev against batch size (C3 shape), device-resident inputs.  python tools/batch_sweep.py"""
import ctypes as C
import sys
import time
sys.path.insert(0, ".")
import numpy as np
import torch
import aeonflux_amd as afx
import bench
from aeonflux_amd import batch

params, key, ip = bench.load_fixture("c3_8attrs_SSPPeeee")
issuer, user = afx.Context(params, key, ip), afx.Context(params, None, ip)
N = 1 << 17
pres, shape = bench.generate(afx, batch, issuer, user, params, 8, "SSPPEEEE", [4, 5, 6, 7], N, 5)
dev = torch.device("cuda", 0)
for lg in range(8, 18):
    n = 1 << lg
    sub = {f: torch.from_numpy(np.ascontiguousarray(pres[f][..., :n, :])).to(dev) for f in batch.PRES_FIELDS}
    sub["enc"] = [{f: torch.from_numpy(np.ascontiguousarray(d[f][..., :n, :])).to(dev) for f in batch.ENC_FIELDS} for d in pres["enc"]]
    soa, keep = batch.presentation_soa(sub, ptr=lambda t: t.data_ptr())
    st = torch.zeros(n, dtype=torch.uint8, device=dev)
    call = lambda: afx.check(afx.lib().afx_verify_presentations_dev(issuer.h, C.byref(shape), C.byref(soa), n, st.data_ptr()))
    call(); issuer.synchronize()
    reps = max(3, min(200, (1 << 18) // n))
    t0 = time.perf_counter()
    for _ in range(reps):
        call()
    issuer.synchronize()
    dt = (time.perf_counter() - t0) / reps
    assert not st.cpu().numpy().any()
    print("2^%-2d items  %8.3f ms per call  %7.3f M presentations/s" % (lg, dt * 1e3, n / dt / 1e6))
issuer.close()
user.close()
