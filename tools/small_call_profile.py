"""Per-kernel time of ONE small Issuer::verify call (C3 shape), both plans: python tools/small_call_profile.py [items]"""
import ctypes as C
import sys
sys.path.insert(0, ".")
import numpy as np
import torch
import aeonflux_amd as afx
import bench
from aeonflux_amd import batch

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
params, key, ip = bench.load_fixture("c3_8attrs_SSPPeeee")
issuer, user = afx.Context(params, key, ip), afx.Context(params, None, ip)
pres, shape = bench.generate(afx, batch, issuer, user, params, 8, "SSPPEEEE", [4, 5, 6, 7], n, 5)
dev = torch.device("cuda", 0)
sub = {f: torch.from_numpy(pres[f]).to(dev) for f in batch.PRES_FIELDS}
sub["enc"] = [{f: torch.from_numpy(d[f]).to(dev) for f in batch.ENC_FIELDS} for d in pres["enc"]]
soa, keep = batch.presentation_soa(sub, ptr=lambda t: t.data_ptr())
st = torch.zeros(n, dtype=torch.uint8, device=dev)
for thr in (0, 4096):
    issuer.set_small_batch_items(thr)
    call = lambda: afx.check(afx.lib().afx_verify_presentations_dev(issuer.h, C.byref(shape), C.byref(soa), n, st.data_ptr()))
    call(); issuer.synchronize()
    issuer.set_timing(True)
    reps = 20
    for _ in range(reps):
        call()
    kt = bench.kernel_times(issuer, reps)
    try:
        ms, k = issuer.get_timing("k_pointsum")
        if k:
            kt["k_pointsum"] = {"ms_per_step": ms / reps, "launches_per_step": k / reps}
    except Exception:
        pass
    issuer.set_timing(False)
    tot = sum(v["ms_per_step"] for v in kt.values())
    print("small_batch_items=%d  %d items: kernels %.3f ms in %d launches; plan %s" % (thr, n, tot, sum(v["launches_per_step"] for v in kt.values()), {k: issuer.plan_stats()[k] for k in ("msm_jobs", "doublings", "var_additions")}))
    for k, v in sorted(kt.items(), key=lambda kv: -kv[1]["ms_per_step"]):
        print("   %-16s %.3f ms  x%d" % (k, v["ms_per_step"], v["launches_per_step"]))
issuer.close()
user.close()
