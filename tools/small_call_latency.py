"""Latency of ONE synchronous host-pointer call of each operation at small batch sizes, for the two plans
(afx_ctx_set_small_batch_items 0 / 4096 (the default)): python tools/small_call_latency.py
(AFX_LATENCY_ITEMS=1,16,... picks the sizes; AFX_LATENCY_KERNELS=1 adds the per-kernel times of the default plan's call)
Shapes: issue n = 16 (C5's layout), show and verify the C3 shape (8 attributes, 4 hidden encrypted points)."""
import os
import sys
import time
sys.path.insert(0, ".")
import numpy as np
import aeonflux_amd as afx
import bench
from aeonflux_amd import batch


def timed(fn, reps=20):
    fn()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    return (time.perf_counter() - t0) / reps * 1e3


rng = np.random.default_rng(1)
rb = lambda *s: rng.integers(0, 256, size=s, dtype=np.uint8)
p5, k5, i5 = bench.load_fixture("c5_16attrs")
iss5 = afx.Context(p5, k5, i5)
p3, k3, i3 = bench.load_fixture("c3_8attrs_SSPPeeee")
iss3, usr3 = afx.Context(p3, k3, i3), afx.Context(p3, None, i3)
print("%-8s %-28s %-28s %-28s" % ("items", "issue n=16  (0 / 4096) ms", "show C3  (0 / 4096) ms", "verify C3  (0 / 4096) ms"))
for n in [int(x) for x in os.environ.get("AFX_LATENCY_ITEMS", "1,16,256,1024").split(",")]:
    kinds5 = [afx.ATTR_PUBLIC_SCALAR] * 8 + [afx.ATTR_PUBLIC_POINT] * 4 + [afx.ATTR_EITHER_POINT] * 4
    vals5 = np.stack([batch.scalars_from_wide(iss5, rb(n, 64)) if i < 8 else batch.points_from_uniform(iss5, rb(n, 64)) for i in range(16)])
    tw, uw, sd = rb(n, 64), rb(n, 64), rb(n, 32)
    # a C3-shape credential batch and what show needs
    layout, hide = "SSPPEEEE", [4, 5, 6, 7]
    kinds3 = [{"S": afx.ATTR_PUBLIC_SCALAR, "P": afx.ATTR_PUBLIC_POINT, "E": afx.ATTR_EITHER_POINT}[c] for c in layout]
    vals3 = np.stack([batch.scalars_from_wide(iss3, rb(n, 64)) if c == "S" else batch.points_from_uniform(iss3, rb(n, 64)) for c in layout])
    M2 = np.stack([batch.points_from_uniform(iss3, rb(n, 64)) for _ in layout])
    m3 = np.stack([batch.scalars_from_wide(iss3, rb(n, 64)) for _ in layout])
    cred, st = batch.issue(iss3, kinds3, vals3, rb(n, 64), rb(n, 64), rb(n, 32))
    sk = [afx.ATTR_SECRET_POINT if i in hide else k for i, k in enumerate(kinds3)]
    a, a0, a1 = (batch.scalars_from_wide(iss3, rb(n, 64)) for _ in range(3))
    gen = lambda idx: np.frombuffer(p3[4 + 32 * idx:4 + 32 * idx + 32], np.uint8)
    pk, ok = batch.multiscalar_mul(iss3, np.stack([a, a0, a1]), np.stack([np.broadcast_to(gen(5 + 8 + 8 + 1 + k), (n, 32)) for k in range(3)]))
    kp = dict(a=a, a0=a0, a1=a1, pk=pk)
    zw, ssd, es = rb(n, 64), rb(n, 32), rb(4, n, 32)
    pres, shape, st = batch.show(usr3, sk, vals3, cred["t"], cred["U"], cred["V"], kp, zw, ssd, es, M2, m3)
    assert not st.any() and not batch.verify_presentations(iss3, shape, pres).any()
    cols = []
    for name, ctx, fn in (("issue", iss5, lambda: batch.issue(iss5, kinds5, vals5, tw, uw, sd)),
                          ("show", usr3, lambda: batch.show(usr3, sk, vals3, cred["t"], cred["U"], cred["V"], kp, zw, ssd, es, M2, m3)),
                          ("verify", iss3, lambda: batch.verify_presentations(iss3, shape, pres))):
        r = []
        for thr in (0, 4096):
            ctx.set_small_batch_items(thr)
            r.append(timed(fn))
        ctx.set_small_batch_items(4096)
        cols.append("%8.3f / %8.3f" % tuple(r))
        if os.environ.get("AFX_LATENCY_KERNELS"):   # where the default plan's call goes, kernel by kernel (HIP events on the engine's stream)
            ctx.set_timing(True)
            for _ in range(10):
                fn()
            kt = bench.kernel_times(ctx, 10)
            ctx.set_timing(False)
            print("   %s, %d items: kernels %.3f ms in %d launches: %s" % (name, n, sum(v["ms_per_step"] for v in kt.values()), sum(v["launches_per_step"] for v in kt.values()),
                                                                            ", ".join("%s %.3f" % (k, v["ms_per_step"]) for k, v in sorted(kt.items(), key=lambda kv: -kv[1]["ms_per_step"]))))
    print("%-8d %-28s %-28s %-28s" % (n, *cols), flush=True)
