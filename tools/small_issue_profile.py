import sys, time
sys.path.insert(0, ".")
import numpy as np
import aeonflux_amd as afx, bench
from aeonflux_amd import batch
rng = np.random.default_rng(1)
rb = lambda *s: rng.integers(0, 256, size=s, dtype=np.uint8)
p5, k5, i5 = bench.load_fixture("c5_16attrs")
ctx = afx.Context(p5, k5, i5)
n = 256
kinds = [afx.ATTR_PUBLIC_SCALAR] * 8 + [afx.ATTR_PUBLIC_POINT] * 4 + [afx.ATTR_EITHER_POINT] * 4
vals = np.stack([batch.scalars_from_wide(ctx, rb(n, 64)) if i < 8 else batch.points_from_uniform(ctx, rb(n, 64)) for i in range(16)])
tw, uw, sd = rb(n, 64), rb(n, 64), rb(n, 32)
names = bench.MSM_KERNELS + bench.OTHER_KERNELS
for thr in (0, 4096):
    ctx.set_small_batch_items(thr)
    batch.issue(ctx, kinds, vals, tw, uw, sd)
    ctx.set_timing(True)
    reps = 10
    for _ in range(reps):
        batch.issue(ctx, kinds, vals, tw, uw, sd)
    kt = bench.kernel_times(ctx, reps)
    ctx.set_timing(False)
    print("thr", thr, "kernels total %.3f ms, launches %d" % (sum(v["ms_per_step"] for v in kt.values()), sum(v["launches_per_step"] for v in kt.values())), ctx.plan_stats())
    for k, v in sorted(kt.items(), key=lambda kv: -kv[1]["ms_per_step"]):
        print("   %-16s %.3f ms x%d" % (k, v["ms_per_step"], v["launches_per_step"]))
