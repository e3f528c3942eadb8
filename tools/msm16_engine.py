#!/usr/bin/env python3
"""The engine's own kernels on the job of tools/ubench/coop_msm.hip: 16-term variable-base MSMs with per-item scalars
(afx_multiscalar_mul; kernel times from the engine's HIP-event timing).  Prints M MSMs/s of table building + chain."""
import sys

import numpy as np

sys.path.insert(0, ".")
import aeonflux_amd as afx  # noqa: E402
import bench  # noqa: E402
from aeonflux_amd import batch  # noqa: E402

lg = int(sys.argv[1]) if len(sys.argv) > 1 else 16
n, T = 1 << lg, 16
params, key, ip = bench.load_fixture("c5_16attrs")
ctx = afx.Context(params, key, ip)
rng = np.random.default_rng(1)
rb = lambda *s: rng.integers(0, 256, size=s, dtype=np.uint8)
pts = np.stack([batch.points_from_uniform(ctx, rb(n, 64)) for _ in range(T)])
scs = np.stack([batch.scalars_from_wide(ctx, rb(n, 64)) for _ in range(T)])
batch.multiscalar_mul(ctx, scs, pts)
ctx.set_timing(True)
reps = 3
for _ in range(reps):
    out, ok = batch.multiscalar_mul(ctx, scs, pts)
assert ok.all()
t = {k: ctx.get_timing(k)[0] / reps for k in ("k_msm_window", "k_msm_tables", "k_decode")}
chain = t["k_msm_window"] + t["k_msm_tables"]
print("# engine kernels, %d-term MSMs, %d of them: %s" % (T, n, {k: round(v, 3) for k, v in t.items()}))
print("engine per-lane Straus (tables + chain) %9.3f ms   %8.3f M MSMs/s" % (chain, n / chain / 1e3))
