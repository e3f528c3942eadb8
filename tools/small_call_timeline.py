"""The launch-by-launch timeline of ONE small host-pointer call (1 item): which kernels its stages run and how long each takes.
Run under rocprofv3 --kernel-trace; then tools/small_call_timeline.py --parse <kernel_trace.csv> prints the last call's launches
in order (start offset, duration, gap to the previous launch).
   rocprofv3 --kernel-trace -d gpurun_out/tl -o tl --output-format csv -- python3 tools/small_call_timeline.py issue
   python tools/small_call_timeline.py --parse gpurun_out/tl/*/tl_kernel_trace.csv"""
import csv
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.chdir(ROOT)


def parse(path):
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    # calls are separated by gaps > 200 us (the host's round trip); print the last complete call
    calls, cur = [], []
    for r in rows:
        if cur and int(r["Start_Timestamp"]) - int(cur[-1]["End_Timestamp"]) > 200000:
            calls.append(cur); cur = []
        cur.append(r)
    calls.append(cur)
    print("launches per call:", [len(c) for c in calls])
    call = [c for c in calls if len(c) >= 8][-1]
    t0 = int(call[0]["Start_Timestamp"])
    prev_end = t0
    print("%d calls seen; the last: %d launches, %.3f ms from first start to last end" % (len(calls), len(call), (int(call[-1]["End_Timestamp"]) - t0) / 1e6))
    for r in call:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        name = r["Kernel_Name"].split("(")[0]
        print("  +%8.1f us  %7.1f us  gap %5.1f  grid %-8s wg %-5s %s" % ((s - t0) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3, r.get("Grid_Size_X", r.get("Grid_Size", "?")), r.get("Workgroup_Size_X", r.get("Workgroup_Size", "?")), name[:90]))
        prev_end = e


if len(sys.argv) > 2 and sys.argv[1] == "--parse":
    parse(sys.argv[2])
    sys.exit(0)

import time
import numpy as np
import aeonflux_amd as afx
import bench
from aeonflux_amd import batch

op = sys.argv[1] if len(sys.argv) > 1 else "issue"
n = 1
rng = np.random.default_rng(1)
rb = lambda *s: rng.integers(0, 256, size=s, dtype=np.uint8)
if op == "issue":
    p5, k5, i5 = bench.load_fixture("c5_16attrs")
    ctx = afx.Context(p5, k5, i5)
    kinds5 = [afx.ATTR_PUBLIC_SCALAR] * 8 + [afx.ATTR_PUBLIC_POINT] * 4 + [afx.ATTR_EITHER_POINT] * 4
    vals5 = np.stack([batch.scalars_from_wide(ctx, rb(n, 64)) if i < 8 else batch.points_from_uniform(ctx, rb(n, 64)) for i in range(16)])
    tw, uw, sd = rb(n, 64), rb(n, 64), rb(n, 32)
    fn = lambda: batch.issue(ctx, kinds5, vals5, tw, uw, sd)
else:
    p3, k3, i3 = bench.load_fixture("c3_8attrs_SSPPeeee")
    iss3, ctx = afx.Context(p3, k3, i3), afx.Context(p3, None, i3)
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
    if op == "show":
        fn = lambda: batch.show(ctx, sk, vals3, cred["t"], cred["U"], cred["V"], kp, zw, ssd, es, M2, m3)
    else:
        pres, shape, st = batch.show(ctx, sk, vals3, cred["t"], cred["U"], cred["V"], kp, zw, ssd, es, M2, m3)
        fn = lambda: batch.verify_presentations(iss3, shape, pres)
for _ in range(5):
    fn()
    time.sleep(0.002)
time.sleep(0.01)
fn()
