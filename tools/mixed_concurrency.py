"""One Issuer::verify request stream of MANY presentation shapes: G shapes x K presentations each, through
afx_verify_presentations_mixed (the shape groups collected into one set of kernel launches: mixed.cpp, afx::Session) against one
afx_verify_presentations call per shape, one after the other (what round 3's library did inside the same entry point).
    python tools/mixed_concurrency.py [groups=64] [items_per_group=16]
Shapes: 8 attributes `S S S S P E E E`; every subset of the four scalars hidden x the last 0..3 group elements hidden = 64 shapes.
Reports ms per request (C call only: the arrays and group structs are built once), and checks that both ways give the same
statuses (a few presentations are damaged)."""
import ctypes as C
import itertools
import sys
import time

sys.path.insert(0, ".")
import numpy as np

import aeonflux_amd as afx
import bench
from aeonflux_amd import batch


def main():
    G = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    K = int(sys.argv[2]) if len(sys.argv) > 2 else 16
    params, key, ip = bench.load_fixture("c3_8attrs_SSPPeeee")
    issuer, user = afx.Context(params, key, ip), afx.Context(params, None, ip)
    layout = "SSSSPEEE"
    hides = []
    for tail in range(4):
        for r in range(5):
            for sub in itertools.combinations(range(4), r):
                hides.append(list(sub) + list(range(8 - tail, 8)))
    hides = hides[:G]
    items, total = [], 0
    for g, hide in enumerate(hides):
        pres, shape = bench.generate(afx, batch, issuer, user, params, 8, layout, hide, K, 5000 + g)
        if g % 5 == 0:
            pres["C_V"][g % K, 3] ^= 1
        items.append((shape, pres))
        total += K
    user.close()
    # group structs once (what batch.verify_mixed builds on every call)
    arr = (afx.PresentationGroup * len(items))()
    keep = []
    for g, (shape, p) in enumerate(items):
        soa, encs = batch.presentation_soa(p)
        pos = np.arange(g, total, len(items), dtype=np.uint64)[:K]   # interleaved: item i of group g stands at i * G + g
        arr[g].shape, arr[g].batch, arr[g].count = shape, soa, K
        arr[g].positions = pos.ctypes.data_as(C.POINTER(C.c_uint64))
        keep.append((soa, encs, pos))
    status = np.full(total, 255, np.uint8)
    L = afx.lib()

    def mixed():
        afx.check(L.afx_verify_presentations_mixed(issuer.h, arr, len(items), status.ctypes.data, total))

    serial_status = np.full(total, 255, np.uint8)
    tmp = np.zeros(K, np.uint8)

    def serial():
        for g in range(len(items)):
            afx.check(L.afx_verify_presentations(issuer.h, C.byref(arr[g].shape), C.byref(arr[g].batch), K, tmp.ctypes.data))
            serial_status[keep[g][2].astype(np.int64)] = tmp

    def timed(fn, reps):
        fn()
        fn()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        return (time.perf_counter() - t0) / reps * 1e3

    t_mixed = timed(mixed, 30)
    t_serial = timed(serial, 5)
    assert np.array_equal(status, serial_status), "statuses differ"
    bad = int(status.sum())
    print("%d shapes x %d presentations = %d items: afx_verify_presentations_mixed %.3f ms per request (%.1f us per shape); one call per shape, in "
          "sequence: %.3f ms (%.3f ms per call); %d rejected, statuses equal" % (len(items), K, total, t_mixed, t_mixed * 1e3 / len(items), t_serial, t_serial / len(items), bad))
    # where the merged request's time goes: kernels (HIP events on the engine's stream) vs everything else
    issuer.set_timing(True)
    for _ in range(10):
        mixed()
    names = ("k_msm_window", "k_msm_tables", "k_table_affine", "k_msm_fixed", "k_msm_naf", "k_pointsum", "k_compress2x", "k_negenc", "k_decode", "k_pointop", "k_hash", "k_scalarop", "k_sccheck",
             "k_finish", "k_fill_u32")
    parts = {}
    for k in names:
        try:
            ms, n = issuer.get_timing(k)
        except Exception:
            continue
        if n:
            parts[k] = (ms / 10, n / 10)
    issuer.set_timing(False)
    print("  kernels of one merged request: %.3f ms in %d launches: %s" % (sum(v[0] for v in parts.values()), int(sum(v[1] for v in parts.values())),
                                                                       ", ".join("%s %.3f" % (k, v[0]) for k, v in sorted(parts.items(), key=lambda kv: -kv[1][0]))))
    issuer.close()


if __name__ == "__main__":
    main()
