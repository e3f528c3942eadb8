#!/usr/bin/env python3
"""Where the host-pointer entry's time goes: afx_verify_presentations (= Issuer::verify over a batch in host memory,
/root/reference/src/issuer.rs:141-147) on C3, 2^20 presentations in pageable numpy arrays, against the device-resident rate of the
same process - by where the CALLING thread runs (unpinned / on the device's NUMA node / on another node), where the caller's
arrays were first touched (near / far), and how the rows travel (afx_ctx_set_host_copy_threads 0 = the runtime's pageable copies
on the caller's thread, N = the context's copy pool).  Five repetitions each: median, min, max.

    python tools/host_pointer_numa.py [--batch 1048576] [--reps 5] > profiles/r06_host_pointer_numa.txt
"""
import argparse
import ctypes as C
import glob
import os
import statistics
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def cpulist(text):
    out = set()
    for tok in text.replace("\n", "").split(","):
        if not tok:
            continue
        a, _, b = tok.partition("-")
        out.update(range(int(a), int(b or a) + 1))
    return out


def topology(bdf):
    nodes = {}
    for p in sorted(glob.glob("/sys/devices/system/node/node[0-9]*")):
        try:
            nodes[int(p.rsplit("node", 1)[1])] = cpulist(open(os.path.join(p, "cpulist")).read())
        except (OSError, ValueError):
            pass
    try:
        dev_node = int(open("/sys/bus/pci/devices/%s/numa_node" % bdf.lower()).read())
    except (OSError, ValueError):
        dev_node = -1
    return nodes, dev_node


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=1 << 20)
    ap.add_argument("--reps", type=int, default=5)
    args = ap.parse_args()
    import torch
    import aeonflux_amd as afx
    from aeonflux_amd import batch
    import bench
    n, layout, hide, _, fixture, desc = bench.WORKLOADS["c3"]
    count = args.batch
    p = torch.cuda.get_device_properties(0)
    bdf = "%04x:%02x:%02x.0" % (p.pci_domain_id, p.pci_bus_id, p.pci_device_id)
    allowed = os.sched_getaffinity(0)
    nodes, dev_node = topology(bdf)
    near = sorted(allowed & nodes.get(dev_node, set()))
    far = sorted(allowed - nodes.get(dev_node, set())) if dev_node >= 0 else []
    print("# tools/host_pointer_numa.py: %s" % desc)
    print("# device %s on NUMA node %d; nodes: %s" % (bdf, dev_node, {k: "%d cpus" % len(v) for k, v in nodes.items()}))
    print("# this process may use %d CPUs (%d usable by quota): %d on the device's node, %d elsewhere; cpu model %s"
          % (len(allowed), bench.usable_cores(), len(near), len(far), bench.cpu_model()))
    params, key, ip = bench.load_fixture(fixture)
    issuer = afx.Context(params, key, ip, device=0)
    user = afx.Context(params, None, ip, device=0)
    parts = [bench.generate(afx, batch, issuer, user, params, n, layout, hide, min(1 << 16, count - o), 1000 + o, fast_tables=True)
             for o in range(0, count, 1 << 16)]
    shape = parts[0][1]
    pres = {f: np.concatenate([q[0][f] for q in parts], axis=-2) for f in batch.PRES_FIELDS}
    pres["enc"] = [{f: np.concatenate([q[0]["enc"][e][f] for q in parts], axis=-2) for f in batch.ENC_FIELDS} for e in range(shape.n_enc_proofs)]
    del parts
    want = bench.corrupt(pres, count, 7)
    user.close()
    issuer.set_secret_independent_addressing(2)
    L = afx.lib()

    # the device-resident rate of this process (what `value` is)
    dev = torch.device("cuda", 0)
    dpres = {f: torch.from_numpy(pres[f]).to(dev) for f in batch.PRES_FIELDS}
    dpres["enc"] = [{f: torch.from_numpy(d[f]).to(dev) for f in batch.ENC_FIELDS} for d in pres["enc"]]
    soa, keep = batch.presentation_soa(dpres, ptr=lambda t: t.data_ptr())
    status = torch.full((count,), 255, dtype=torch.uint8, device=dev)
    for _ in range(2):
        afx.check(L.afx_verify_presentations_dev(issuer.h, C.byref(shape), C.byref(soa), count, status.data_ptr()))
    torch.cuda.synchronize()
    assert np.array_equal(status.cpu().numpy(), want)
    dres = []
    for _ in range(args.reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        afx.check(L.afx_verify_presentations_dev(issuer.h, C.byref(shape), C.byref(soa), count, status.data_ptr()))
        torch.cuda.synchronize()
        dres.append(count / (time.perf_counter() - t0))
    resident = statistics.median(dres)
    print("# device-resident (afx_verify_presentations_dev): median %.0f presentations/s (min %.0f, max %.0f)" % (resident, min(dres), max(dres)))
    del dpres, soa, status
    torch.cuda.empty_cache()

    def copy_of(arrays, cpus):
        """the caller's arrays, first touched from `cpus`"""
        if cpus:
            os.sched_setaffinity(0, cpus)
        out = {f: np.array(arrays[f], copy=True) for f in batch.PRES_FIELDS}
        out["enc"] = [{f: np.array(d[f], copy=True) for f in batch.ENC_FIELDS} for d in arrays["enc"]]
        os.sched_setaffinity(0, allowed)
        return out

    def measure(arrays, caller_cpus, threads):
        issuer.set_host_copy_threads(threads)
        hsoa, keep_h = batch.presentation_soa(arrays)
        st = np.full(count, 255, np.uint8)
        os.sched_setaffinity(0, caller_cpus or allowed)
        try:
            afx.check(L.afx_verify_presentations(issuer.h, C.byref(shape), C.byref(hsoa), count, st.ctypes.data))   # warm-up: buffers, pool
            rates = []
            for _ in range(args.reps):
                st[:] = 255
                t0 = time.perf_counter()
                afx.check(L.afx_verify_presentations(issuer.h, C.byref(shape), C.byref(hsoa), count, st.ctypes.data))
                rates.append(count / (time.perf_counter() - t0))
                assert np.array_equal(st, want)
        finally:
            os.sched_setaffinity(0, allowed)
        return rates

    placements = [("caller unpinned", None)]
    if near:
        placements.append(("caller on the device's node", set(near)))
    if far:
        placements.append(("caller on another node", set(far)))
    memories = [("arrays where numpy put them", pres)]
    if far:
        memories.append(("arrays first touched on another node", copy_of(pres, set(far))))
    if near and far:
        memories.append(("arrays first touched on the device's node", copy_of(pres, set(near))))
    print("%-44s %-34s %8s %12s %12s %12s %8s" % ("arrays", "calling thread", "threads", "median /s", "min /s", "max /s", "of dev"))
    for mname, arrays in memories:
        for pname, cpus in placements:
            for threads in (0, 1, 2, 4, 8):
                r = measure(arrays, cpus, threads)
                med = statistics.median(r)
                print("%-44s %-34s %8d %12.0f %12.0f %12.0f %7.1f%%" % (mname, pname, threads, med, min(r), max(r), 100.0 * med / resident))
                sys.stdout.flush()
    issuer.close()


if __name__ == "__main__":
    main()
