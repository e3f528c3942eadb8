"""Small calls from several host threads.

python tools/concurrent_small_calls.py --one-context [--threads 1,2,4,...] [--items 1] [--ops verify,issue,show]
    K threads make synchronous host-pointer calls of `items` items on ONE context - what a server behind the crate's `&self`
    methods does (Issuer::verify, /root/reference/src/issuer.rs:141-147) - through the native driver tools/coalesce_drive.cpp
    (no Python in the timed loop).  Every call's statuses and output bytes are compared with those of one whole-batch call.
    Beside every row: the same K threads on K contexts (the round-4 escape hatch), the one context with collection switched
    off (afx_ctx_set_coalescing(ctx, 0, 0): the round-4 library), and K x the CPU oracle's single-thread rate.
python tools/concurrent_small_calls.py
    the round-3/4 table: a context per thread, calls of 64 and 1024 presentations, from Python threads."""
import argparse
import json
import os
import struct
import subprocess
import sys
import tempfile
import threading
import time
sys.path.insert(0, ".")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import aeonflux_amd as afx
import bench
from aeonflux_amd import batch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def write_dump(path, arrays):
    with open(path, "wb") as f:
        f.write(b"AFXD" + struct.pack("<I", len(arrays)))
        for name, a in arrays.items():
            b = a if isinstance(a, (bytes, bytearray)) else np.ascontiguousarray(a).tobytes()
            f.write(name.encode().ljust(32, b"\0") + struct.pack("<Q", len(b)) + b)


def build_driver():
    exe = os.path.join(tempfile.gettempdir(), "afx_coalesce_drive")
    subprocess.run(["g++", "-O2", "-std=c++17", "-pthread", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tools", "coalesce_drive.cpp"), "-o", exe,
                    "-L", os.path.join(ROOT, "aeonflux_amd", "lib"), "-laeonflux_gpu", "-Wl,-rpath," + os.path.join(ROOT, "aeonflux_amd", "lib")], check=True)
    return exe


def dumps(total, tmp):
    """one whole-batch call of each operation on `total` items: inputs + the bytes every small call must reproduce"""
    rng = np.random.default_rng(11)
    rb = lambda *s: rng.integers(0, 256, size=s, dtype=np.uint8)
    out = {}
    # Issuer::verify, C3 shape; 1 in 16 corrupted
    params, key, ip = bench.load_fixture("c3_8attrs_SSPPeeee")
    iss, usr = afx.Context(params, key, ip), afx.Context(params, None, ip)
    pres, shape = bench.generate(afx, batch, iss, usr, params, 8, "SSPPEEEE", [4, 5, 6, 7], total, 5)
    for i in range(0, total, 16):
        pres["responses"][1, i, 3] ^= 1
    want = batch.verify_presentations(iss, shape, pres)
    assert want.sum() == len(range(0, total, 16))
    d = {"count": struct.pack("<I", total), "params": params, "key": key, "ip": ip, "shape": bytes(shape), "want_status": want}
    d.update({f: pres[f] for f in batch.PRES_FIELDS})
    for e, q in enumerate(pres["enc"]):
        d.update({"enc%d_%s" % (e, f): q[f] for f in batch.ENC_FIELDS})
    out["verify"] = os.path.join(tmp, "verify.afxd")
    write_dump(out["verify"], d)
    # AnonymousCredential::show of the same credentials' layout
    layout, hide = "SSPPEEEE", [4, 5, 6, 7]
    kinds = [{"S": afx.ATTR_PUBLIC_SCALAR, "P": afx.ATTR_PUBLIC_POINT, "E": afx.ATTR_EITHER_POINT}[c] for c in layout]
    vals = np.stack([batch.scalars_from_wide(iss, rb(total, 64)) if c == "S" else batch.points_from_uniform(iss, rb(total, 64)) for c in layout])
    M2 = np.stack([batch.points_from_uniform(iss, rb(total, 64)) for _ in layout])
    m3 = np.stack([batch.scalars_from_wide(iss, rb(total, 64)) for _ in layout])
    cred, st = batch.issue(iss, kinds, vals, rb(total, 64), rb(total, 64), rb(total, 32))
    sk = [afx.ATTR_SECRET_POINT if i in hide else k for i, k in enumerate(kinds)]
    a, a0, a1 = (batch.scalars_from_wide(iss, rb(total, 64)) for _ in range(3))
    gen = lambda idx: np.frombuffer(params[4 + 32 * idx:4 + 32 * idx + 32], np.uint8)
    pk, ok = batch.multiscalar_mul(iss, np.stack([a, a0, a1]), np.stack([np.broadcast_to(gen(5 + 8 + 8 + 1 + k), (total, 32)) for k in range(3)]))
    zw, ssd, es = rb(total, 64), rb(total, 32), rb(4, total, 32)
    p2, sh2, st2 = batch.show(usr, sk, vals, cred["t"], cred["U"], cred["V"], dict(a=a, a0=a0, a1=a1, pk=pk), zw, ssd, es, M2, m3)
    assert not st2.any()
    d = {"count": struct.pack("<I", total), "params": params, "key": key, "ip": ip, "n_attributes": struct.pack("<I", 8), "kinds": bytes(sk), "values": vals, "M2": M2, "m3": m3,
         "t": cred["t"], "U": cred["U"], "V": cred["V"], "a": a, "a0": a0, "a1": a1, "pk": pk, "z_wide": zw, "rng_seed": ssd, "enc_seeds": es, "want_status": st2}
    d.update({"want_" + f: p2[f] for f in batch.PRES_FIELDS})
    for e, q in enumerate(p2["enc"]):
        d.update({"want_enc%d_%s" % (e, f): q[f] for f in batch.ENC_FIELDS})
    out["show"] = os.path.join(tmp, "show.afxd")
    write_dump(out["show"], d)
    usr.close()
    iss.close()
    # Issuer::issue, C5 layout (16 attributes)
    p5, k5, i5 = bench.load_fixture("c5_16attrs")
    iss5 = afx.Context(p5, k5, i5)
    kinds5 = [afx.ATTR_PUBLIC_SCALAR] * 8 + [afx.ATTR_PUBLIC_POINT] * 4 + [afx.ATTR_EITHER_POINT] * 4
    vals5 = np.stack([batch.scalars_from_wide(iss5, rb(total, 64)) if i < 8 else batch.points_from_uniform(iss5, rb(total, 64)) for i in range(16)])
    tw, uw, sd = rb(total, 64), rb(total, 64), rb(total, 32)
    o5, st5 = batch.issue(iss5, kinds5, vals5, tw, uw, sd)
    assert not st5.any()
    d = {"count": struct.pack("<I", total), "params": p5, "key": k5, "ip": i5, "n_attributes": struct.pack("<I", 16), "kinds": bytes(kinds5), "values": vals5,
         "t_wide": tw, "U_wide": uw, "rng_seed": sd, "want_status": st5}
    d.update({"want_" + f: o5[f] for f in ("t", "U", "V", "challenge", "responses")})
    out["issue"] = os.path.join(tmp, "issue.afxd")
    write_dump(out["issue"], d)
    iss5.close()
    return out


def cpu_single_thread_verify_rate():
    """the CPU oracle's Issuer::verify rate on one thread (C3 shape): what each of a server's threads does without the engine"""
    import ctypes as C
    import oracle   # checker / CPU baseline only
    params, key, ip = bench.load_fixture("c3_8attrs_SSPPeeee")
    iss, usr = afx.Context(params, key, ip), afx.Context(params, None, ip)
    pres, shape = bench.generate(afx, batch, iss, usr, params, 8, "SSPPEEEE", [4, 5, 6, 7], 128, 5)
    usr.close()
    iss.close()
    olib = oracle.load(native=True)
    olib.afxo_ctx_new.restype = C.c_void_p
    octx = olib.afxo_ctx_new(params, len(params), key, len(key), ip)
    soa, keep = batch.presentation_soa(pres)
    ost = np.full(128, 255, np.uint8)
    t0 = time.perf_counter()
    olib.afxo_verify_presentations_soa(octx, C.byref(oracle.Shape.from_buffer_copy(bytes(shape))), C.byref(oracle.PresentationSoA.from_buffer_copy(bytes(soa))), 128, ost.ctypes.data, 1)
    dt = time.perf_counter() - t0
    assert not ost.any()
    return 128 / dt


def cpu_single_thread_prover_rates():
    """the CPU oracle's Issuer::issue (16 attributes, C5's layout) and AnonymousCredential::show (C3 shape) rates on one thread"""
    from tests.helpers import make_credentials
    out = {}
    d = make_credentials(16, "SSSSSSSSPPPPEEEE", 24, b"cpu-issue-rate")
    cr = d["creds"]
    t0 = time.perf_counter()
    for c in cr:
        assert d["issuer"].issue(c["kinds"], c["values"], *c["rnd"])[0] == 0
    out["issue"] = len(cr) / (time.perf_counter() - t0)
    d = make_credentials(8, "SSPPEEEE", 24, b"cpu-show-rate")
    kinds = [4 if i >= 4 else k for i, k in enumerate(d["creds"][0]["kinds"])]
    take, user = d["take"], d["user"]
    args = [(c, user.keypair_derive(take(64)), take(64), take(32), take(32 * 4)) for c in d["creds"]]
    t0 = time.perf_counter()
    for c, kp, z, sd, es in args:
        assert user.show(kinds, c["values"], c["t"], c["U"], c["V"], kp, z, sd, es)[0] == 0
    out["show"] = len(args) / (time.perf_counter() - t0)
    return out


def one_context(args):
    threads = [int(x) for x in args.threads.split(",")]
    ops = args.ops.split(",")
    exe = build_driver()
    with tempfile.TemporaryDirectory() as tmp:
        files = dumps(max(threads) * args.items, tmp)
        cpu = dict(cpu_single_thread_prover_rates(), verify=cpu_single_thread_verify_rate())
        print("# tools/concurrent_small_calls.py --one-context: K threads x synchronous host-pointer calls of %d item(s) through the C ABI (native driver,"
              " tools/coalesce_drive.cpp), every call's bytes checked against one whole-batch call.  CPU oracle on one thread: Issuer::verify (C3 shape) %.0f/s,"
              " Issuer::issue (16 attributes) %.0f/s, show (C3 shape) %.0f/s" % (args.items, cpu["verify"], cpu["issue"], cpu["show"]))
        print("%-7s %-4s | %-34s | %-22s | %-22s | %-10s" % ("op", "K", "ONE context (calls/s  p50  p99 ms  calls/launch set  lock held: us/call staging, us/set launching)",
                                                              "collection off (calls/s p99)", "K contexts (calls/s p99)", "K x CPU thread"))
        for op in ops:
            for k in threads:
                calls = args.calls if args.calls else max(50, min(400, 6000 // k))
                row = []
                for mode in (["one"], ["one", "0", "0"], ["each"]):
                    r = subprocess.run([exe, files[op], op, str(k), str(calls), str(args.items)] + mode, capture_output=True, text=True, timeout=600)
                    if r.returncode != 0:
                        print("driver failed:", r.stdout[-300:], r.stderr[-600:])
                        raise SystemExit(1)
                    row.append(json.loads(r.stdout.strip().splitlines()[-1]))
                a, b, c = row
                assert not (a["wrong"] or b["wrong"] or c["wrong"])
                print("%-7s %-4d | %8.0f  %6.3f  %6.3f  %6.1f  %5.1f  %6.1f   | %8.0f  %6.3f       | %8.0f  %6.3f       | %8.0f" % (
                    op, k, a["calls_per_s"], a["p50_ms"], a["p99_ms"], a["calls_per_launch_set"], a.get("staging_us_per_call", 0.0), a.get("launch_us_per_set", 0.0),
                    b["calls_per_s"], b["p99_ms"], c["calls_per_s"], c["p99_ms"], cpu[op] * k), flush=True)


def context_each():
    params, key, ip = bench.load_fixture("c3_8attrs_SSPPeeee")
    gen_i, gen_u = afx.Context(params, key, ip), afx.Context(params, None, ip)
    pres, shape = bench.generate(afx, batch, gen_i, gen_u, params, 8, "SSPPEEEE", [4, 5, 6, 7], 1024, 5)
    gen_u.close()
    gen_i.close()
    print("%-8s %-10s %-14s %-18s %-12s" % ("items", "contexts", "calls/s", "presentations/s", "ms per call"))
    for n in (64, 1024):
        sub = {f: np.ascontiguousarray(pres[f][..., :n, :]) for f in batch.PRES_FIELDS}
        sub["enc"] = [{f: np.ascontiguousarray(d[f][..., :n, :]) for f in batch.ENC_FIELDS} for d in pres["enc"]]
        for k in (1, 2, 4, 8, 16):
            ctxs = [afx.Context(params, key, ip) for _ in range(k)]
            for c in ctxs:
                assert not batch.verify_presentations(c, shape, sub).any()
            reps, lat = 60, [0.0] * k

            def work(i):
                t0 = time.perf_counter()
                for _ in range(reps):
                    batch.verify_presentations(ctxs[i], shape, sub)
                lat[i] = (time.perf_counter() - t0) / reps

            ths = [threading.Thread(target=work, args=(i,)) for i in range(k)]
            t0 = time.perf_counter()
            for t in ths:
                t.start()
            for t in ths:
                t.join()
            dt = time.perf_counter() - t0
            print("%-8d %-10d %-14.0f %-18.0f %-12.3f" % (n, k, k * reps / dt, k * reps * n / dt, 1e3 * sum(lat) / k), flush=True)
            for c in ctxs:
                c.close()


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--one-context", action="store_true")
    ap.add_argument("--threads", default="1,2,4,8,16,32,64")
    ap.add_argument("--items", type=int, default=1)
    ap.add_argument("--ops", default="verify,issue,show")
    ap.add_argument("--calls", type=int, default=0, help="calls per thread (default: about 6000 calls per row in all, 50 ... 400 per thread)")
    a = ap.parse_args()
    one_context(a) if a.one_context else context_each()
