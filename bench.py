#!/usr/bin/env python3
"""bench.py — presentations verified per second on MI355X (BASELINE.json metric).

One "step" = one pass of Issuer::verify (afx_verify_presentations_dev) over one batch of synthetic
presentations whose struct-of-arrays fields are already resident in HBM.  N > 1: one process per GPU, each
verifying its own batch (host-sharded, no collective on the data path); value = all ranks' presentations /
max-over-ranks time.  Inputs are produced by the engine itself (GPU issue -> GPU show), 1 % are corrupted, and
every status byte is checked.  The `cpu_baseline` leg times the ORACLE's restated CPU path (kind "port") on a
bounded sample of the same inputs on this machine's host cores and cross-checks the GPU's status bytes.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOADS = {
    # name: (n, issue layout, hidden positions, batch per GPU, fixture flow holding params/key)
    "c2": (4, "SSPE", [0, 3], 1 << 16, "readme_4attrs_sSPe",
           "C2: batch verify 2^16 presentations, 4 attributes (s S P e: 1 hidden scalar, 1 public scalar, 1 public point, "
           "1 hidden encrypted point)"),
    "c3": (8, "SSPPEEEE", [4, 5, 6, 7], 1 << 20, "c3_8attrs_SSPPeeee",
           "C3: batch verify 2^20 presentations, 8 attributes (S S P P e e e e: 4 hidden encrypted points)"),
    # C4 = the C3 statement, 2^22 presentations split over the ranks (strong scaling: 2^22 / N per GPU)
    "c4": (8, "SSPPEEEE", [4, 5, 6, 7], 1 << 22, "c3_8attrs_SSPPeeee",
           "C4: batch verify 2^22 presentations in all, 8 attributes (S S P P e e e e), host-sharded over the ranks"),
}


def algorithmic_bytes(shape):
    """bytes read + 1 status byte written per presentation"""
    n, hs, hp = shape.n_attributes, shape.n_hidden_scalars, shape.n_enc_proofs
    pub = sum(1 for i in range(n) if shape.kinds[i] in (0, 2))
    return 32 * (4 + hs) + 96 + 32 * n + 32 * pub + 448 * hp + (n + 2 * hs + 4 * hp) + 1


def traffic_file(secret=False):
    """the newest committed traffic measurement, profiles/rNN_traffic.json (rNN_secret_traffic.json: the same passes with
    secret-independent addressing on, tools/collect_secret_mode_traffic.sh)"""
    import glob
    found = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_%straffic.json" % ("secret_" if secret else ""))))
    return found[-1] if found else None


# timing experiments with deliberately wrong kernels (tools/ab_bench.sh with an experiment build): skip the result checks
# and say so in the output line.  Never set for a reported number.
UNCHECKED = os.environ.get("AFX_BENCH_UNCHECKED") == "1"


# the device code a traffic measurement belongs to: every file the kernels are compiled from
KERNEL_SOURCES = ("kernels.hip", "fe.cuh", "fe10.cuh", "sc.cuh", "ge.cuh", "keccak.cuh", "constants.cuh", "plan.h")


def kernel_sources_sha256():
    import hashlib
    h = hashlib.sha256()
    for name in KERNEL_SOURCES:
        with open(os.path.join(ROOT, "aeonflux_amd", "csrc", name), "rb") as f:
            h.update(name.encode() + b"\0" + f.read())
    return h.hexdigest()


def traffic_is_stale(secret=False):
    """True when the committed traffic measurement was taken on other device code than this tree's (or says nothing about it):
    tools/traffic_json.py stores the sha256 of the kernel sources it measured"""
    try:
        with open(traffic_file(secret)) as f:
            return json.load(f).get("kernel_sources_sha256") != kernel_sources_sha256()
    except (OSError, ValueError, TypeError):
        return True


def measured_traffic(workload, kernel, secret=False):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 PMC passes of this same command
    (profiles/rNN_traffic.json, the newest round's; collected with tools/collect_profiles.sh: separate FETCH_SIZE / WRITE_SIZE passes,
    FETCH doubled per MI355X_MICROARCH §HBM).  PMC counters cannot be read from inside the process, so the figure is the
    committed measurement of the same workload, not a live one.  C4 launches are C3 launches (2^19-item passes)."""
    try:
        with open(traffic_file(secret)) as f:
            d = json.load(f).get({"c4": "c3"}.get(workload, workload))
        return d.get(kernel) if isinstance(d, dict) else d
    except (OSError, ValueError, TypeError):
        return None


MSM_KERNELS = ("k_msm_window", "k_msm_naf", "k_msm_fixed", "k_msm_tables")
OTHER_KERNELS = ("k_table_affine", "k_compress2x", "k_negenc", "k_pointsum", "k_decode", "k_pointop", "k_hash", "k_scalarop", "k_sccheck", "k_from_uniform", "k_reduce_wide", "k_finish", "k_fill_u32")


# the kernels doing field arithmetic beside the multiscalar ones (the time base of the "valu" figure)
FIELD_KERNELS = ("k_table_affine", "k_compress2x", "k_negenc", "k_pointsum", "k_decode", "k_pointop", "k_from_uniform")


def kernel_times(ctx, steps):
    """per-kernel totals of the timed steps, measured by the engine with HIP events on its own stream"""
    out = {}
    for k in MSM_KERNELS + OTHER_KERNELS:
        try:
            ms, n = ctx.get_timing(k)
        except Exception:   # an older build of the library (same-box A/Bs, tools/ab_bench.sh) may not know the kernel
            continue
        if n:
            out[k] = {"ms_per_step": ms / steps, "launches_per_step": n / steps, "avg_launch_ms": ms / n}
    return out


def roofline_of(kt, workload, ab, items_per_step, secret=False):
    """the contract's roofline object for the dominant kernel of the step (the largest ms_per_step)"""
    dom = max((k for k in kt if k in MSM_KERNELS), key=lambda k: kt[k]["ms_per_step"])
    d = kt[dom]
    items_per_launch = items_per_step / d["launches_per_step"]
    achieved = ab * items_per_launch / (d["avg_launch_ms"] / 1e3) / 1e9
    # PMC counters cannot be read from inside the process: the traffic figure is the committed measurement of this same command
    # (profiles/rNN_traffic.json) - but only while it was taken on THIS tree's device code (its kernel_sources_sha256)
    stale = traffic_is_stale(secret)
    return {"bound": "hbm", "achieved": achieved, "peak": 8000.0, "unit": "GB/s", "frac": achieved / 8000.0,
            "traffic": None if stale else measured_traffic(workload, dom, secret), "traffic_stale": stale,
            "traffic_source": os.path.relpath(traffic_file(secret), ROOT) if traffic_file(secret) else None,
            "kernel": dom, "launches_per_step": d["launches_per_step"],
            "avg_launch_ms": d["avg_launch_ms"], "items_per_launch": items_per_launch, "algorithmic_bytes_per_launch": ab * items_per_launch,
            "kernel_ms_per_step": sum(kt[k]["ms_per_step"] for k in kt if k in MSM_KERNELS),
            "kernel_ms_per_step_is": "the k_msm_* kernels (chains + their table builds) summed: " + " + ".join(k for k in kt if k in MSM_KERNELS) +
                                     "; avg_launch_ms is the dominant kernel's (%s) alone" % dom,
            "kernels_ms_per_step": {k: round(v["ms_per_step"], 4) for k, v in kt.items()},
            "kernel_launches_per_step": {k: v["launches_per_step"] for k, v in kt.items()},
            "note": "integer-ALU bound path: the compute-side figure is in \"valu\""}


def usable_cores():
    """host cores this process may actually use: affinity mask, capped by the cgroup CPU quota"""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, int(float(q) / float(per))))
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, q // per))
        except (OSError, ValueError):
            pass
    return n


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def load_fixture(name):
    with open(os.path.join(ROOT, "tests", "golden", "flows.json")) as f:
        flows = json.load(f)["flows"]
    r = next(x for x in flows if x["name"] == name)
    return bytes.fromhex(r["params"]), bytes.fromhex(r["key"]), bytes.fromhex(r["issuer_params"])


def generate(afx, batch, issuer, user, params, n, layout, hide, count, seed, fast_tables=False):
    """synthetic credentials -> presentations, all arithmetic on the GPU (issue, show).  fast_tables: the two contexts run their
    prover-side calls with afx_ctx_set_secret_independent_addressing 0 from here on (the bench's input generation: synthetic keys
    and nonces are no secrets, and generation is not what is measured; the caller sets the mode it measures afterwards)."""
    if fast_tables:
        issuer.set_secret_independent_addressing(0)
        user.set_secret_independent_addressing(0)
    rng = np.random.default_rng(seed)
    rb = lambda *shape: rng.integers(0, 256, size=shape, dtype=np.uint8)
    values = np.zeros((n, count, 32), np.uint8)
    M2 = np.zeros((n, count, 32), np.uint8)
    m3 = np.zeros((n, count, 32), np.uint8)
    kinds = []
    for i, c in enumerate(layout):
        if c == "S":
            kinds.append(afx.ATTR_PUBLIC_SCALAR)
            values[i] = batch.scalars_from_wide(issuer, rb(count, 64))
        elif c == "P":
            kinds.append(afx.ATTR_PUBLIC_POINT)
            values[i] = batch.points_from_uniform(issuer, rb(count, 64))
        else:  # plaintext attribute: synthetic (M1, M2, m3) — the verifier never sees how they relate
            kinds.append(afx.ATTR_EITHER_POINT)
            values[i] = batch.points_from_uniform(issuer, rb(count, 64))
            M2[i] = batch.points_from_uniform(issuer, rb(count, 64))
            m3[i] = batch.scalars_from_wide(issuer, rb(count, 64))
    iss, st = batch.issue(issuer, kinds, values, rb(count, 64), rb(count, 64), rb(count, 32))
    assert not st.any(), "issue failed"
    skinds = list(kinds)
    for i in hide:
        skinds[i] = afx.ATTR_SECRET_SCALAR if skinds[i] == afx.ATTR_PUBLIC_SCALAR else afx.ATTR_SECRET_POINT
    nsp = sum(1 for k in skinds if k == afx.ATTR_SECRET_POINT)
    g = max(3, n)
    gen = lambda idx: np.frombuffer(params[4 + 32 * idx:4 + 32 * idx + 32], np.uint8)
    a, a0, a1 = (batch.scalars_from_wide(issuer, rb(count, 64)) for _ in range(3))
    bases = np.stack([np.broadcast_to(gen(5 + g + n + 1 + k), (count, 32)) for k in range(3)])
    pk, ok = batch.multiscalar_mul(issuer, np.stack([a, a0, a1]), bases)
    assert ok.all()
    pres, shape, st = batch.show(user, skinds, values, iss["t"], iss["U"], iss["V"], dict(a=a, a0=a0, a1=a1, pk=pk),
                                 rb(count, 64), rb(count, 32), rb(max(nsp, 1), count, 32), M2, m3)
    assert not st.any(), "show failed"
    return pres, shape


def corrupt(pres, count, seed):
    """1 % of the items get one corrupted field; returns the expected status vector"""
    rng = np.random.default_rng(seed)
    want = np.zeros(count, np.uint8)
    idx = rng.choice(count, size=max(1, count // 100), replace=False)
    for j, i in enumerate(idx):
        mode = j % 5
        if mode == 0:
            pres["responses"][0, i, 3] ^= 0x10
        elif mode == 1:
            pres["C_V"][i, 7] ^= 0x01
        elif mode == 2 and pres["enc"]:
            pres["enc"][0]["E1"][i, 11] ^= 0x04
        elif mode == 3:
            pres["C_y"][0, i, :] = 0          # identity commitment
        else:
            pres["C_x_0"][i, :] = 0xFF        # non-canonical encoding
        want[i] = 1
    return want


def bench_issue(args, afx, batch, torch, dist, rank, world, local_rank, embedded=None):
    """secondary metric: credentials issued / s, C5 = 2^20 issuances, 16 attributes S x8 P x4 E x4.
    embedded = {"modes": [...], "steps": K, "warmup": W}: called from the C3 run (N = 1) for the line's config.secondary - the same
    inputs timed under each secret mode, the lines returned instead of printed, no CPU timing leg (the oracle still checks bytes)."""
    n, layout, count = 16, "SSSSSSSSPPPPEEEE", ((args.batch if not embedded else 0) or (1 << 20))
    steps, warmup = (embedded["steps"], embedded["warmup"]) if embedded else (args.steps, args.warmup)
    modes = embedded["modes"] if embedded else [secret_mode_of(args)]
    params, key, ip = load_fixture("c5_16attrs")
    issuer = afx.Context(params, key, ip, device=local_rank)
    issuer.set_secret_independent_addressing(0)   # input generation (reductions, Elligator maps of public attribute values)
    rng = np.random.default_rng(4242 + rank)
    rb = lambda *shape: rng.integers(0, 256, size=shape, dtype=np.uint8)
    dev = torch.device("cuda", local_rank)
    values = np.zeros((n, count, 32), np.uint8)
    kinds = []
    for i, c in enumerate(layout):
        kinds.append({"S": afx.ATTR_PUBLIC_SCALAR, "P": afx.ATTR_PUBLIC_POINT, "E": afx.ATTR_EITHER_POINT}[c])
        for o in range(0, count, 1 << 18):
            w = rb(min(1 << 18, count - o), 64)
            values[i, o:o + w.shape[0]] = batch.scalars_from_wide(issuer, w) if c == "S" else batch.points_from_uniform(issuer, w)
    d_in = {k: torch.from_numpy(v).to(dev) for k, v in dict(values=values, t_wide=rb(count, 64), U_wide=rb(count, 64), seed=rb(count, 32)).items()}
    d_out = {k: torch.zeros((count, 32), dtype=torch.uint8, device=dev) for k in ("t", "U", "V", "challenge")}
    d_out["responses"] = torch.zeros((n + 5, count, 32), dtype=torch.uint8, device=dev)
    status = torch.full((count,), 255, dtype=torch.uint8, device=dev)
    req = afx.AttributesSoA()
    req.n_attributes = n
    for i, k in enumerate(kinds):
        req.kinds[i] = k
    req.values = d_in["values"].data_ptr()
    rnd = afx.IssueRandomness(d_in["t_wide"].data_ptr(), d_in["U_wide"].data_ptr(), d_in["seed"].data_ptr())
    out = afx.IssuanceSoA(*(d_out[k].data_ptr() for k in ("t", "U", "V", "challenge", "responses")))
    fn = afx.lib().afx_issue_dev

    def step():
        afx.check(fn(issuer.h, C.byref(req), C.byref(rnd), count, C.byref(out), status.data_ptr()))

    def fence():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
    octx = None
    lines = []
    for mode in modes:
        issuer.set_secret_independent_addressing(mode)
        for k in d_out:
            d_out[k].zero_()
        status.fill_(255)
        for _ in range(warmup):
            step()
        torch.cuda.synchronize()
        assert UNCHECKED or not status.cpu().numpy().any(), "issue failed"
        # parity spot check of the first 64 credentials against the CPU oracle (checker only)
        if rank == 0 and not args.no_cpu_baseline and not UNCHECKED:
            import oracle
            octx = octx or oracle.Ctx(params, key, ip)
            h = {k: v[..., :64, :].cpu().numpy() for k, v in d_out.items()}
            hin = {k: v[..., :64, :].cpu().numpy() for k, v in d_in.items()}
            for i in range(64):
                vals = [bytes(hin["values"][k, i]) + bytes(64) for k in range(n)]
                st, t, U, V, ch, resp = octx.issue(kinds, vals, bytes(hin["t_wide"][i]), bytes(hin["U_wide"][i]), bytes(hin["seed"][i]))
                assert st == 0 and t == bytes(h["t"][i]) and U == bytes(h["U"][i]) and V == bytes(h["V"][i]) and ch == bytes(h["challenge"][i])
                assert all(resp[k] == bytes(h["responses"][k, i]) for k in range(n + 5)), "GPU issuance differs from the oracle"
        issuer.set_timing(True)
        fence()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        fence()
        elapsed = time.perf_counter() - t0
        kt = kernel_times(issuer, steps)
        valu = valu_side(issuer, count, sum(kt[k]["ms_per_step"] for k in kt if k in MSM_KERNELS + FIELD_KERNELS))
        secret_terms = issuer.plan_stats()["secret_terms"]
        issuer.set_timing(False)
        assert UNCHECKED or not status.cpu().numpy().any(), "issue failed in the timed steps"
        if dist is not None:
            elapsed = dist.max_over_ranks(elapsed)
        cpu = None
        if rank == 0 and world == 1 and not args.no_cpu_baseline and not embedded:   # the CPU baseline is an N=1 figure
            S = 256
            hin = {k: v[..., :S, :].cpu().numpy() for k, v in d_in.items()}
            t0 = time.perf_counter()
            for i in range(S):
                vals = [bytes(hin["values"][k, i]) + bytes(64) for k in range(n)]
                octx.issue(kinds, vals, bytes(hin["t_wide"][i]), bytes(hin["U_wide"][i]), bytes(hin["seed"][i]))
            cpu = {"value": S / (time.perf_counter() - t0), "unit": "credentials/s", "cores": 1, "kind": "port", "cpu_model": cpu_model(),
                   "sample": "first %d issuances of the same batch through the oracle (one thread, called from python)" % S}
        ab = 32 * n + n + 160 + 96 + 32 * (n + 6)
        lines.append({
            "metric": "credentials issued/sec (secondary; aMAC tag + issuance NIZK)", "value": count * world * steps / elapsed,
            "unit": "credentials/s", "n_gpus": world, "steps": steps, "warmup": warmup, "ms_per_step": elapsed / steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "int64", "data": "synthetic (random attribute values)",
            "config": {"workload": "C5: batch issue 2^20 credentials, 16 attributes (S x8, P x4, E x4)", "credentials_per_gpu": count,
                       "secret_independent_addressing": SECRET_MODE_NAMES[mode], "secret_terms_in_plan": secret_terms,
                       "algorithmic_bytes_per_credential": ab, "parallelism": "host-sharded x%d, no collective" % world, "rank_devices": RANK_DEVICES,
                       "dist_backend": dist.backend if dist is not None else None, "dist_backend_fallback": dist.fallback if dist is not None else None},
            "roofline": roofline_of(kt, "c5", ab, count, mode != 0),
            "valu": with_value_per_mhz(valu, count * world * steps / elapsed), "cpu_baseline": cpu})
    issuer.close()
    if embedded:
        return lines
    if rank == 0:
        print(json.dumps(lines[0]))
    if dist is not None:
        dist.destroy()


def bench_show(args, afx, batch, torch, dist, rank, world, local_rank, embedded=None):
    """secondary metric: presentations created / s (AnonymousCredential::show), C2 shape, device-resident inputs
    (embedded: as in bench_issue)"""
    n, layout, hide, count = 4, "SSPE", [0, 3], ((args.batch if not embedded else 0) or (1 << 16))
    steps, warmup = (embedded["steps"], embedded["warmup"]) if embedded else (args.steps, args.warmup)
    modes = embedded["modes"] if embedded else [secret_mode_of(args)]
    params, key, ip = load_fixture("readme_4attrs_sSPe")
    issuer = afx.Context(params, key, ip, device=local_rank)
    user = afx.Context(params, None, ip, device=local_rank)
    issuer.set_secret_independent_addressing(0)   # the issuer only makes the synthetic credentials here
    rng = np.random.default_rng(99 + rank)
    rb = lambda *shape: rng.integers(0, 256, size=shape, dtype=np.uint8)
    dev = torch.device("cuda", local_rank)
    values, M2, m3 = (np.zeros((n, count, 32), np.uint8) for _ in range(3))
    kinds = []
    for i, c in enumerate(layout):
        kinds.append({"S": afx.ATTR_PUBLIC_SCALAR, "P": afx.ATTR_PUBLIC_POINT, "E": afx.ATTR_EITHER_POINT}[c])
        values[i] = batch.scalars_from_wide(issuer, rb(count, 64)) if c == "S" else batch.points_from_uniform(issuer, rb(count, 64))
        if c == "E":
            M2[i] = batch.points_from_uniform(issuer, rb(count, 64))
            m3[i] = batch.scalars_from_wide(issuer, rb(count, 64))
    iss, st = batch.issue(issuer, kinds, values, rb(count, 64), rb(count, 64), rb(count, 32))
    assert not st.any()
    skinds = list(kinds)
    for i in hide:
        skinds[i] = afx.ATTR_SECRET_SCALAR if skinds[i] == afx.ATTR_PUBLIC_SCALAR else afx.ATTR_SECRET_POINT
    g = max(3, n)
    gen = lambda idx: np.frombuffer(params[4 + 32 * idx:4 + 32 * idx + 32], np.uint8)
    a, a0, a1 = (batch.scalars_from_wide(issuer, rb(count, 64)) for _ in range(3))
    pk, ok = batch.multiscalar_mul(issuer, np.stack([a, a0, a1]), np.stack([np.broadcast_to(gen(5 + g + n + 1 + k), (count, 32)) for k in range(3)]))
    host_in = dict(values=values, M2=M2, m3=m3, t=iss["t"], U=iss["U"], V=iss["V"], a=a, a0=a0, a1=a1, pk=pk,
                   z_wide=rb(count, 64), seed=rb(count, 32), enc_seeds=rb(1, count, 32))
    d = {k: torch.from_numpy(np.ascontiguousarray(v)).to(dev) for k, v in host_in.items()}
    o = {k: torch.zeros((count, 32), dtype=torch.uint8, device=dev) for k in ("challenge", "C_x_0", "C_x_1", "C_V")}
    o["responses"] = torch.zeros((4, count, 32), dtype=torch.uint8, device=dev)
    o["C_y"] = torch.zeros((n, count, 32), dtype=torch.uint8, device=dev)
    o["attr_values"] = torch.zeros((n, count, 32), dtype=torch.uint8, device=dev)
    eo = {f: torch.zeros(((6, count, 32) if f == "responses" else (count, 32)), dtype=torch.uint8, device=dev) for f in batch.ENC_FIELDS}
    cs = afx.CredentialsSoA()
    cs.n_attributes = n
    for i, k in enumerate(skinds):
        cs.kinds[i] = k
    for f in ("values", "M2", "m3", "t", "U", "V"):
        setattr(cs, f, d[f].data_ptr())
    kp = afx.KeypairsSoA(*(d[f].data_ptr() for f in ("a", "a0", "a1", "pk")))
    rnd = afx.ShowRandomness(d["z_wide"].data_ptr(), d["seed"].data_ptr(), d["enc_seeds"].data_ptr())
    eouts = (afx.EncProofOut * 1)()
    for f in batch.ENC_FIELDS:
        setattr(eouts[0], f, eo[f].data_ptr())
    out = afx.PresentationOut()
    for f in batch.PRES_FIELDS:
        setattr(out, f, o[f].data_ptr())
    out.enc = C.cast(eouts, C.POINTER(afx.EncProofOut))
    shape = afx.Shape()
    status = torch.full((count,), 255, dtype=torch.uint8, device=dev)
    fn = afx.lib().afx_show_dev

    def step():
        afx.check(fn(user.h, C.byref(cs), C.byref(kp), C.byref(rnd), count, C.byref(out), C.byref(shape), status.data_ptr()))

    def fence():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
    octx = None
    lines = []
    for mode in modes:
        user.set_secret_independent_addressing(mode)
        for t_ in list(o.values()) + list(eo.values()):
            t_.zero_()
        status.fill_(255)
        for _ in range(warmup):
            step()
        torch.cuda.synchronize()
        assert not status.cpu().numpy().any(), "show failed"
        # every presentation made here must verify on the issuer's side (GPU), and the first 16 must equal the oracle's bytes
        pres = {f: o[f].cpu().numpy() for f in batch.PRES_FIELDS}
        pres["enc"] = [{f: eo[f].cpu().numpy() for f in batch.ENC_FIELDS}]
        assert not batch.verify_presentations(issuer, shape, pres).any(), "a shown presentation does not verify"
        if rank == 0 and not args.no_cpu_baseline:
            import oracle
            octx = octx or oracle.Ctx(params, None, ip)
            for i in range(16):
                vals = [bytes(values[k, i]) + bytes(M2[k, i]) + bytes(m3[k, i]) for k in range(n)]
                kpb = bytes(a[i]) + bytes(a0[i]) + bytes(a1[i]) + bytes(pk[i])
                st, p = octx.show(skinds, vals, bytes(iss["t"][i]), bytes(iss["U"][i]), bytes(iss["V"][i]), kpb, bytes(host_in["z_wide"][i]),
                                  bytes(host_in["seed"][i]), bytes(host_in["enc_seeds"][0, i]))
                assert st == 0 and bytes(p.challenge) == bytes(pres["challenge"][i]) and bytes(p.enc[0].challenge) == bytes(pres["enc"][0]["challenge"][i])
                assert all(bytes(p.responses[k]) == bytes(pres["responses"][k, i]) for k in range(4)), "GPU show differs from the oracle"
        user.set_timing(True)
        fence()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        fence()
        elapsed = time.perf_counter() - t0
        kt = kernel_times(user, steps)
        valu = valu_side(user, count, sum(kt[k]["ms_per_step"] for k in kt if k in MSM_KERNELS + FIELD_KERNELS))
        secret_terms = user.plan_stats()["secret_terms"]
        user.set_timing(False)
        if dist is not None:
            elapsed = dist.max_over_ranks(elapsed)
        ab = 32 * n + n + 96 + 32 + 128 + 96 + 64 + 32 + 32 + 907   # credential + keypair + randomness read, presentation written
        lines.append({
            "metric": "credential presentations created/sec (secondary; AnonymousCredential::show)", "value": count * world * steps / elapsed,
            "unit": "presentations/s", "n_gpus": world, "steps": steps, "warmup": warmup, "ms_per_step": elapsed / steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "int64", "data": "synthetic (GPU-issued credentials)",
            "config": {"workload": "show 2^16 credentials, 4 attributes (s S P e)", "credentials_per_gpu": count, "algorithmic_bytes_per_presentation": ab,
                       "secret_independent_addressing": SECRET_MODE_NAMES[mode], "secret_terms_in_plan": secret_terms,
                       "parallelism": "host-sharded x%d, no collective" % world, "rank_devices": RANK_DEVICES},
            "roofline": roofline_of(kt, "show", ab, count, mode != 0),
            "valu": with_value_per_mhz(valu, count * world * steps / elapsed), "cpu_baseline": None})
    issuer.close()
    user.close()
    if embedded:
        return lines
    if rank == 0:
        print(json.dumps(lines[0]))
    if dist is not None:
        dist.destroy()


# v_mad_i64_i32 issue cost at the occupancy the kernels run at (2 waves per SIMD), measured with a pure multiply-add loop
# compiled from plain C (tools/ubench/mad_sustained.hip, profiles/r02_mad_sustained.txt): <= 4.68 cycles per wave-instruction;
# that loop sustains 32.2 T mads/s at the ~2.29 GHz the power cap allows it.  (Round 1 priced the instruction at 5.54 cycles
# from an inline-asm loop; that overstated every "valu" fraction by 18 %.)  No published figure exists for this opcode.
MAD_CYCLES = 4.68
MAD_PEAK_T = 64 / MAD_CYCLES * 1024 * 2.4e9 / 1e12      # at the nominal 2.4 GHz
MAD_SUSTAINED_T = 32.2                                  # what a pure multiply-add stream sustains at the power cap


NOMINAL_MHZ = 2400.0
MADS_PER_MUL, MADS_PER_SQ = 98, 62
MADS_PER_CHAIN_MUL, MADS_PER_CHAIN_SQ = 100, 55   # inside the inversion / square-root chains: the 10 x 25.5-bit form (fe10.cuh)


def valu_side(ctx, items_per_step, field_kernel_ms_per_step):
    """compute-side figure beside the HBM roofline: 32x32->64-bit multiply-adds of the field arithmetic per second.
    Counts come from the engine's own plan of the last call (afx_ctx_get_plan_stats); a field multiplication issues 98
    multiply-adds (81 products, 16 to fold the high columns, 1 for the wrap), a squaring 62 (45 + 16 + 1): 9 x 29-bit limbs,
    aeonflux_amd/csrc/fe.cuh; inside the inversion / square-root chains (chain_mul, chain_sq of the stats) 100 and 55: fe10.cuh."""
    st = ctx.plan_stats()
    mads = (MADS_PER_MUL * (st["field_mul"] - st["chain_mul"]) + MADS_PER_SQ * (st["field_sq"] - st["chain_sq"])
            + MADS_PER_CHAIN_MUL * st["chain_mul"] + MADS_PER_CHAIN_SQ * st["chain_sq"])
    achieved = mads * items_per_step / (field_kernel_ms_per_step / 1e3) / 1e12 if field_kernel_ms_per_step > 0 else 0.0
    # measured inside the timed k_msm_window launches (shader-clock counter / 100 MHz counter) by 64 blocks spread over each launch's
    # duration and over the eight XCDs: the median (a launch starts at the boost clock and settles at what the power cap leaves)
    mhz = ctx.core_clock_mhz()
    try:
        samples = ctx.core_clock_samples()
    except Exception:   # an older build of the library (same-box A/Bs)
        samples = []
    at_clock = MAD_PEAK_T * mhz / NOMINAL_MHZ if mhz > 0 else None
    # the operation-count floor: what the statement's schedule costs if every multiply-add issued back to back - doublings shared per
    # chain (252 per variable-base job), 64 additions per 4-bit-window term, one square root per decoded point, one inversion per
    # encoding row: the counts in "per_item", which follow from the statement's term counts alone (Straus, 4-bit signed windows)
    floor = None
    if mads and at_clock:
        floor = {"mads_per_item": mads, "items_per_s_at_nominal_clock": MAD_PEAK_T * 1e12 / mads, "items_per_s_at_measured_clock": at_clock * 1e12 / mads,
                 "field_kernels_items_per_s": (items_per_step / (field_kernel_ms_per_step / 1e3)) if field_kernel_ms_per_step > 0 else None,
                 "note": "items/s the field arithmetic of this statement would reach with the multiply-add port busy every cycle (peak = %.2f cycles per "
                         "wave-instruction, measured); field_kernels_items_per_s / items_per_s_at_measured_clock = frac_at_measured_clock" % MAD_CYCLES}
    return {"unit": "T multiply-adds/s (v_mad_i64_i32 / v_mad_u64_u32)", "achieved": achieved, "peak": MAD_PEAK_T, "frac": achieved / MAD_PEAK_T,
            "core_clock_mhz_measured": mhz, "peak_at_measured_clock": at_clock, "frac_at_measured_clock": (achieved / at_clock) if at_clock else None,
            "clock_samples": [round(x, 1) for x in samples],
            "clock_spread_mhz": {"min": round(samples[0], 1), "p25": round(samples[len(samples) // 4], 1), "median": round(mhz, 1),
                                 "p75": round(samples[(3 * len(samples)) // 4], 1), "max": round(samples[-1], 1), "blocks": len(samples)} if samples else None,
            "operation_count_floor": floor,
            # the boxes of the pool run this path at 1.85-1.98 GHz (socket power cap): items per second and measured MHz of THIS
            # process's field kernels is the figure that compares across boxes and rounds
            "items_per_s_per_mhz_field_kernels": (items_per_step / (field_kernel_ms_per_step / 1e3) / mhz) if (mhz > 0 and field_kernel_ms_per_step > 0) else None,
            "clock_note": "peak is at the nominal 2400 MHz; the kernels run at the socket power cap, below it",
            "peak_source": "tools/ubench/mad_sustained.hip on this GPU (no published figure): %.2f cycles per wave-instruction" % MAD_CYCLES,
            "peak_sustained_pure_mad_loop": MAD_SUSTAINED_T, "frac_of_sustained": achieved / MAD_SUSTAINED_T, "per_item": dict(st, mads=mads),
            "time_base": "summed durations of the kernels doing field arithmetic (k_msm_*, k_compress2x, k_negenc, k_pointsum, k_decode, k_pointop, k_from_uniform)"}


def secret_mode_of(args):
    """the afx_ctx_set_secret_independent_addressing mode of the measured context"""
    if args.secret_mode is not None:
        return args.secret_mode
    return 1 if args.secret_independent else 2


def with_value_per_mhz(valu, value):
    """the line's `value` (whole-job items per second) per measured MHz of the core clock: what compares across the pool's boxes"""
    mhz = valu.get("core_clock_mhz_measured") or 0
    out = dict(valu, value_per_mhz=(value / mhz) if mhz > 0 else None)
    fl = valu.get("operation_count_floor")
    if fl:   # the whole step (transcripts, table building, staging included) against the field arithmetic's floor
        out["operation_count_floor"] = dict(fl, value_over_floor_at_measured_clock=value / fl["items_per_s_at_measured_clock"])
    return out


RANK_DEVICES = None
SECRET_MODE_NAMES = {0: "nowhere (mode 0)", 1: "everywhere (mode 1)", 2: "prover-side calls (mode 2, the default)"}


PG_FAILED_EXIT = 75   # a rank whose process group did not form exits with this code (EX_TEMPFAIL): launch_ranks retries once


class Ranks:
    """The measurement's process group: a barrier on both sides of the timed steps and one MAX all-reduce of a double (the data path
    has no collective).  Every rank first joins a gloo group (TCP on 127.0.0.1: it forms wherever processes can talk at all); RCCL -
    what the contract asks for - is then tried as a SECOND group, with one test all-reduce, and the ranks AGREE over gloo whether it
    formed everywhere: all of them use it, or all of them stay on gloo (`fallback` says why).  Ranks deciding one by one would end up
    on different backends and hang in the first barrier; a rank that cannot even join the gloo group exits with PG_FAILED_EXIT."""

    def __init__(self, torch, rank, world, local_rank, want):
        import datetime
        import torch.distributed as dist
        self.torch, self.dist, self.world, self.dev = torch, dist, world, torch.device("cuda", local_rank)
        self.group, self.backend, self.fallback = None, "gloo", None
        # (gloo and RCCL announce themselves with printf: while the groups form, file descriptor 1 is the process's stderr, so that
        # rank 0's stdout stays the ONE JSON line of the contract)
        sys.stdout.flush()
        saved = os.dup(1)
        os.dup2(2, 1)
        try:
            self._form(torch, dist, rank, world, want)
        finally:
            sys.stdout.flush()
            try:
                C.CDLL(None).fflush(None)   # (the C library's own stdout buffer: RCCL's banner would otherwise surface at exit)
            except OSError:
                pass
            os.dup2(saved, 1)
            os.close(saved)

    def _form(self, torch, dist, rank, world, want):
        import datetime
        try:
            dist.init_process_group(backend="gloo", timeout=datetime.timedelta(minutes=10))
        except Exception as e:   # noqa: BLE001
            sys.stderr.write("bench.py: rank %d: the gloo process group did not form (%s)\n" % (rank, e))
            sys.exit(PG_FAILED_EXIT)
        if want.split()[0] != "nccl":
            if want != "gloo":
                self.fallback = want
            return
        ok, why = 1, ""
        if not dist.is_nccl_available():
            ok, why = 0, "this torch build has no RCCL"
        else:
            os.environ.setdefault("TORCH_NCCL_BLOCKING_WAIT", "1")   # a collective that times out raises here instead of ending the process
            try:
                g = dist.new_group(backend="nccl", timeout=datetime.timedelta(minutes=3))
                t = torch.ones(1, device=self.dev)
                dist.all_reduce(t, group=g)
                torch.cuda.synchronize()
                if int(t.item()) != world:
                    ok, why = 0, "test all-reduce returned %s" % t.item()
            except Exception as e:   # noqa: BLE001
                ok, why, g = 0, "%s: %s" % (type(e).__name__, str(e).splitlines()[0][:200] if str(e) else ""), None
        flag = torch.tensor([ok], dtype=torch.int32)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)   # (gloo) everybody, or nobody
        if int(flag.item()) == 1:
            self.group, self.backend = g, "nccl"
        else:
            whys = [None] * world
            dist.all_gather_object(whys, why)
            self.fallback = "the RCCL group did not form on every rank (%s): the barrier and the MAX all-reduce ran on gloo" % \
                "; ".join("rank %d: %s" % (r, w) for r, w in enumerate(whys) if w)

    def barrier(self):
        if self.group is not None:
            self.dist.barrier(group=self.group, device_ids=[self.dev.index])
        else:
            self.dist.barrier()

    def max_over_ranks(self, seconds):
        t = self.torch.tensor([seconds], dtype=self.torch.float64, device=self.dev if self.group is not None else "cpu")
        if self.group is not None:
            self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX, group=self.group)
        else:
            self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def all_gather_object(self, obj):
        out = [None] * self.world
        self.dist.all_gather_object(out, obj)
        return out

    def destroy(self):
        self.dist.barrier()
        self.dist.destroy_process_group()


def gather_rank_devices(torch, dist, world, local_rank):
    """["0000:05:00.0", ...]: the PCI bus id of the device each rank runs on, in rank order (hipDeviceGetPCIBusId, via torch's properties)"""
    p = torch.cuda.get_device_properties(local_rank)
    try:
        mine = "%04x:%02x:%02x.0" % (p.pci_domain_id, p.pci_bus_id, p.pci_device_id)
    except AttributeError:   # a torch build without the PCI fields: the device's uuid, or its index as a last resort
        mine = str(getattr(p, "uuid", "device-%d" % local_rank))
    if dist is None or world == 1:
        return [mine]
    return dist.all_gather_object(mine)


def free_port():
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_once(n, argv, worker, timeout_s):
    """one launch of N ranks; returns (rc, rank 0's stdout bytes, the first failing rank's exit code or None)"""
    import subprocess
    import threading
    port = free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), AFX_BENCH_LAUNCHED="1")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen(worker + list(argv), env=env, stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    t0 = time.time()
    rc, first = 0, None
    out0 = [b""]

    def drain():
        out0[0] = procs[0].stdout.read()
    th = threading.Thread(target=drain, daemon=True)
    th.start()
    live = set(range(n))
    while live:
        for r in sorted(live):
            code = procs[r].poll()
            if code is None:
                continue
            live.discard(r)
            if code != 0 and rc == 0:
                rc = code if code > 0 else 1
                first = code
                sys.stderr.write("bench.py: rank %d exited with code %d; stopping the other ranks\n" % (r, code))
                for o in live:
                    procs[o].terminate()   # exactly the processes started above
        if timeout_s is not None and time.time() - t0 > timeout_s and live:
            sys.stderr.write("bench.py: ranks %s still running after %.0f s; stopping them\n" % (sorted(live), timeout_s))
            for o in live:
                procs[o].kill()
            rc = rc or 124
            timeout_s = None
        if live:
            time.sleep(0.05)
    th.join(timeout=10)
    return rc, out0[0], first


def launch_ranks(n, argv, worker=None, timeout_s=None):
    """`python bench.py --gpus N` without a launcher around it: start N fresh child processes, one rank per GPU, with the
    environment torch.distributed.run would give them (RANK, LOCAL_RANK, WORLD_SIZE, MASTER_ADDR/PORT), relay rank 0's
    output line and return non-zero if any rank failed.  The parent never touches the GPU (children are new processes, not
    an exec of one that initialised HIP).  `worker` is the child command (tests substitute a stub).
    A launch whose first failing rank says its process group did not form (PG_FAILED_EXIT: a rendezvous port taken between
    free_port() and the ranks' bind, a backend that cannot start on this node) is repeated ONCE with fresh children, a new port
    and `--dist-backend gloo`; the second launch's line records why (config.dist_backend_fallback)."""
    worker = worker or [sys.executable, os.path.abspath(__file__)]
    rc, out0, first = launch_once(n, argv, worker, timeout_s)
    if first == PG_FAILED_EXIT and "--dist-backend-fallback" not in argv:
        sys.stderr.write("bench.py: a rank's process group did not form; launching the ranks once more on gloo\n")
        retry = [a for a in argv]
        if "--dist-backend" in retry:
            i = retry.index("--dist-backend")
            del retry[i:i + 2]
        retry = [a for a in retry if not a.startswith("--dist-backend=")]
        retry += ["--dist-backend", "gloo", "--dist-backend-fallback", "launcher: the first launch's process group did not form (a rank exited with code %d)" % PG_FAILED_EXIT]
        rc, out0, first = launch_once(n, retry, worker, timeout_s)
    sys.stdout.write(out0.decode(errors="replace"))
    sys.stdout.flush()
    return rc


def preflight(args):
    """can `--gpus N` run here?  One JSON line; exit code 0 when every check passes.  Touches the GPUs (properties, free memory):
    run it as its own process, never before a launch in the same one."""
    n = args.gpus
    out = {"preflight": True, "gpus_asked": n, "checks": {}}
    ok = True

    def check(name, passed, detail):
        nonlocal ok
        out["checks"][name] = {"ok": bool(passed), "detail": detail}
        ok = ok and bool(passed)
    try:
        import torch
        import torch.distributed as dist
        have = torch.cuda.device_count()
        check("devices", have >= n, "%d visible, %d asked for" % (have, n))
        ids, free = [], []
        for d in range(min(have, n)):
            p = torch.cuda.get_device_properties(d)
            try:
                ids.append("%04x:%02x:%02x.0" % (p.pci_domain_id, p.pci_bus_id, p.pci_device_id))
            except AttributeError:
                ids.append(str(getattr(p, "uuid", d)))
            free.append(torch.cuda.mem_get_info(d)[0])
        check("distinct_devices", len(set(ids)) == len(ids), ids)
        # a pass of 2^19 C3 presentations: ~70 KB of workspace per item + the batch resident (2425 B per presentation)
        workload = args.workload or ("c3" if n == 1 else "c4")
        per_gpu = args.batch or (WORKLOADS[workload][3] // (n if workload == "c4" else 1) if workload in WORKLOADS else 1 << 20)
        need = min(per_gpu, 1 << 19) * 70 * 1024 + per_gpu * 2425 * 2
        check("free_hbm", all(f >= need for f in free) and bool(free), {"need_bytes_per_gpu": need, "free_bytes": free})
        check("rccl", dist.is_nccl_available(), "torch.distributed.is_nccl_available(); without it the barrier and the MAX all-reduce run on gloo")
        try:
            avail = int(next(l.split()[1] for l in open("/proc/meminfo") if l.startswith("MemAvailable"))) * 1024
        except (OSError, StopIteration, ValueError):
            avail = None
        host_need = n * per_gpu * 2425 * 4   # every rank holds its batch in numpy arrays, their concatenation parts, and the corrupted copy
        check("host_memory", avail is None or avail >= host_need, {"need_bytes": host_need, "available_bytes": avail})
        check("usable_cores", usable_cores() >= n, "%d usable for %d ranks" % (usable_cores(), n))
        try:
            import aeonflux_amd as afx
            afx.lib()
            check("library", True, os.path.relpath(afx.LIB_PATH, ROOT))
        except Exception as e:   # noqa: BLE001
            check("library", False, "%s: %s" % (type(e).__name__, e))
    except Exception as e:   # noqa: BLE001
        check("torch", False, "%s: %s" % (type(e).__name__, e))
    out["ok"] = ok
    print(json.dumps(out))
    return 0 if ok else 1


def tile_items(a, reps):
    return a if reps == 1 else np.ascontiguousarray(np.concatenate([a] * reps, axis=-2))


def group_api_rate(afx, batch, params, key, ip, shape, pres, want, devices, copies, reps=1):
    """What the single `&self` call Issuer::verify (/root/reference/src/issuer.rs:141-147) becomes on a node: ONE process, one
    afx_group over `devices`, afx_group_verify_presentations on host arrays (contiguous split, one host thread and two
    streams per member, no collective).  The batch is `reps` copies of this rank's (PCIe and staging inclusive, never
    `value`).  Returns (spread of presentations/s over `reps` repetitions, items)."""
    g = afx.Group(params, key, ip, devices)
    try:
        big = {f: tile_items(pres[f], copies) for f in batch.PRES_FIELDS}
        big["enc"] = [{f: tile_items(d[f], copies) for f in batch.ENC_FIELDS} for d in pres["enc"]]
        total = big["challenge"].shape[0]
        exp = np.concatenate([want] * copies)
        soa, keep = batch.presentation_soa(big)
        st = np.full(total, 255, np.uint8)
        fn = afx.lib().afx_group_verify_presentations
        rates = timed_reps(lambda: afx.check(fn(g.h, C.byref(shape), C.byref(soa), total, st.ctypes.data)), total, max(1, reps))
        assert UNCHECKED or np.array_equal(st, exp), "group API statuses differ from the expected ones"
        return spread(rates), total
    finally:
        g.close()


def spread(rates):
    """{"median", "min", "max", "reps"} of a leg's repetitions (items per second each)"""
    import statistics
    return {"median": statistics.median(rates), "min": min(rates), "max": max(rates), "reps": len(rates)}


def timed_reps(call, items, reps):
    """`reps` timed repetitions of a synchronous host call over `items` items, after one untimed warm-up (buffers, plans)"""
    call()
    rates = []
    for _ in range(reps):
        t0 = time.perf_counter()
        call()
        rates.append(items / (time.perf_counter() - t0))
    return rates


def verify_secondary(afx, batch, torch, local_rank, workload, steps, warmup, check_oracle):
    """one more verification workload of BASELINE.json for the line's config.secondary (N = 1): generated, 1 % corrupted, resident in
    HBM, every status byte checked, timed like the headline; with `check_oracle` the first 256 statuses and recomputed challenges are
    compared with the oracle's too"""
    n, layout, hide, count, fixture, desc = WORKLOADS[workload]
    params, key, ip = load_fixture(fixture)
    issuer = afx.Context(params, key, ip, device=local_rank)
    user = afx.Context(params, None, ip, device=local_rank)
    pres, shape = generate(afx, batch, issuer, user, params, n, layout, hide, count, 555, fast_tables=True)
    want = corrupt(pres, count, 11)
    user.close()
    issuer.set_secret_independent_addressing(2)
    dev = torch.device("cuda", local_rank)
    dpres = {f: torch.from_numpy(pres[f]).to(dev) for f in batch.PRES_FIELDS}
    dpres["enc"] = [{f: torch.from_numpy(d[f]).to(dev) for f in batch.ENC_FIELDS} for d in pres["enc"]]
    soa, keep = batch.presentation_soa(dpres, ptr=lambda t: t.data_ptr())
    status = torch.full((count,), 255, dtype=torch.uint8, device=dev)
    fn = afx.lib().afx_verify_presentations_dev
    for _ in range(warmup):
        afx.check(fn(issuer.h, C.byref(shape), C.byref(soa), count, status.data_ptr()))
    torch.cuda.synchronize()
    got = status.cpu().numpy()
    assert UNCHECKED or np.array_equal(got, want), "%s: status mismatch" % workload
    if check_oracle and not UNCHECKED:
        import oracle   # checker only
        S = min(256, count)
        sub = {f: np.ascontiguousarray(pres[f][..., :S, :]) for f in batch.PRES_FIELDS}
        sub["enc"] = [{f: np.ascontiguousarray(d[f][..., :S, :]) for f in batch.ENC_FIELDS} for d in pres["enc"]]
        osoa, keep2 = batch.presentation_soa(sub)
        octx = oracle.Ctx(params, key, ip)
        ost = np.full(S, 255, np.uint8)
        oracle.lib().afxo_verify_presentations_soa(octx.h, C.byref(oracle.Shape.from_buffer_copy(bytes(shape))),
                                                   C.byref(oracle.PresentationSoA.from_buffer_copy(bytes(osoa))), S, ost.ctypes.data, 4)
        assert np.array_equal(ost, got[:S]), "%s: GPU and oracle statuses differ" % workload
    issuer.set_timing(True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        afx.check(fn(issuer.h, C.byref(shape), C.byref(soa), count, status.data_ptr()))
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    kt = kernel_times(issuer, steps)
    valu = valu_side(issuer, count, sum(kt[k]["ms_per_step"] for k in kt if k in MSM_KERNELS + FIELD_KERNELS))
    issuer.set_timing(False)
    assert UNCHECKED or np.array_equal(status.cpu().numpy(), want), "%s: status mismatch after the timed steps" % workload
    ab = algorithmic_bytes(shape)
    line = {"metric": "credential presentations verified/sec", "value": count * steps / elapsed, "unit": "presentations/s", "steps": steps, "warmup": warmup,
            "ms_per_step": elapsed / steps * 1e3, "config": {"workload": desc, "presentations_per_gpu": count, "algorithmic_bytes_per_presentation": ab},
            "roofline": roofline_of(kt, workload, ab, count, False), "valu": with_value_per_mhz(valu, count * steps / elapsed)}
    issuer.close()
    return line


def brief(line):
    """what config.secondary keeps of a secondary workload's full line"""
    r, v = line["roofline"], line["valu"]
    return {"workload": line["config"]["workload"], "metric": line["metric"], "value": line["value"], "unit": line["unit"], "ms_per_step": line["ms_per_step"],
            "steps": line["steps"], "warmup": line["warmup"],
            "secret_independent_addressing": line["config"].get("secret_independent_addressing"), "secret_terms_in_plan": line["config"].get("secret_terms_in_plan"),
            "algorithmic_bytes_per_item": line["config"].get("algorithmic_bytes_per_presentation", line["config"].get("algorithmic_bytes_per_credential")),
            "roofline": {k: r[k] for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "avg_launch_ms", "items_per_launch", "traffic", "traffic_stale")},
            "kernels_ms_per_step": r["kernels_ms_per_step"],
            "valu_frac_at_measured_clock": v.get("frac_at_measured_clock"), "core_clock_mhz_measured": v.get("core_clock_mhz_measured"),
            "value_per_mhz": v.get("value_per_mhz"), "mads_per_item": (v.get("per_item") or {}).get("mads")}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default=None, choices=sorted(WORKLOADS) + ["c5", "show"],
                    help="default: c3 on one GPU (the largest single-GPU configuration of BASELINE.json), c4 (2^22 presentations split "
                         "over the ranks) on several")
    ap.add_argument("--batch", type=int, default=0, help="presentations per GPU (default: the workload's)")
    ap.add_argument("--cpu-sample", type=int, default=16384)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-group-api", action="store_true", help="skip the afx_group_verify_presentations leg (rank 0, after the timed steps)")
    ap.add_argument("--no-secondary", action="store_true", help="skip config.secondary (N = 1, C3 only: C2, C5 in modes 2 and 0, show, a few steps each "
                    "after the headline's timed region)")
    ap.add_argument("--host-reps", type=int, default=5, help="repetitions of each host-side leg (host-pointer, serialized, group API): median / min / max")
    ap.add_argument("--pipelining", action="store_true", help="alternate steps between the engine's two streams (measured slower: the "
                    "path is VALU-bound, overlap only adds contention; default off)")
    ap.add_argument("--secret-mode", type=int, choices=(0, 1, 2), default=None,
                    help="afx_ctx_set_secret_independent_addressing of the measured context: 2 = the library's default (secrets never pick a table "
                         "address on the prover-side calls: issue, show), 1 = also the issuer key's terms of Issuer::verify, 0 = nowhere (the figures of "
                         "rounds 1-3).  Default: the library's default; the C3 / C4 verification lines are the same in modes 0 and 2")
    ap.add_argument("--secret-independent", action="store_true", help="= --secret-mode 1: afx_ctx_set_secret_independent_addressing everywhere (cost "
                    "measurement; results are the same bytes)")
    ap.add_argument("--dist-backend", default="nccl", help="nccl (RCCL; the driver's launch) or gloo (testing two ranks on one GPU)")
    ap.add_argument("--dist-backend-fallback", default=None, help=argparse.SUPPRESS)   # set by launch_ranks on its second launch: why the first one failed
    ap.add_argument("--preflight", action="store_true", help="check that --gpus N can run here (devices and their PCI ids, RCCL, free HBM, host "
                    "memory, the library), print one JSON line and exit 0 / 1 without running anything")
    args = ap.parse_args()

    if args.preflight:
        sys.exit(preflight(args))
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # no launcher around this process: be the launcher (before anything here touches the GPU)
        # a rank that hangs (a process group that half-formed) must not hang the launcher: an hour covers generation + parity
        # checks + the timed steps of the largest workload many times over
        sys.exit(launch_ranks(args.gpus, sys.argv[1:], timeout_s=float(os.environ.get("AFX_BENCH_LAUNCH_TIMEOUT_S", "3600"))))
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        sys.stderr.write("bench.py: --gpus %d but the launcher started %d ranks; reporting what runs\n" % (args.gpus, world))
    local_rank = int(os.environ.get("AFX_BENCH_DEVICE", os.environ.get("LOCAL_RANK", "0")))
    if args.workload is None:
        args.workload = "c3" if world == 1 else "c4"
    import torch
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the engine has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1 or (os.environ.get("AFX_BENCH_PG_ALWAYS") == "1" and "RANK" in os.environ):
        # (AFX_BENCH_PG_ALWAYS: a one-rank launch under torch.distributed.run still forms the group - how the RCCL leg is
        # exercised on a one-GPU box)
        dist = Ranks(torch, rank, world, local_rank, args.dist_backend)
    # which physical device every rank drives (PCI bus id): two ranks on one GPU would report a curve that is not one - refused,
    # unless AFX_BENCH_DEVICE put them there on purpose (the launch-path check on a one-GPU box)
    global RANK_DEVICES
    RANK_DEVICES = gather_rank_devices(torch, dist, world, local_rank)
    if len(set(RANK_DEVICES)) != len(RANK_DEVICES) and "AFX_BENCH_DEVICE" not in os.environ:
        if dist is not None:
            dist.destroy()
        raise SystemExit("bench.py: ranks share a device: %s (set AFX_BENCH_DEVICE to run several ranks on one GPU on purpose)" % RANK_DEVICES)
    import aeonflux_amd as afx
    from aeonflux_amd import batch

    if args.workload == "c5":
        return bench_issue(args, afx, batch, torch, dist, rank, world, local_rank)
    if args.workload == "show":
        return bench_show(args, afx, batch, torch, dist, rank, world, local_rank)
    n, layout, hide, count, fixture, desc = WORKLOADS[args.workload]
    if args.workload == "c4":
        count //= world
    if args.batch:
        count = args.batch
    params, key, ip = load_fixture(fixture)
    issuer = afx.Context(params, key, ip, device=local_rank)
    user = afx.Context(params, None, ip, device=local_rank)
    t0 = time.time()
    gen_chunk = 1 << 16
    parts = [generate(afx, batch, issuer, user, params, n, layout, hide, min(gen_chunk, count - o), 1000 + 97 * rank + o, fast_tables=True)
             for o in range(0, count, gen_chunk)]
    shape = parts[0][1]
    pres = {f: np.concatenate([p[0][f] for p in parts], axis=-2) for f in batch.PRES_FIELDS}
    pres["enc"] = [{f: np.concatenate([p[0]["enc"][e][f] for p in parts], axis=-2) for f in batch.ENC_FIELDS}
                   for e in range(shape.n_enc_proofs)]
    del parts
    want = corrupt(pres, count, 7 + rank)
    user.close()
    gen_s = time.time() - t0
    issuer.set_secret_independent_addressing(secret_mode_of(args))   # after generation: only the timed verification pays

    # inputs resident in HBM before the timed region
    dev = torch.device("cuda", local_rank)
    dpres = {f: torch.from_numpy(pres[f]).to(dev) for f in batch.PRES_FIELDS}
    dpres["enc"] = [{f: torch.from_numpy(d[f]).to(dev) for f in batch.ENC_FIELDS} for d in pres["enc"]]
    soa, keep = batch.presentation_soa(dpres, ptr=lambda t: t.data_ptr())
    status = torch.full((count,), 255, dtype=torch.uint8, device=dev)
    fn = afx.lib().afx_verify_presentations_dev

    def step():
        afx.check(fn(issuer.h, C.byref(shape), C.byref(soa), count, status.data_ptr()))

    def fence():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    got = status.cpu().numpy()
    if not np.array_equal(got, want) and not UNCHECKED:
        bad = np.nonzero(got != want)[0]
        raise SystemExit("status mismatch at %d items, first %s" % (bad.size, bad[:8]))
    issuer.set_pipelining(args.pipelining)
    issuer.set_timing(True)
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0
    kt = kernel_times(issuer, args.steps)
    valu = valu_side(issuer, count, sum(kt[k]["ms_per_step"] for k in kt if k in MSM_KERNELS + FIELD_KERNELS))
    issuer.set_timing(False)
    issuer.set_pipelining(False)
    if dist is not None:
        elapsed = dist.max_over_ranks(elapsed)
    got = status.cpu().numpy()
    assert UNCHECKED or np.array_equal(got, want), "status mismatch after the timed steps"
    ranks_seen, had_group = 1, dist is not None
    pg_backend = pg_fallback = None
    if dist is not None:
        # the measurement is over: every rank leaves the process group here; rank 0 goes on alone with the host-side legs
        ranks_seen, pg_backend, pg_fallback = dist.world, dist.backend, dist.fallback
        dist.destroy()
        dist = None
        if rank != 0:
            del dpres, soa, status
            issuer.close()
            return
    # PCIe-inclusive rate through the host-pointer entry point (never `value`; DESIGN.md quotes it): `--host-reps` repetitions
    # after a warm-up call, reported as median / min / max (one repetition cannot tell a slow box from a hiccup)
    pcie = wire_rate = None
    group_rate = group_items = group_err = None
    reps = max(1, args.host_reps)
    if rank == 0:
        hsoa, keep_h = batch.presentation_soa(pres)
        hst = np.full(count, 255, np.uint8)
        pcie = spread(timed_reps(lambda: afx.check(afx.lib().afx_verify_presentations(issuer.h, C.byref(shape), C.byref(hsoa), count, hst.ctypes.data)), count, reps))
        assert UNCHECKED or np.array_equal(hst, want)
        # ... and as one serialized AFXP blob in pageable host memory (afx_verify_presentations_wire: the transposition to
        # columns happens on the GPU, slice by slice); a quarter of the batch keeps the host-side packing short
        from aeonflux_amd import wire as wire_mod
        wn = min(count, 1 << 18)
        wsub = {f: np.ascontiguousarray(pres[f][..., :wn, :]) for f in batch.PRES_FIELDS}
        wsub["enc"] = [{f: np.ascontiguousarray(d[f][..., :wn, :]) for f in batch.ENC_FIELDS} for d in pres["enc"]]
        blob = wire_mod.pack_presentations(shape, wsub)
        wst, wcnt = np.full(wn, 255, np.uint8), C.c_size_t(0)
        wire_rate = spread(timed_reps(lambda: afx.check(afx.lib().afx_verify_presentations_wire(issuer.h, blob, len(blob), wst.ctypes.data, wn, C.byref(wcnt))), wn, reps))
        assert UNCHECKED or np.array_equal(wst, want[:wn])
        del blob, wsub
        # ... and what ONE small synchronous host-pointer call costs (the latency plan: DESIGN.md section 3): the first 64
        # presentations of the batch, mean over 20 calls after two warm-up calls (the first assembles and keeps the plan)
        sn = min(count, 64)
        ssub = {f: np.ascontiguousarray(pres[f][..., :sn, :]) for f in batch.PRES_FIELDS}
        ssub["enc"] = [{f: np.ascontiguousarray(d[f][..., :sn, :]) for f in batch.ENC_FIELDS} for d in pres["enc"]]
        ssoa, keep_s = batch.presentation_soa(ssub)
        sst = np.full(sn, 255, np.uint8)
        for _ in range(2):
            afx.check(afx.lib().afx_verify_presentations(issuer.h, C.byref(shape), C.byref(ssoa), sn, sst.ctypes.data))
        t0 = time.perf_counter()
        for _ in range(20):
            afx.check(afx.lib().afx_verify_presentations(issuer.h, C.byref(shape), C.byref(ssoa), sn, sst.ctypes.data))
        small_call_ms = (time.perf_counter() - t0) / 20 * 1e3
        assert UNCHECKED or np.array_equal(sst, want[:sn])
        # ... and through ONE afx_group over the node's GPUs (one process, host arrays): the in-library split
        del dpres, soa, status
        torch.cuda.empty_cache()
        if not args.no_group_api:
            devices = [local_rank] * world if "AFX_BENCH_DEVICE" in os.environ else list(range(world))
            try:
                group_rate, group_items = group_api_rate(afx, batch, params, key, ip, shape, pres, want, devices, world, reps)
            except Exception as e:   # a missing device must not cost the headline line
                group_err = "%s: %s" % (type(e).__name__, e)

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:   # the CPU baseline is an N=1 figure
        import oracle   # ORACLE: only here, as the timed CPU baseline and the checker
        olib = oracle.load(native=True)
        olib.afxo_ctx_new.restype = C.c_void_p
        S = min(args.cpu_sample, count)
        threads = usable_cores()
        sub = {f: np.ascontiguousarray(pres[f][..., :S, :]) for f in batch.PRES_FIELDS}
        sub["enc"] = [{f: np.ascontiguousarray(d[f][..., :S, :]) for f in batch.ENC_FIELDS} for d in pres["enc"]]
        osoa, keep2 = batch.presentation_soa(sub)
        octx = olib.afxo_ctx_new(params, len(params), key, len(key), ip)
        oshape = oracle.Shape.from_buffer_copy(bytes(shape))
        osoa = oracle.PresentationSoA.from_buffer_copy(bytes(osoa))
        ost = np.full(S, 255, np.uint8)
        t0 = time.perf_counter()
        olib.afxo_verify_presentations_soa(octx, C.byref(oshape), C.byref(osoa), S, ost.ctypes.data, threads)
        cpu_s = time.perf_counter() - t0
        assert np.array_equal(ost, got[:S]), "GPU and CPU-oracle statuses differ on the sample"
        S1 = min(S, 2048)   # single-thread figure on its own contiguous sub-batch (rows are count-strided): ~5 s of one core
        sub1 = {f: np.ascontiguousarray(pres[f][..., :S1, :]) for f in batch.PRES_FIELDS}
        sub1["enc"] = [{f: np.ascontiguousarray(d[f][..., :S1, :]) for f in batch.ENC_FIELDS} for d in pres["enc"]]
        osoa1, keep3 = batch.presentation_soa(sub1)
        osoa1 = oracle.PresentationSoA.from_buffer_copy(bytes(osoa1))
        ost1 = np.full(S1, 255, np.uint8)
        t0 = time.perf_counter()
        olib.afxo_verify_presentations_soa(octx, C.byref(oshape), C.byref(osoa1), S1, ost1.ctypes.data, 1)
        cpu1_s = time.perf_counter() - t0
        assert np.array_equal(ost1, got[:S1])
        cpu = {"value": S / cpu_s, "unit": "presentations/s", "cores": threads, "kind": "port", "cpu_model": cpu_model(),
               "sample": "value: first %d presentations of the same batch, oracle/ restated CPU path (gcc -O3 -march=native, 5x51 limbs, "
                         "NAF-5 Straus), %d threads = the cores this process may use (scheduler affinity mask capped by the cgroup CPU "
                         "quota; the machine reports %d hardware threads), %.1f s; single_thread_value: the first %d of them on one thread, "
                         "%.1f s; statuses equal to the GPU's in both" % (S, threads, os.cpu_count() or 0, cpu_s, S1, cpu1_s),
               "items": S, "seconds": cpu_s,
               "single_thread_value": S1 / cpu1_s, "single_thread_items": S1, "single_thread_seconds": cpu1_s}

    # the rest of BASELINE.json's matrix in the same line (N = 1, the C3 run): C2, C5 in the library's default mode and with the
    # fast tables, show likewise - a few steps each, every result checked (statuses; bytes of the first items against the oracle)
    secondary = None
    if rank == 0 and world == 1 and args.workload == "c3" and not args.no_secondary and not args.batch:
        issuer.close()
        issuer = None
        del pres
        torch.cuda.empty_cache()
        t0 = time.time()
        emb = {"steps": max(1, min(args.steps, 5)), "warmup": max(1, min(args.warmup, 2)), "modes": [2, 0]}
        secondary = {}
        try:
            secondary["c2"] = brief(verify_secondary(afx, batch, torch, local_rank, "c2", max(1, min(args.steps, 20)), emb["warmup"], not args.no_cpu_baseline))
            c5 = bench_issue(args, afx, batch, torch, None, 0, 1, local_rank, embedded=emb)
            secondary["c5"], secondary["c5_fast"] = brief(c5[0]), brief(c5[1])
            sh = bench_show(args, afx, batch, torch, None, 0, 1, local_rank, embedded=dict(emb, steps=max(1, min(args.steps, 20))))
            secondary["show"], secondary["show_fast"] = brief(sh[0]), brief(sh[1])
        except Exception as e:   # a secondary workload must not cost the headline line (the failure is in the line)
            secondary["error"] = "%s: %s" % (type(e).__name__, e)
        secondary["seconds"] = round(time.time() - t0, 1)
        secondary["note"] = ("after the headline's timed region, same process and device; c5 / show in the library's default secret mode (2), c5_fast / "
                             "show_fast with afx_ctx_set_secret_independent_addressing 0; full lines: python bench.py --workload c2|c5|show [--secret-mode 0]")

    if rank == 0:
        total = count * world * args.steps
        ab = algorithmic_bytes(shape)
        out = {
            "metric": "credential presentations verified/sec", "value": total / elapsed, "unit": "presentations/s",
            "n_gpus": ranks_seen, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "strong" if args.workload == "c4" else "weak", "vs_baseline": None, "dtype": "int64",
            "data": "synthetic (GPU-issued and GPU-shown credentials, random attribute values, 1% corrupted; all distinct)",
            "config": {"workload": desc, "presentations_per_gpu": count, "attributes": n, "shape": layout, "hidden": hide,
                       "algorithmic_bytes_per_presentation": ab, "parallelism": "host-sharded x%d, no collective" % world,
                       "step_pipelining": "2 streams" if args.pipelining else "off",
                       "secret_independent_addressing": {0: "nowhere (mode 0)", 1: "everywhere (mode 1)", 2: "prover-side calls (mode 2, the default)"}[secret_mode_of(args)],
                       "input_generation_s": round(gen_s, 2),
                       # the host-side legs: median of `reps` repetitions each (the spread beside it), PCIe and staging inclusive
                       "host_pointer_api_presentations_per_s": pcie and pcie["median"], "host_pointer_api_spread": pcie,
                       "host_pointer_api_over_value": (pcie["median"] / (total / elapsed)) if pcie else None,
                       "wire_blob_api_presentations_per_s": wire_rate and wire_rate["median"], "wire_blob_api_spread": wire_rate,
                       "small_call_ms": None if pcie is None else round(small_call_ms, 4), "small_call_items": 64,
                       "n1_vs_n_note": "the N=1 default workload is C3 (2^20 presentations on the one GPU, \"weak\"); N>1 defaults to C4 (2^22 in all, "
                                       "2^22/N per GPU, \"strong\"): the curve's first point is a different batch size from the rest - immaterial above "
                                       "2^17 items per GPU, where a pass fills the device",
                       "ranks_seen": ranks_seen, "rank_devices": RANK_DEVICES, "dist_backend": pg_backend, "dist_backend_fallback": pg_fallback or args.dist_backend_fallback, "launcher": "bench.py" if os.environ.get("AFX_BENCH_LAUNCHED") else
                       ("external" if world > 1 else "none"),
                       "group_api_presentations_per_s": group_rate and group_rate["median"], "group_api_spread": group_rate, "group_api_items": group_items,
                       "group_api_note": "one process, afx_group_verify_presentations over %d device(s) on host arrays (%d copies of "
                                         "rank 0's batch), PCIe and staging inclusive" % (world, world),
                       "group_api_error": group_err, "secondary": secondary},
            "roofline": roofline_of(kt, args.workload, ab, count, secret_mode_of(args) == 1),
            "valu": with_value_per_mhz(valu, total / elapsed),
            "cpu_baseline": cpu,
        }
        if UNCHECKED:
            out["unchecked_experiment"] = True
        print(json.dumps(out))
    if issuer is not None:
        issuer.close()


if __name__ == "__main__":
    main()
