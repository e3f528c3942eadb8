"""Batch front end over the C ABI with numpy arrays (host pointers) — the Python mirror of the reference's
`Issuer::issue`, `CredentialIssuance::verify`, `AnonymousCredential::show`, `Issuer::verify` for whole batches.

Layout convention everywhere: a field of a batch is a uint8 array of shape [count, 32]; a repeated field is
[k, count, 32] (k-major), exactly the struct-of-arrays layout of include/aeonflux_gpu.h.
"""
import ctypes as C

import numpy as np

from . import (AttributesSoA, CredentialsSoA, EncProofOut, EncProofSoA, IssuanceGroup, IssuanceSoA, IssueGroup, IssueRandomness, KeypairsSoA,
               PresentationOut, PresentationSoA, Shape, ShowGroup, ShowRandomness, check, lib)

ENC_FIELDS = ("challenge", "responses", "pk", "E1", "E2", "C_y_1", "C_y_2", "C_y_3", "C_y_2p")
PRES_FIELDS = ("challenge", "responses", "C_x_0", "C_x_1", "C_V", "C_y", "attr_values")


def _u8(a):
    a = np.ascontiguousarray(a, dtype=np.uint8)
    return a


def _attrs(kinds, values):
    s = AttributesSoA()
    s.n_attributes = len(kinds)
    for i, k in enumerate(kinds):
        s.kinds[i] = k
    s.values = values.ctypes.data
    return s


def _issue_args(n_ctx, kinds, values, t_wide, U_wide, rng_seed):
    """(AttributesSoA, IssueRandomness, IssuanceSoA, output dict, count, keepalive) of one layout's requests"""
    values, t_wide, U_wide, rng_seed = map(_u8, (values, t_wide, U_wide, rng_seed))
    cnt = t_wide.shape[0]
    req = _attrs(kinds, values)
    rnd = IssueRandomness(t_wide.ctypes.data, U_wide.ctypes.data, rng_seed.ctypes.data)
    o = {k: np.zeros((cnt, 32), np.uint8) for k in ("t", "U", "V", "challenge")}
    o["responses"] = np.zeros((n_ctx + 5, cnt, 32), np.uint8)
    out = IssuanceSoA(*(o[k].ctypes.data for k in ("t", "U", "V", "challenge", "responses")))
    return req, rnd, out, o, cnt, (values, t_wide, U_wide, rng_seed)


def issue(ctx, kinds, values, t_wide, U_wide, rng_seed, first=None, n=None):
    """Issuer::issue over a batch.  values [n,count,32], t_wide/U_wide [count,64], rng_seed [count,32].
    Returns (dict(t,U,V,challenge,responses[n+5,count,32]), status[count]).  ctx may be a Group (all its GPUs); with
    first/n only that range of the batch is issued (afx_issue_range; the other output rows stay zero / status 255)."""
    req, rnd, out, o, cnt, keep = _issue_args(ctx.n, kinds, values, t_wide, U_wide, rng_seed)
    status = np.full(cnt, 255, np.uint8)
    if first is not None:
        check(lib().afx_issue_range(ctx.h, C.byref(req), C.byref(rnd), cnt, first, n, C.byref(out), status.ctypes.data))
    elif hasattr(ctx, "member"):
        check(lib().afx_group_issue(ctx.h, C.byref(req), C.byref(rnd), cnt, C.byref(out), status.ctypes.data))
    else:
        check(lib().afx_issue(ctx.h, C.byref(req), C.byref(rnd), cnt, C.byref(out), status.ctypes.data))
    return o, status


def _positions(items, total=None):
    """per group: its items' places in the caller's order.  An item may carry "positions"; otherwise the groups are contiguous."""
    out, nxt = [], 0
    for it, cnt in items:
        pos = np.ascontiguousarray(it["positions"], dtype=np.uint64) if it.get("positions") is not None else np.arange(nxt, nxt + cnt, dtype=np.uint64)
        assert pos.size == cnt
        out.append(pos)
        nxt += cnt
    return out, nxt


def issue_mixed(ctx, items):
    """Issuer::issue over requests of DIFFERENT attribute layouts in one call (afx_issue_mixed / afx_group_issue_mixed).
    items = [dict(kinds, values [n,count,32], t_wide, U_wide, rng_seed, positions=None)], one per layout group (layouts may
    repeat).  Returns ([output dict per group], status in the caller's order)."""
    arr = (IssueGroup * max(1, len(items)))()
    keep, outs, counts = [], [], []
    for g, it in enumerate(items):
        req, rnd, out, o, cnt, k = _issue_args(ctx.n, it["kinds"], it["values"], it["t_wide"], it["U_wide"], it["rng_seed"])
        arr[g].requests, arr[g].rnd, arr[g].out, arr[g].count = req, rnd, out, cnt
        keep.append(k)
        outs.append(o)
        counts.append(cnt)
    pos, total = _positions(list(zip(items, counts)))
    total = max([total] + [int(p.max()) + 1 for p in pos if p.size])
    for g, p in enumerate(pos):
        arr[g].positions = p.ctypes.data_as(C.POINTER(C.c_uint64))
    status = np.full(total, 255, np.uint8)
    fn = lib().afx_group_issue_mixed if hasattr(ctx, "member") else lib().afx_issue_mixed
    check(fn(ctx.h, arr, len(items), status.ctypes.data, total))
    return outs, status


def _issuance_args(kinds, values, issuance, n_responses=None):
    values = _u8(values)
    iss = {k: _u8(issuance[k]) for k in ("t", "U", "V", "challenge", "responses")}
    cnt = iss["t"].shape[0]
    req = _attrs(kinds, values)
    s = IssuanceSoA(*(iss[k].ctypes.data for k in ("t", "U", "V", "challenge", "responses")))
    nr = iss["responses"].shape[0] if n_responses is None else n_responses
    return req, s, nr, cnt, (values, iss)


def verify_issuances(ctx, kinds, values, issuance, n_responses=None, first=None, n=None):
    """CredentialIssuance::verify over a batch; issuance = dict as returned by issue().  ctx may be a Group; with first/n only
    that range is verified (afx_verify_issuances_range; the other status bytes stay 255)."""
    req, s, nr, cnt, keep = _issuance_args(kinds, values, issuance, n_responses)
    status = np.full(cnt, 255, np.uint8)
    if first is not None:
        check(lib().afx_verify_issuances_range(ctx.h, C.byref(req), C.byref(s), nr, cnt, first, n, status.ctypes.data))
    elif hasattr(ctx, "member"):
        check(lib().afx_group_verify_issuances(ctx.h, C.byref(req), C.byref(s), nr, cnt, status.ctypes.data))
    else:
        check(lib().afx_verify_issuances(ctx.h, C.byref(req), C.byref(s), nr, cnt, status.ctypes.data))
    return status


def verify_issuances_mixed(ctx, items):
    """CredentialIssuance::verify over issuances of DIFFERENT layouts in one call.  items = [dict(kinds, values, issuance,
    n_responses=None, positions=None)].  Returns the statuses in the caller's order."""
    arr = (IssuanceGroup * max(1, len(items)))()
    keep, counts = [], []
    for g, it in enumerate(items):
        req, s, nr, cnt, k = _issuance_args(it["kinds"], it["values"], it["issuance"], it.get("n_responses"))
        arr[g].attrs, arr[g].issuances, arr[g].n_responses, arr[g].count = req, s, nr, cnt
        keep.append(k)
        counts.append(cnt)
    pos, total = _positions(list(zip(items, counts)))
    total = max([total] + [int(p.max()) + 1 for p in pos if p.size])
    for g, p in enumerate(pos):
        arr[g].positions = p.ctypes.data_as(C.POINTER(C.c_uint64))
    status = np.full(total, 255, np.uint8)
    fn = lib().afx_group_verify_issuances_mixed if hasattr(ctx, "member") else lib().afx_verify_issuances_mixed
    check(fn(ctx.h, arr, len(items), status.ctypes.data, total))
    return status


def _show_args(kinds, values, t, U, V, keypairs, z_wide, rng_seed, enc_seeds=None, M2=None, m3=None):
    """(CredentialsSoA, KeypairsSoA or None, ShowRandomness, PresentationOut, output dict, count, keepalive) of one layout's credentials"""
    n = len(kinds)
    values, t, U, V, z_wide, rng_seed = map(_u8, (values, t, U, V, z_wide, rng_seed))
    cnt = t.shape[0]
    nsp = sum(1 for k in kinds if k == 4)
    hs = sum(1 for k in kinds if k == 1)
    cs = CredentialsSoA()
    cs.n_attributes = n
    for i, k in enumerate(kinds):
        cs.kinds[i] = k
    keep = [values, t, U, V, z_wide, rng_seed]
    cs.values, cs.t, cs.U, cs.V = values.ctypes.data, t.ctypes.data, U.ctypes.data, V.ctypes.data
    if nsp:
        M2, m3, enc_seeds = _u8(M2), _u8(m3), _u8(enc_seeds)
        keep += [M2, m3, enc_seeds]
        cs.M2, cs.m3 = M2.ctypes.data, m3.ctypes.data
    kp = None
    if keypairs is not None:
        ka = {f: _u8(keypairs[f]) for f in ("a", "a0", "a1", "pk")}
        keep.append(ka)
        kp = KeypairsSoA(*(ka[f].ctypes.data for f in ("a", "a0", "a1", "pk")))
    rnd = ShowRandomness(z_wide.ctypes.data, rng_seed.ctypes.data, enc_seeds.ctypes.data if nsp else None)
    o = {k: np.zeros((cnt, 32), np.uint8) for k in ("challenge", "C_x_0", "C_x_1", "C_V")}
    o["responses"] = np.zeros((3 + hs, cnt, 32), np.uint8)
    o["C_y"] = np.zeros((n, cnt, 32), np.uint8)
    o["attr_values"] = np.zeros((n, cnt, 32), np.uint8)
    o["enc"] = []
    eouts = (EncProofOut * max(1, nsp))()
    for e in range(nsp):
        d = {f: np.zeros(((6, cnt, 32) if f == "responses" else (cnt, 32)), np.uint8) for f in ENC_FIELDS}
        for f, v in d.items():
            setattr(eouts[e], f, v.ctypes.data)
        o["enc"].append(d)
    out = PresentationOut()
    for f in PRES_FIELDS:
        setattr(out, f, o[f].ctypes.data)
    out.enc = C.cast(eouts, C.POINTER(EncProofOut))
    keep.append(eouts)
    return cs, kp, rnd, out, o, cnt, keep


def show(ctx, kinds, values, t, U, V, keypairs, z_wide, rng_seed, enc_seeds=None, M2=None, m3=None, first=None, n_items=None):
    """AnonymousCredential::show over a batch.  kinds: AFX_ATTR_* after hide/reveal.  values/M2/m3 [n,count,32];
    keypairs: dict(a,a0,a1,pk -> [count,32]) or None; enc_seeds [#secret points, count, 32].
    Returns (presentation dict incl. 'enc' list, Shape, status).  ctx may be a Group; with first/n_items only that range
    of the batch is shown (afx_show_range; the other output rows stay zero / status 255)."""
    cs, kp, rnd, out, o, cnt, keep = _show_args(kinds, values, t, U, V, keypairs, z_wide, rng_seed, enc_seeds, M2, m3)
    shape = Shape()
    status = np.full(cnt, 255, np.uint8)
    kpp = C.byref(kp) if kp is not None else None
    if first is not None:
        check(lib().afx_show_range(ctx.h, C.byref(cs), kpp, C.byref(rnd), cnt, first, n_items, C.byref(out), C.byref(shape), status.ctypes.data))
    elif hasattr(ctx, "member"):
        check(lib().afx_group_show(ctx.h, C.byref(cs), kpp, C.byref(rnd), cnt, C.byref(out), C.byref(shape), status.ctypes.data))
    else:
        check(lib().afx_show(ctx.h, C.byref(cs), kpp, C.byref(rnd), cnt, C.byref(out), C.byref(shape), status.ctypes.data))
    return o, shape, status


def show_mixed(ctx, items):
    """AnonymousCredential::show over credentials of DIFFERENT layouts in one call (afx_show_mixed / afx_group_show_mixed).
    items = [dict(kinds, values, t, U, V, keypairs, z_wide, rng_seed, enc_seeds=None, M2=None, m3=None, positions=None)].
    Returns ([(presentation dict, Shape) per group], status in the caller's order)."""
    arr = (ShowGroup * max(1, len(items)))()
    keep, outs, counts = [], [], []
    for g, it in enumerate(items):
        cs, kp, rnd, out, o, cnt, k = _show_args(it["kinds"], it["values"], it["t"], it["U"], it["V"], it.get("keypairs"), it["z_wide"], it["rng_seed"],
                                                 it.get("enc_seeds"), it.get("M2"), it.get("m3"))
        arr[g].creds, arr[g].rnd, arr[g].out, arr[g].count = cs, rnd, out, cnt
        if kp is not None:
            arr[g].keypairs = C.pointer(kp)
        keep.append((k, kp))
        outs.append(o)
        counts.append(cnt)
    pos, total = _positions(list(zip(items, counts)))
    total = max([total] + [int(p.max()) + 1 for p in pos if p.size])
    for g, p in enumerate(pos):
        arr[g].positions = p.ctypes.data_as(C.POINTER(C.c_uint64))
    status = np.full(total, 255, np.uint8)
    fn = lib().afx_group_show_mixed if hasattr(ctx, "member") else lib().afx_show_mixed
    check(fn(ctx.h, arr, len(items), status.ctypes.data, total))
    return [(o, Shape.from_buffer_copy(bytes(arr[g].shape_out))) for g, o in enumerate(outs)], status


def presentation_soa(p, ptr=lambda a: a.ctypes.data):
    """afx_presentation_soa over a presentation dict (numpy arrays by default; pass ptr=lambda t: t.data_ptr()
    for torch device tensors).  Returns (soa, keepalive)."""
    encs = (EncProofSoA * max(1, len(p["enc"])))()
    for e, d in enumerate(p["enc"]):
        for f in ENC_FIELDS:
            setattr(encs[e], f, ptr(d[f]))
    soa = PresentationSoA()
    for f in PRES_FIELDS:
        setattr(soa, f, ptr(p[f]))
    soa.enc = C.cast(encs, C.POINTER(EncProofSoA))
    return soa, encs


def verify_presentations(ctx, shape, p, first=None, n=None):
    """Issuer::verify over a batch of presentations held in host numpy arrays.  ctx may be a Group (all its GPUs); with
    first/n only that range is verified (afx_verify_presentations_range; the other status bytes stay 255)."""
    p = dict(p, **{f: _u8(p[f]) for f in PRES_FIELDS}, enc=[{f: _u8(d[f]) for f in ENC_FIELDS} for d in p["enc"]])
    cnt = p["challenge"].shape[0]
    soa, keep = presentation_soa(p)
    status = np.full(cnt, 255, np.uint8)
    if first is not None:
        check(lib().afx_verify_presentations_range(ctx.h, C.byref(shape), C.byref(soa), cnt, first, n, status.ctypes.data))
    elif hasattr(ctx, "member"):
        check(lib().afx_group_verify_presentations(ctx.h, C.byref(shape), C.byref(soa), cnt, status.ctypes.data))
    else:
        check(lib().afx_verify_presentations(ctx.h, C.byref(shape), C.byref(soa), cnt, status.ctypes.data))
    return status


def points_from_uniform(ctx, wide):
    wide = _u8(wide)
    out = np.zeros((wide.shape[0], 32), np.uint8)
    check(lib().afx_points_from_uniform_bytes(ctx.h, wide.ctypes.data, wide.shape[0], out.ctypes.data))
    return out


def scalars_from_wide(ctx, wide):
    wide = _u8(wide)
    out = np.zeros((wide.shape[0], 32), np.uint8)
    check(lib().afx_scalars_from_wide_bytes(ctx.h, wide.ctypes.data, wide.shape[0], out.ctypes.data))
    return out


def multiscalar_mul(ctx, scalars, points):
    """out[i] = sum_k scalars[k,i] * points[k,i];  scalars, points: [n_terms, count, 32]"""
    scalars, points = _u8(scalars), _u8(points)
    nt, cnt = scalars.shape[0], scalars.shape[1]
    out = np.zeros((cnt, 32), np.uint8)
    ok = np.zeros(cnt, np.uint8)
    check(lib().afx_multiscalar_mul(ctx.h, nt, scalars.ctypes.data, points.ctypes.data, cnt, out.ctypes.data, ok.ctypes.data))
    return out, ok


def shape_key(shape):
    """the batch-uniform part of a presentation, as a hashable key (SURVEY.md §7 "heterogeneous batches")"""
    return bytes(shape)


def verify_mixed(ctx, items):
    """Issuer::verify over presentations of different shapes: items = [(Shape, presentation dict with count 1 or more)].
    The items of one shape become one afx_presentation_group (arrays concatenated here: bytes only); the library runs one GPU
    batch per group (afx_verify_presentations_mixed, or its afx_group_* form when ctx is a Group) and writes every status at
    its presentation's place in the order given.  Returns the list of per-item status arrays."""
    from . import PresentationGroup
    groups, start, total = {}, [], 0
    for pos, (shape, p) in enumerate(items):
        k = _u8(p["challenge"]).shape[0]
        groups.setdefault(shape_key(shape), (shape, []))[1].append((total, k, p))
        start.append((total, k))
        total += k
    arr = (PresentationGroup * max(1, len(groups)))()
    keep = []
    for g, (shape, members) in enumerate(groups.values()):
        cat = {f: np.concatenate([_u8(p[f]) for _, _, p in members], axis=-2) for f in PRES_FIELDS}
        cat["enc"] = [{f: np.concatenate([_u8(p["enc"][e][f]) for _, _, p in members], axis=-2) for f in ENC_FIELDS}
                      for e in range(shape.n_enc_proofs)]
        soa, encs = presentation_soa(cat)
        positions = np.concatenate([np.arange(o, o + k, dtype=np.uint64) for o, k, _ in members])
        arr[g].shape, arr[g].batch, arr[g].count = shape, soa, positions.size
        arr[g].positions = positions.ctypes.data_as(C.POINTER(C.c_uint64))
        keep.append((cat, soa, encs, positions))
    status = np.full(total, 255, np.uint8)
    fn = lib().afx_group_verify_presentations_mixed if hasattr(ctx, "member") else lib().afx_verify_presentations_mixed
    check(fn(ctx.h, arr, len(groups), status.ctypes.data, total))
    return [status[o:o + k] for o, k in start]
