"""Wire format of a batch of same-shape presentations ("AFXP" v1, include/aeonflux_gpu.h): the reference crate
defines no serialisation for ProofOfValidCredential (src/nizk/presentation.rs:117-127), so this is the engine's
own, in the crate's `u32le n || 32-byte items` style (src/parameters.rs:155-184).  Pure byte shuffling."""
import struct

import numpy as np

from . import Shape

ENC_ORDER = ("challenge", "responses", "pk", "E1", "E2", "C_y_1", "C_y_2", "C_y_3", "C_y_2p")


def header(shape, count):
    n, nr, hs, ne = shape.n_attributes, shape.n_responses, shape.n_hidden_scalars, shape.n_enc_proofs
    pub = sum(1 for i in range(n) if shape.kinds[i] in (0, 2))
    cells = 1 + nr + 3 + n + pub + 14 * ne
    h = b"AFXP" + struct.pack("<7I", 1, count, cells, n, nr, hs, ne)
    h += bytes(shape.kinds[:n])
    h += b"".join(struct.pack("<H", shape.hidden_scalar_indices[i]) for i in range(hs))
    h += b"".join(struct.pack("<H", shape.enc_indices[i]) for i in range(ne))
    h += bytes(-len(h) % 32)
    return h, cells


def pack_presentations(shape, p):
    """SoA presentation dict ([count,32] / [k,count,32] uint8 arrays) -> wire bytes"""
    count = p["challenge"].shape[0]
    n = shape.n_attributes
    cols = [p["challenge"][None], p["responses"], p["C_x_0"][None], p["C_x_1"][None], p["C_V"][None], p["C_y"]]
    cols += [p["attr_values"][i][None] for i in range(n) if shape.kinds[i] in (0, 2)]
    for d in p["enc"]:
        cols += [d[f] if d[f].ndim == 3 else d[f][None] for f in ENC_ORDER]
    soa = np.concatenate([np.asarray(c, dtype=np.uint8) for c in cols], axis=0)      # [cells, count, 32]
    h, cells = header(shape, count)
    assert soa.shape == (cells, count, 32)
    return h + np.ascontiguousarray(soa.transpose(1, 0, 2)).tobytes()


def unpack_presentations(blob):
    """wire bytes -> (Shape, SoA presentation dict)"""
    assert blob[:4] == b"AFXP"
    ver, count, cells, n, nr, hs, ne = struct.unpack("<7I", blob[4:32])
    assert ver == 1
    shape = Shape()
    shape.n_attributes, shape.n_responses, shape.n_hidden_scalars, shape.n_enc_proofs = n, nr, hs, ne
    o = 32
    for i in range(n):
        shape.kinds[i] = blob[o + i]
    o += n
    for i in range(hs):
        shape.hidden_scalar_indices[i] = struct.unpack("<H", blob[o:o + 2])[0]
        o += 2
    for i in range(ne):
        shape.enc_indices[i] = struct.unpack("<H", blob[o:o + 2])[0]
        o += 2
    o = (o + 31) & ~31
    rec = np.frombuffer(blob, dtype=np.uint8, offset=o).reshape(count, cells, 32).transpose(1, 0, 2)
    it = iter(range(cells))
    take = lambda k: np.ascontiguousarray(np.stack([rec[next(it)] for _ in range(k)]))
    p = {"challenge": take(1)[0], "responses": take(nr), "C_x_0": take(1)[0], "C_x_1": take(1)[0], "C_V": take(1)[0], "C_y": take(n)}
    av = np.zeros((n, count, 32), np.uint8)
    for i in range(n):
        if shape.kinds[i] in (0, 2):
            av[i] = take(1)[0]
    p["attr_values"] = av
    p["enc"] = []
    for _ in range(ne):
        p["enc"].append({f: (take(6) if f == "responses" else take(1)[0]) for f in ENC_ORDER})
    return shape, p


def pack_mixed(items):
    """[(Shape, SoA presentation dict)] -> AFXP sections back to back, in the order given (a request stream of mixed shapes)"""
    return b"".join(pack_presentations(shape, p) for shape, p in items)


def verify_mixed_wire(ctx, blob):
    """afx_verify_presentations_mixed_wire: statuses of a stream of AFXP sections, in stream order.  ctx may be a Group."""
    import ctypes as C
    from . import check, lib
    n = C.c_size_t(0)
    cap = max(1, len(blob) // 32)   # a record is at least one cell
    status = np.full(cap, 255, np.uint8)
    fn = lib().afx_group_verify_presentations_mixed_wire if hasattr(ctx, "member") else lib().afx_verify_presentations_mixed_wire
    check(fn(ctx.h, blob, len(blob), status.ctypes.data, cap, C.byref(n)))
    return status[:n.value]


def verify_wire(ctx, blob, first=None, n=None):
    """Issuer::verify over one AFXP batch (afx_verify_presentations_wire); ctx may be a Group (its devices take byte ranges of the
    blob); with first/n only those records are verified (afx_verify_presentations_wire_range; the other status bytes stay 255)."""
    import ctypes as C
    from . import check, lib
    cnt = C.c_size_t(0)
    cap = max(1, len(blob) // 32)
    status = np.full(cap, 255, np.uint8)
    if first is not None:
        check(lib().afx_verify_presentations_wire_range(ctx.h, blob, len(blob), first, n, status.ctypes.data, cap, C.byref(cnt)))
    elif hasattr(ctx, "member"):
        check(lib().afx_group_verify_presentations_wire(ctx.h, blob, len(blob), status.ctypes.data, cap, C.byref(cnt)))
    else:
        check(lib().afx_verify_presentations_wire(ctx.h, blob, len(blob), status.ctypes.data, cap, C.byref(cnt)))
    return status[:cnt.value]


# ---- CredentialIssuance batches ("AFXI" v1) -------------------------------------------------------
def pack_issuances(kinds, values, iss):
    """kinds: AFX_ATTR_* per position; values [n,count,32]; iss: dict t,U,V,challenge [count,32], responses [nr,count,32]"""
    n = len(kinds)
    values = np.asarray(values, dtype=np.uint8).reshape(n, -1, 32) if n else np.zeros((0, iss["t"].shape[0], 32), np.uint8)
    count, nr = iss["t"].shape[0], iss["responses"].shape[0]
    cols = [iss["t"][None], iss["U"][None], iss["V"][None], iss["challenge"][None], iss["responses"], values]
    soa = np.concatenate([np.asarray(c, dtype=np.uint8) for c in cols], axis=0)
    cells = 4 + nr + n
    assert soa.shape == (cells, count, 32)
    h = b"AFXI" + struct.pack("<5I", 1, count, cells, n, nr) + bytes(kinds)
    h += bytes(-len(h) % 32)
    return h + np.ascontiguousarray(soa.transpose(1, 0, 2)).tobytes()


def unpack_issuances(blob):
    """wire bytes -> (kinds, values [n,count,32], issuance dict)"""
    assert blob[:4] == b"AFXI"
    ver, count, cells, n, nr = struct.unpack("<5I", blob[4:24])
    assert ver == 1 and cells == 4 + nr + n
    kinds = list(blob[24:24 + n])
    o = (24 + n + 31) & ~31
    rec = np.frombuffer(blob, dtype=np.uint8, offset=o).reshape(count, cells, 32).transpose(1, 0, 2)
    c = lambda a: np.ascontiguousarray(a)
    iss = {"t": c(rec[0]), "U": c(rec[1]), "V": c(rec[2]), "challenge": c(rec[3]), "responses": c(rec[4:4 + nr])}
    return kinds, c(rec[4 + nr:]), iss
