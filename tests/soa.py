"""Pack per-item presentations (oracle.Presentation) into the C ABI's struct-of-arrays batch layout
(include/aeonflux_gpu.h).  Test helper shared by CPU and GPU tests."""
import ctypes as C

import numpy as np

import oracle


def shape_of(p):
    s = oracle.Shape()
    s.n_attributes = p.n_attributes
    for i in range(p.n_attributes):
        s.kinds[i] = p.kinds[i]
    s.n_responses = p.n_responses
    s.n_hidden_scalars = p.n_hidden_scalars
    for i in range(p.n_hidden_scalars):
        s.hidden_scalar_indices[i] = p.hidden_scalar_indices[i]
    s.n_enc_proofs = p.n_enc_proofs
    for i in range(p.n_enc_proofs):
        s.enc_indices[i] = p.enc[i].index
    return s


def presentation_arrays(pres):
    """numpy uint8 arrays in SoA layout for a list of same-shape presentations"""
    p0 = pres[0]
    cnt, n, nr, ne = len(pres), p0.n_attributes, p0.n_responses, p0.n_enc_proofs

    def col(get):
        return np.stack([np.frombuffer(bytes(get(p)), dtype=np.uint8) for p in pres]).copy()
    a = {
        "challenge": col(lambda p: p.challenge),
        "responses": np.stack([col(lambda p, k=k: p.responses[k]) for k in range(nr)]) if nr else np.zeros((0, cnt, 32), np.uint8),
        "C_x_0": col(lambda p: p.C_x_0), "C_x_1": col(lambda p: p.C_x_1), "C_V": col(lambda p: p.C_V),
        "C_y": np.stack([col(lambda p, k=k: p.C_y[k]) for k in range(n)]),
        "attr_values": np.stack([col(lambda p, k=k: p.attr_values[k]) for k in range(n)]),
        "enc": [],
    }
    for e in range(ne):
        a["enc"].append({
            "challenge": col(lambda p: p.enc[e].challenge),
            "responses": np.stack([col(lambda p, k=k: p.enc[e].responses[k]) for k in range(6)]),
            **{f: col(lambda p, f=f: getattr(p.enc[e], f)) for f in ("pk", "E1", "E2", "C_y_1", "C_y_2", "C_y_3", "C_y_2p")},
        })
    return a


def soa_from_arrays(a, ptr=lambda x: x.ctypes.data):
    """build the ctypes afx_presentation_soa over arrays (numpy host arrays by default)"""
    enc_structs = (oracle.EncProofSoA * max(1, len(a["enc"])))()
    for e, d in enumerate(a["enc"]):
        for f in ("challenge", "responses", "pk", "E1", "E2", "C_y_1", "C_y_2", "C_y_3", "C_y_2p"):
            setattr(enc_structs[e], f, ptr(d[f]))
    soa = oracle.PresentationSoA()
    for f in ("challenge", "responses", "C_x_0", "C_x_1", "C_V", "C_y", "attr_values"):
        setattr(soa, f, ptr(a[f]))
    soa.enc = C.cast(enc_structs, C.POINTER(oracle.EncProofSoA))
    return soa, enc_structs


def pack_presentations(pres):
    a = presentation_arrays(pres)
    soa, enc_structs = soa_from_arrays(a)
    return shape_of(pres[0]), soa, (a, enc_structs)
