"""Wire format.  CPU: pack/unpack round trip and header validation through the C ABI's
host-side parser.  GPU: serialized batch -> transposing loader -> Issuer::verify equals the oracle."""
import ctypes as C
import struct

import numpy as np
import pytest


def synthetic(count=5, seed=3):
    import aeonflux_amd as afx
    rng = np.random.default_rng(seed)
    sh = afx.Shape()
    sh.n_attributes, sh.n_responses, sh.n_hidden_scalars, sh.n_enc_proofs = 4, 4, 1, 1
    for i, k in enumerate((1, 0, 2, 3)):
        sh.kinds[i] = k
    sh.hidden_scalar_indices[0] = 0
    sh.enc_indices[0] = 3
    rb = lambda *s: rng.integers(0, 256, size=s, dtype=np.uint8)
    p = {"challenge": rb(count, 32), "responses": rb(4, count, 32), "C_x_0": rb(count, 32), "C_x_1": rb(count, 32), "C_V": rb(count, 32),
         "C_y": rb(4, count, 32), "attr_values": rb(4, count, 32),
         "enc": [{f: (rb(6, count, 32) if f == "responses" else rb(count, 32)) for f in ("challenge", "responses", "pk", "E1", "E2", "C_y_1", "C_y_2", "C_y_3", "C_y_2p")}]}
    p["attr_values"][0] = 0
    p["attr_values"][3] = 0   # secret positions do not travel
    return sh, p


def test_pack_unpack_roundtrip_and_c_parser():
    import aeonflux_amd as afx
    from aeonflux_amd import wire
    sh, p = synthetic()
    blob = wire.pack_presentations(sh, p)
    assert len(blob) == 64 + 5 * 28 * 32   # README shape: 28 cells = 896 B per presentation
    sh2, p2 = wire.unpack_presentations(blob)
    assert bytes(sh2) == bytes(sh)
    for k in ("challenge", "responses", "C_x_0", "C_x_1", "C_V", "C_y", "attr_values"):
        assert np.array_equal(p[k], p2[k]), k
    for f, v in p["enc"][0].items():
        assert np.array_equal(v, p2["enc"][0][f]), f
    lib = afx.lib()
    shape, count, off = afx.Shape(), C.c_size_t(0), C.c_size_t(0)
    assert lib.afx_wire_parse(blob, len(blob), C.byref(shape), C.byref(count), C.byref(off)) == 0
    assert (count.value, off.value, bytes(shape)) == (5, 64, bytes(sh))
    assert lib.afx_wire_cells_per_record(C.byref(sh)) == 28 and lib.afx_wire_header_bytes(C.byref(sh)) == 64
    # malformed blobs are rejected by the host parser, never read out of bounds
    bad = [blob[:-1], blob + b"\0", b"XFXP" + blob[4:], blob[:4] + struct.pack("<I", 2) + blob[8:], blob[:12] + struct.pack("<I", 27) + blob[16:],
           blob[:16] + struct.pack("<I", 99) + blob[20:], blob[:8] + struct.pack("<I", 1 << 30) + blob[12:], blob[:20], blob[:32] + b"\x09" + blob[33:]]
    for b in bad:
        assert lib.afx_wire_parse(b, len(b), C.byref(shape), C.byref(count), C.byref(off)) == afx.E_BAD_ARGS


def test_c_packers_write_what_the_python_packers_write():
    """afx_wire_pack_presentations / afx_issuance_wire_pack (host code of the library: bytes only, no GPU) against the Python
    packers of aeonflux_amd/wire.py, which the GPU tests feed to the verifier; size queries, short buffers and null arrays"""
    import aeonflux_amd as afx
    from aeonflux_amd import batch, wire
    lib = afx.lib()
    for count in (5, 1, 0):
        sh, p = synthetic(count=max(count, 1))
        p = {k: (v[..., :count, :] if k != "enc" else [{f: a[..., :count, :] for f, a in d.items()} for d in v]) for k, v in p.items()}
        p = {k: (np.ascontiguousarray(v) if k != "enc" else [{f: np.ascontiguousarray(a) for f, a in d.items()} for d in v]) for k, v in p.items()}
        want = wire.pack_presentations(sh, p)
        soa, keep = batch.presentation_soa(p)
        n = C.c_size_t(0)
        assert lib.afx_wire_pack_presentations(C.byref(sh), C.byref(soa), count, None, 0, C.byref(n)) == 0 and n.value == len(want)
        buf = np.full(len(want) + 8, 0xEE, np.uint8)
        assert lib.afx_wire_pack_presentations(C.byref(sh), C.byref(soa), count, buf.ctypes.data, len(want), C.byref(n)) == 0
        assert bytes(buf[:len(want)]) == want and (buf[len(want):] == 0xEE).all()
        assert lib.afx_wire_pack_presentations(C.byref(sh), C.byref(soa), count, buf.ctypes.data, len(want) - 1, C.byref(n)) == afx.E_BAD_ARGS
    null = afx.PresentationSoA()
    assert lib.afx_wire_pack_presentations(C.byref(sh), C.byref(null), 3, buf.ctypes.data, buf.size, C.byref(n)) == afx.E_BAD_ARGS
    rng = np.random.default_rng(9)
    rb = lambda *s: rng.integers(0, 256, size=s, dtype=np.uint8)
    kinds, cnt = [0, 2, 3, 0, 1], 7
    values = rb(5, cnt, 32)
    iss = {"t": rb(cnt, 32), "U": rb(cnt, 32), "V": rb(cnt, 32), "challenge": rb(cnt, 32), "responses": rb(10, cnt, 32)}
    want = wire.pack_issuances(kinds, values, iss)
    at = afx.AttributesSoA()
    at.n_attributes = 5
    for i, k in enumerate(kinds):
        at.kinds[i] = k
    at.values = values.ctypes.data
    s = afx.IssuanceSoA(*(iss[k].ctypes.data for k in ("t", "U", "V", "challenge", "responses")))
    assert lib.afx_issuance_wire_pack(C.byref(at), C.byref(s), 10, cnt, None, 0, C.byref(n)) == 0 and n.value == len(want)
    buf = np.zeros(len(want), np.uint8)
    assert lib.afx_issuance_wire_pack(C.byref(at), C.byref(s), 10, cnt, buf.ctypes.data, buf.size, C.byref(n)) == 0 and bytes(buf) == want
    at.kinds[2] = 9
    assert lib.afx_issuance_wire_pack(C.byref(at), C.byref(s), 10, cnt, buf.ctypes.data, buf.size, C.byref(n)) == afx.E_BAD_ARGS


@pytest.mark.gpu
@pytest.mark.parametrize("n,layout,hide,count", [(4, "SSPE", [0, 3], 130), (8, "SSPPEEEE", [4, 5, 6, 7], 20), (2, "SP", [], 3)])
def test_wire_verify_matches_oracle(n, layout, hide, count):
    import aeonflux_amd as afx
    from aeonflux_amd import wire
    from tests.helpers import corrupt, make_batch
    from tests.soa import presentation_arrays, shape_of
    params, key, ip, issuer, pres = make_batch(n, layout, hide, count, b"wire-%d" % n)
    corrupt(pres, b"wire-corrupt")
    want = [issuer.verify_presentation(p) for p in pres]
    a = presentation_arrays(pres)
    sh = afx.Shape.from_buffer_copy(bytes(shape_of(pres[0])))
    blob = wire.pack_presentations(sh, a)
    ctx = afx.Context(params, key, ip)
    status = np.full(count, 9, np.uint8)
    cnt = C.c_size_t(0)
    afx.check(afx.lib().afx_verify_presentations_wire(ctx.h, blob, len(blob), status.ctypes.data, count, C.byref(cnt)))
    ctx.close()
    assert cnt.value == count and status.tolist() == want


def test_issuance_wire_roundtrip_and_c_parser():
    import aeonflux_amd as afx
    from aeonflux_amd import wire
    rng = np.random.default_rng(11)
    rb = lambda *s: rng.integers(0, 256, size=s, dtype=np.uint8)
    kinds, count = [0, 0, 2, 3], 7
    values = rb(4, count, 32)
    iss = {"t": rb(count, 32), "U": rb(count, 32), "V": rb(count, 32), "challenge": rb(count, 32), "responses": rb(9, count, 32)}
    blob = wire.pack_issuances(kinds, values, iss)
    assert len(blob) == 32 + count * (4 + 9 + 4) * 32
    k2, v2, i2 = wire.unpack_issuances(blob)
    assert k2 == kinds and np.array_equal(v2, values) and all(np.array_equal(iss[f], i2[f]) for f in iss)
    lib = afx.lib()
    n, kk, nr, cnt, off = C.c_uint32(0), (C.c_uint8 * 32)(), C.c_uint32(0), C.c_size_t(0), C.c_size_t(0)
    args = (C.byref(n), kk, C.byref(nr), C.byref(cnt), C.byref(off))
    assert lib.afx_issuance_wire_parse(blob, len(blob), *args) == 0
    assert (n.value, list(kk[:4]), nr.value, cnt.value, off.value) == (4, kinds, 9, count, 32)
    assert lib.afx_issuance_wire_header_bytes(4) == 32 and lib.afx_issuance_wire_header_bytes(16) == 64 and lib.afx_issuance_wire_header_bytes(33) == 0
    bad = [blob[:-1], blob + b"\0", b"AFXP" + blob[4:], blob[:4] + struct.pack("<I", 2) + blob[8:], blob[:12] + struct.pack("<I", 16) + blob[16:],
           blob[:16] + struct.pack("<I", 40) + blob[20:], blob[:8] + struct.pack("<I", 1 << 30) + blob[12:], blob[:20], blob[:24] + b"\x09" + blob[25:]]
    for b in bad:
        assert lib.afx_issuance_wire_parse(b, len(b), *args) == afx.E_BAD_ARGS


@pytest.mark.gpu
def test_issuance_wire_verify_matches_oracle_and_issuer_parameters():
    import aeonflux_amd as afx
    from aeonflux_amd import wire
    from tests.helpers import make_credentials
    n, cnt = 4, 40
    d = make_credentials(n, "SSPE", cnt, b"wire-iss")
    kinds = list(d["creds"][0]["kinds"])
    values = np.zeros((n, cnt, 32), np.uint8)
    iss = {k: np.zeros((cnt, 32), np.uint8) for k in ("t", "U", "V", "challenge")}
    iss["responses"] = np.zeros((n + 5, cnt, 32), np.uint8)
    want = []
    for i, cr in enumerate(d["creds"]):
        t, V, resp = bytearray(cr["t"]), bytearray(cr["V"]), [bytearray(r) for r in cr["responses"]]
        if i % 5 == 1:
            resp[2][7] ^= 1
        if i % 5 == 3:
            V = bytearray(d["creds"][(i + 1) % cnt]["V"])
        want.append(d["user"].issuance_verify(kinds, cr["values"], bytes(t), cr["U"], bytes(V), cr["challenge"], [bytes(r) for r in resp]))
        for k in range(n):
            values[k, i] = np.frombuffer(cr["values"][k][:32], np.uint8)
        for name, v in (("t", bytes(t)), ("U", cr["U"]), ("V", bytes(V)), ("challenge", cr["challenge"])):
            iss[name][i] = np.frombuffer(v, np.uint8)
        for k in range(n + 5):
            iss["responses"][k, i] = np.frombuffer(bytes(resp[k]), np.uint8)
    blob = wire.pack_issuances(kinds, values, iss)
    user = afx.Context(d["params"], None, d["ip"])
    assert user.verify_issuances_wire(blob).tolist() == want and 0 in want and 1 in want
    assert user.issuer_parameters() == d["ip"]
    user.close()
    issuer = afx.Context(d["params"], d["key"], d["ip"])
    assert issuer.issuer_parameters() == d["ip"]
    issuer.close()
