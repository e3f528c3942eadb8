"""Writes the two fixtures the Rust side is pinned with (CPU only; run from the repository root):

  tests/golden/wire.json       per golden flow: the presentation as ONE AFXP v1 section and the issuance as ONE AFXI v1 batch, written by
                               the library's own C packers (afx_wire_pack_presentations / afx_issuance_wire_pack: host code, no GPU).
                               integration/aeonflux_gpu.rs's `presentation_to_bytes` / `issuance_to_bytes` must reproduce these bytes
                               (integration/pin_against_crate.rs asserts it); tests/test_pin_fixture.py asserts the C packers still do.
  tests/golden/flows.pin.txt   the golden flows of tests/golden/flows.json as `key value...` lines, one flow between `flow <name>` and
                               `end`: what integration/pin_against_crate.rs reads (the crate has no JSON reader among its
                               dependencies) to rebuild the crate's own structs and ask the crate's own `verify` for its verdict.

python tests/gen_pin_fixture.py [--check]     --check: compare with the committed files instead of writing them"""
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")
ENC_FIELDS = ("pk", "E1", "E2", "C_y_1", "C_y_2", "C_y_3", "C_y_2p")
H = bytes.fromhex


def u8(hexes):
    return np.stack([np.frombuffer(H(h), np.uint8) for h in hexes]).copy() if hexes else np.zeros((0, 32), np.uint8)


def afxp_of(afx, pr):
    """the flow's presentation (JSON) through afx_wire_pack_presentations"""
    sh = afx.Shape()
    n = pr["n_attributes"]
    sh.n_attributes, sh.n_responses, sh.n_hidden_scalars, sh.n_enc_proofs = n, pr["n_responses"], len(pr["hidden_scalar_indices"]), len(pr["enc"])
    for i, k in enumerate(pr["kinds"]):
        sh.kinds[i] = k
    for i, h in enumerate(pr["hidden_scalar_indices"]):
        sh.hidden_scalar_indices[i] = h
    for i, q in enumerate(pr["enc"]):
        sh.enc_indices[i] = q["index"]
    cols = {"challenge": u8([pr["challenge"]]), "responses": u8(pr["responses"])[:, None, :].copy(), "C_x_0": u8([pr["C_x_0"]]), "C_x_1": u8([pr["C_x_1"]]),
            "C_V": u8([pr["C_V"]]), "C_y": u8(pr["C_y"])[:, None, :].copy(), "attr_values": u8(pr["attr_values"])[:, None, :].copy()}
    encs = (afx.EncProofSoA * max(1, len(pr["enc"])))()
    keep = []
    for e, q in enumerate(pr["enc"]):
        d = {"challenge": u8([q["challenge"]]), "responses": u8(q["responses"])[:, None, :].copy(), **{f: u8([q[f]]) for f in ENC_FIELDS}}
        keep.append(d)
        for f, a in d.items():
            setattr(encs[e], f, a.ctypes.data)
    soa = afx.PresentationSoA()
    for f, a in cols.items():
        setattr(soa, f, a.ctypes.data)
    soa.enc = C.cast(encs, C.POINTER(afx.EncProofSoA))
    need = C.c_size_t(0)
    assert afx.lib().afx_wire_pack_presentations(C.byref(sh), C.byref(soa), 1, None, 0, C.byref(need)) == 0
    blob = np.zeros(need.value, np.uint8)
    assert afx.lib().afx_wire_pack_presentations(C.byref(sh), C.byref(soa), 1, blob.ctypes.data, blob.size, C.byref(need)) == 0
    return blob.tobytes()


def afxi_of(afx, i):
    """the flow's issuance (JSON) through afx_issuance_wire_pack"""
    n, nr = len(i["kinds"]), len(i["responses"])
    at = afx.AttributesSoA()
    at.n_attributes = n
    for k, kind in enumerate(i["kinds"]):
        at.kinds[k] = kind
    values = u8([v[:64] for v in i["values"]])[:, None, :].copy()      # the first 32 bytes of every 96-byte attribute record: scalar, point or M1
    at.values = values.ctypes.data
    cols = {"t": u8([i["t"]]), "U": u8([i["U"]]), "V": u8([i["V"]]), "challenge": u8([i["challenge"]]), "responses": u8(i["responses"])[:, None, :].copy()}
    iss = afx.IssuanceSoA(*(cols[f].ctypes.data for f in ("t", "U", "V", "challenge", "responses")))
    need = C.c_size_t(0)
    assert afx.lib().afx_issuance_wire_pack(C.byref(at), C.byref(iss), nr, 1, None, 0, C.byref(need)) == 0
    blob = np.zeros(need.value, np.uint8)
    assert afx.lib().afx_issuance_wire_pack(C.byref(at), C.byref(iss), nr, 1, blob.ctypes.data, blob.size, C.byref(need)) == 0
    return blob.tobytes()


def build():
    import aeonflux_amd as afx
    flows = json.load(open(os.path.join(GOLDEN, "flows.json")))["flows"]
    wire, lines = {}, ["# written by tests/gen_pin_fixture.py from tests/golden/flows.json and the library's C packers; read by integration/pin_against_crate.rs",
                       "# kinds: issue.* / show.* are amacs::Attribute kinds (0 PublicScalar 1 SecretScalar 2 PublicPoint 3 EitherPoint 4 SecretPoint, 96-byte records:",
                       "#   value | M2 | m3); present.kinds are EncryptedAttribute kinds (0 PublicScalar 1 SecretScalar 2 PublicPoint 3 SecretPoint)"]
    for r in flows:
        i = r["issue"]
        lines += ["flow " + r["name"], "n %d" % r["n"], "params " + r["params"], "key " + r["key"], "issuer_params " + r["issuer_params"],
                  "issue.kinds " + " ".join(str(k) for k in i["kinds"]), "issue.values " + " ".join(i["values"]), "issue.status %d" % i["status"]]
        w = {}
        if i["status"] == 0:
            w["afxi"] = afxi_of(afx, i).hex()
            lines += ["issue.t " + i["t"], "issue.U " + i["U"], "issue.V " + i["V"], "issue.challenge " + i["challenge"],
                      "issue.responses " + " ".join(i["responses"]), "issuance_verify %d" % r["issuance_verify"], "issue.afxi " + w["afxi"]]
        s = r["show"]
        lines += ["show.kinds " + " ".join(str(k) for k in s["kinds"]), "show.status %d" % s["status"]]
        if s["status"] == 0:
            pr = r["presentation"]
            w["afxp"] = afxp_of(afx, pr).hex()
            lines += ["present.kinds " + " ".join(str(k) for k in pr["kinds"]), "present.hidden " + " ".join(str(h) for h in pr["hidden_scalar_indices"]),
                      "present.challenge " + pr["challenge"], "present.responses " + " ".join(pr["responses"]),
                      "present.C_x_0 " + pr["C_x_0"], "present.C_x_1 " + pr["C_x_1"], "present.C_V " + pr["C_V"], "present.C_y " + " ".join(pr["C_y"]),
                      "present.attr_values " + " ".join(pr["attr_values"]), "present.enc %d" % len(pr["enc"])]
            for e, q in enumerate(pr["enc"]):
                lines += ["present.enc.%d.index %d" % (e, q["index"]), "present.enc.%d.challenge %s" % (e, q["challenge"]),
                          "present.enc.%d.responses %s" % (e, " ".join(q["responses"]))] + ["present.enc.%d.%s %s" % (e, f, q[f]) for f in ENC_FIELDS]
            lines += ["verify %d" % r["verify"], "present.afxp " + w["afxp"]]
        lines.append("end")
        wire[r["name"]] = w
    doc = {"_source": "tests/gen_pin_fixture.py: the golden flows' presentation / issuance through afx_wire_pack_presentations / afx_issuance_wire_pack "
                      "(count = 1); the Rust writers of integration/aeonflux_gpu.rs must produce the same bytes", "wire": wire}
    return json.dumps(doc, indent=1) + "\n", "\n".join(lines) + "\n"


if __name__ == "__main__":
    wire_json, pin_txt = build()
    targets = ((os.path.join(GOLDEN, "wire.json"), wire_json), (os.path.join(GOLDEN, "flows.pin.txt"), pin_txt))
    if "--check" in sys.argv:
        for path, text in targets:
            assert open(path).read() == text, path + " is stale: python tests/gen_pin_fixture.py"
        print("fixtures are current")
    else:
        for path, text in targets:
            open(path, "w").write(text)
        print("wrote", [p for p, _ in targets])
