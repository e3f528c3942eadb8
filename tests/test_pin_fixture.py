"""The fixtures that pin the Rust side (integration/aeonflux_gpu.rs, integration/pin_against_crate.rs) and, when a maintainer has run
the crate-side test, the flows the aeonflux crate ITSELF made (tests/golden/flows_from_crate.json: SURVEY.md section 8c's way from
"parity unpinned" to pinned).

CPU: tests/golden/wire.json and flows.pin.txt are what tests/gen_pin_fixture.py writes today (the library's C packers, host code);
     the wire bytes parse back through the C parsers; crate-made flows (if present) get the crate's verdicts from the ORACLE.
GPU: crate-made flows (if present) get the crate's verdicts from the engine, through the column-array door and the wire door."""
import ctypes as C
import json
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
CRATE_FLOWS = os.environ.get("AFX_CRATE_FLOWS", os.path.join(GOLDEN, "flows_from_crate.json"))   # (the override: a rehearsal of these tests on oracle-made flows)
H = bytes.fromhex


def crate_flows():
    if not os.path.exists(CRATE_FLOWS):
        pytest.skip("tests/golden/flows_from_crate.json is not there: run integration/pin_against_crate.rs inside the crate (INTEGRATION.md section 5)")
    return json.load(open(CRATE_FLOWS))["flows"]


def test_committed_fixtures_are_what_the_c_packers_write():
    from tests import gen_pin_fixture as g
    wire_json, pin_txt = g.build()
    assert open(os.path.join(GOLDEN, "wire.json")).read() == wire_json, "stale: python tests/gen_pin_fixture.py"
    assert open(os.path.join(GOLDEN, "flows.pin.txt")).read() == pin_txt, "stale: python tests/gen_pin_fixture.py"


def test_wire_fixture_parses_back_through_the_c_parsers(flows):
    import aeonflux_amd as afx
    from aeonflux_amd import wire
    doc = json.load(open(os.path.join(GOLDEN, "wire.json")))["wire"]
    lib = afx.lib()
    seen = 0
    for r in flows:
        w = doc[r["name"]]
        if "afxp" in w:
            blob, pr = H(w["afxp"]), r["presentation"]
            shape, count, off = afx.Shape(), C.c_size_t(0), C.c_size_t(0)
            assert lib.afx_wire_parse(blob, len(blob), C.byref(shape), C.byref(count), C.byref(off)) == 0
            assert count.value == 1 and list(shape.kinds[:shape.n_attributes]) == pr["kinds"] and shape.n_enc_proofs == len(pr["enc"])
            sh2, p2 = wire.unpack_presentations(blob)
            assert p2["challenge"][0].tobytes().hex() == pr["challenge"] and [p2["C_y"][k, 0].tobytes().hex() for k in range(sh2.n_attributes)] == pr["C_y"]
            for e, q in enumerate(pr["enc"]):
                assert p2["enc"][e]["C_y_2p"][0].tobytes().hex() == q["C_y_2p"] and p2["enc"][e]["responses"][5, 0].tobytes().hex() == q["responses"][5]
            seen += 1
        if "afxi" in w:
            blob, i = H(w["afxi"]), r["issue"]
            n, kinds, nr, count, off = C.c_uint32(0), (C.c_uint8 * 32)(), C.c_uint32(0), C.c_size_t(0), C.c_size_t(0)
            assert lib.afx_issuance_wire_parse(blob, len(blob), C.byref(n), kinds, C.byref(nr), C.byref(count), C.byref(off)) == 0
            assert (n.value, nr.value, count.value, list(kinds[:n.value])) == (len(i["kinds"]), len(i["responses"]), 1, i["kinds"])
            rec = blob[off.value:]
            assert rec[:32].hex() == i["t"] and rec[96:128].hex() == i["challenge"] and rec[32 * (4 + nr.value):32 * (5 + nr.value)].hex() == i["values"][0][:64]
            seen += 1
    assert seen >= 25


def oracle_presentation(r):
    from tests.helpers import pres_from_json
    return pres_from_json(r)


def test_crate_made_flows_get_the_crates_verdicts_from_the_oracle():
    """the ORACLE's verifiers on what the reference itself issued and showed (and on damaged copies): SURVEY.md section 8c, direction (b)"""
    import oracle
    flows = crate_flows()
    assert len(flows) >= 10 and any(r["verify"] == 1 for r in flows) and any(r["verify"] == 0 for r in flows)
    for r in flows:
        issuer = oracle.Ctx(H(r["params"]), H(r["key"]), H(r["issuer_params"]))
        user = oracle.Ctx(H(r["params"]), None, H(r["issuer_params"]))
        i = r["issue"]
        assert user.issuance_verify(i["kinds"], [H(v) for v in i["values"]], H(i["t"]), H(i["U"]), H(i["V"]), H(i["challenge"]), [H(x) for x in i["responses"]]) == r["issuance_verify"], r["name"]
        assert issuer.verify_presentation(oracle_presentation(r)) == r["verify"], r["name"]


@pytest.mark.gpu
def test_crate_made_flows_get_the_crates_verdicts_from_the_gpu():
    """... and the engine's, through the column-array door, the serialized door and the issuance verifier"""
    import aeonflux_amd as afx
    from aeonflux_amd import batch, wire
    from tests.helpers import gpu_verify
    from tests.soa import presentation_arrays, shape_of
    for r in crate_flows():
        ictx = afx.Context(H(r["params"]), H(r["key"]), H(r["issuer_params"]))
        uctx = afx.Context(H(r["params"]), None, H(r["issuer_params"]))
        p = oracle_presentation(r)     # (a plain container of the JSON's bytes)
        assert gpu_verify(afx, ictx, [p]) == [r["verify"]], r["name"]
        blob = wire.pack_presentations(afx.Shape.from_buffer_copy(bytes(shape_of(p))), presentation_arrays([p]))
        assert wire.verify_wire(ictx, blob).tolist() == [r["verify"]], r["name"]
        i = r["issue"]
        values = np.stack([np.frombuffer(H(v)[:32], np.uint8) for v in i["values"]])[:, None, :].copy()
        iss = {k: np.frombuffer(H(i[k]), np.uint8)[None, :].copy() for k in ("t", "U", "V", "challenge")}
        iss["responses"] = np.stack([np.frombuffer(H(x), np.uint8) for x in i["responses"]])[:, None, :].copy()
        assert batch.verify_issuances(uctx, i["kinds"], values, iss).tolist() == [r["issuance_verify"]], r["name"]
        ictx.close()
        uctx.close()
