"""CPU-only: seeded mutation fuzzing of the entry points that take NETWORK bytes - afx_wire_parse, afx_wire_section_bytes,
afx_verify_presentations_wire, afx_verify_presentations_mixed_wire, afx_issuance_wire_parse, afx_verify_issuances_wire - on the
host simulation of the engine (fake HIP runtime, tests/hostsim/fake_hip.cpp) built with AddressSanitizer + UBSan.  Valid AFXP / AFXI /
mixed streams from the packers are damaged in >= 10^5 ways (every edge value in every header word, truncations around every
32-byte boundary, spliced and duplicated sections, random bit flips / truncations / field copies); every call must answer AFX_OK or
AFX_E_BAD_ARGS - never a sanitizer report, never a crash, never another code.  The mutation loop is C++ (tests/hostsim/wire_fuzz.cpp:
Python itself runs ~50x slower under the sanitizer's allocator); this file writes the valid streams with the Python packers and
runs it.  (Sanitizers run on the CPU build only.)"""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "aeonflux_amd", "csrc")
MUTATIONS = int(os.environ.get("AFX_FUZZ_MUTATIONS", "120000"))


@pytest.fixture(scope="module")
def fuzzer(tmp_path_factory):
    out = str(tmp_path_factory.mktemp("wirefuzz") / "wire_fuzz")
    srcs = [os.path.join(CSRC, f) for f in ("engine.cpp", "plans.cpp", "statements.cpp", "statements_prove.cpp", "statements_setup.cpp", "group.cpp", "mixed.cpp", "wire.cpp")]
    srcs += [os.path.join(ROOT, "tests", "hostsim", "fake_hip.cpp"), os.path.join(ROOT, "tests", "hostsim", "wire_fuzz.cpp")]
    r = subprocess.run(["g++", "-g", "-O1", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-std=c++17", "-D__HIP_PLATFORM_AMD__",
                        "-I/opt/rocm/include", "-pthread", "-o", out] + srcs, capture_output=True, text=True)
    if r.returncode != 0:
        pytest.skip("cannot build the fuzzer: " + r.stderr[-400:])
    return out


def test_byte_parsers_survive_a_hundred_thousand_mutations(fuzzer, tmp_path):
    import aeonflux_amd as afx
    from aeonflux_amd import batch, wire
    from tests.helpers import make_credentials
    d = make_credentials(4, "SSPE", 1, b"wire-fuzz")
    z = lambda *s: np.zeros(s, np.uint8)

    def shape_of(kinds, hs_idx, enc_idx, nr):
        sh = afx.Shape()
        sh.n_attributes, sh.n_responses, sh.n_hidden_scalars, sh.n_enc_proofs = len(kinds), nr, len(hs_idx), len(enc_idx)
        for i, k in enumerate(kinds):
            sh.kinds[i] = k
        for i, k in enumerate(hs_idx):
            sh.hidden_scalar_indices[i] = k
        for i, k in enumerate(enc_idx):
            sh.enc_indices[i] = k
        return sh

    def pres(sh, cnt):
        n, ne = sh.n_attributes, sh.n_enc_proofs
        return {"challenge": z(cnt, 32), "responses": z(sh.n_responses, cnt, 32), "C_x_0": z(cnt, 32), "C_x_1": z(cnt, 32), "C_V": z(cnt, 32), "C_y": z(n, cnt, 32),
                "attr_values": z(n, cnt, 32), "enc": [{f: (z(6, cnt, 32) if f == "responses" else z(cnt, 32)) for f in batch.ENC_FIELDS} for _ in range(ne)]}
    shA, shB = shape_of((1, 0, 2, 3), (0,), (3,), 4), shape_of((0, 0, 2, 2), (), (), 3)
    blobA, blobB = wire.pack_presentations(shA, pres(shA, 5)), wire.pack_presentations(shB, pres(shB, 3))
    iss = {k: z(3, 32) for k in ("t", "U", "V", "challenge")}
    iss["responses"] = z(9, 3, 32)
    files = {"params.bin": d["params"], "key.bin": d["key"], "ip.bin": d["ip"], "a.afxp": blobA, "b.afxp": blobB,
             "mixed.afxp": blobA + blobB + wire.pack_presentations(shA, pres(shA, 2)), "i.afxi": wire.pack_issuances([0, 0, 2, 3], z(4, 3, 32), iss)}
    for name, data in files.items():
        (tmp_path / name).write_bytes(data)
    r = subprocess.run([fuzzer, str(tmp_path), str(MUTATIONS)], env=dict(os.environ, ASAN_OPTIONS="detect_leaks=0"), capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0 and "wire fuzz ok" in r.stdout, (r.stdout[-1500:], r.stderr[-3000:])
    assert int(r.stdout.split("wire fuzz ok:")[1].split()[0]) >= min(MUTATIONS, 100000), r.stdout
