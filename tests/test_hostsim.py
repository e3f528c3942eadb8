"""CPU-only: the engine's host half (context parsing, plan assembly, STROBE schedule compilation, C ABI argument
handling) built against a fake HIP runtime with no-op kernels (tests/hostsim/fake_hip.cpp) under ASan/UBSan, and
driven through every entry point.  Results are meaningless (the kernels are stubs); what is checked is that the
host code builds every launch list without touching memory it does not own and fails cleanly on bad input."""
import os
import subprocess
import sys
import textwrap

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "aeonflux_amd", "csrc")


@pytest.fixture(scope="module")
def hostsim_lib(tmp_path_factory):
    out = str(tmp_path_factory.mktemp("hostsim") / "libafx_hostsim.so")
    srcs = [os.path.join(CSRC, f) for f in ("engine.cpp", "plans.cpp", "statements.cpp", "statements_prove.cpp", "statements_setup.cpp", "group.cpp", "mixed.cpp", "wire.cpp")]
    srcs.append(os.path.join(ROOT, "tests", "hostsim", "fake_hip.cpp"))
    cmd = ["g++", "-g", "-O1", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-fPIC", "-std=c++17",
           "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", "-shared", "-pthread", "-o", out] + srcs
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        pytest.skip("cannot build the host simulation: " + r.stderr[-400:])
    return out


DRIVER = r"""
import os, sys, ctypes as C
sys.path.insert(0, %(root)r)
import numpy as np
import aeonflux_amd as afx
afx.LIB_PATH = %(lib)r
from aeonflux_amd import batch, wire
from tests.helpers import make_credentials
from tests.soa import presentation_arrays, shape_of
for n, layout, hide in ((4, "SSPE", [0, 3]), (16, "SSSSSSSSPPPPEEEE", [12, 13, 14, 15]), (8, "SSPPEEEE", [4, 5, 6, 7]), (1, "S", []), (3, "ESS", [0])):
    d = make_credentials(n, layout, 3, b"hostsim-%%d" %% n)
    ctx = afx.Context(d["params"], d["key"], d["ip"])
    creds = d["creds"]
    kinds = creds[0]["kinds"]
    values = np.stack([np.stack([np.frombuffer(c["values"][i][:32], np.uint8) for c in creds]) for i in range(n)])
    rb = lambda *s: np.zeros(s, np.uint8)
    iss, st = batch.issue(ctx, kinds, values, rb(3, 64), rb(3, 64), rb(3, 32))
    batch.verify_issuances(ctx, kinds, values, iss)
    batch.verify_issuances(ctx, kinds, values, iss, n_responses=2)
    k2 = list(kinds)
    for i in hide:
        k2[i] = 1 if k2[i] == 0 else 4
    nsp = sum(1 for k in k2 if k == 4)
    kp = {f: rb(3, 32) for f in ("a", "a0", "a1", "pk")}
    pres, shape, st = batch.show(ctx, k2, values, iss["t"], iss["U"], iss["V"], kp, rb(3, 64), rb(3, 32), rb(max(nsp, 1), 3, 32), values, values)
    batch.verify_presentations(ctx, shape, pres)
    # challenge trace: rows for the main proof and each proof of encryption; too small a buffer is refused
    ctx.set_challenge_trace(1 + nsp, 3)
    batch.verify_presentations(ctx, shape, pres)
    batch.verify_issuances(ctx, kinds, values, iss)
    assert ctx.get_challenge_trace().shape == (1 + nsp, 3, 32)
    ctx.set_challenge_trace(1, 2)
    try:
        batch.verify_presentations(ctx, shape, pres)
        raise SystemExit("trace overflow accepted")
    except afx.AfxError as e:
        assert e.rc == afx.E_BAD_ARGS
    ctx.set_challenge_trace(0, 0)
    iblob = wire.pack_issuances(kinds, values, iss)
    assert len(ctx.verify_issuances_wire(iblob)) == 3 and len(ctx.issuer_parameters()) == 64
    blob = wire.pack_presentations(shape, pres)
    stt, cnt = np.zeros(3, np.uint8), C.c_size_t(0)
    assert afx.lib().afx_verify_presentations_wire(ctx.h, blob, len(blob), stt.ctypes.data, 3, C.byref(cnt)) == 0 and cnt.value == 3
    assert afx.lib().afx_verify_presentations_wire(ctx.h, blob[:-1], len(blob) - 1, stt.ctypes.data, 3, C.byref(cnt)) == afx.E_BAD_ARGS
    # a stream of mixed shapes: this statement's shape twice around a second shape (hidden scalar revealed / nothing else),
    # as back-to-back AFXP sections and as struct-of-arrays groups with positions; truncated streams are refused
    sh2 = afx.Shape.from_buffer_copy(bytes(shape))
    sh2.n_enc_proofs = 0
    pres2 = dict(pres, enc=[])
    mixed_stream = wire.pack_presentations(shape, pres) + wire.pack_presentations(sh2, pres2) + wire.pack_presentations(shape, pres)
    assert wire.verify_mixed_wire(ctx, mixed_stream).tolist() == [0x5a] * 9
    assert afx.lib().afx_verify_presentations_mixed_wire(ctx.h, mixed_stream[:-32], len(mixed_stream) - 32, stt.ctypes.data, 3, C.byref(cnt)) == afx.E_BAD_ARGS
    assert afx.lib().afx_verify_presentations_mixed_wire(ctx.h, mixed_stream, len(mixed_stream), stt.ctypes.data, 3, C.byref(cnt)) == afx.E_BAD_ARGS and cnt.value == 9
    sl = C.c_size_t(0)
    assert afx.lib().afx_wire_section_bytes(mixed_stream, len(mixed_stream), C.byref(sl)) == 0 and sl.value == len(blob)
    assert afx.lib().afx_wire_section_bytes(mixed_stream, len(blob) - 1, C.byref(sl)) == afx.E_BAD_ARGS
    got = batch.verify_mixed(ctx, [(shape, pres), (sh2, pres2), (shape, pres)])
    assert [g.tolist() for g in got] == [[0x5a] * 3] * 3
    # requests / issuances / credentials of several layouts in one call, statuses scattered to the caller's order; a position used
    # twice or outside the status array is refused before anything runs
    kflip = [2 if k == 0 else k for k in kinds]
    ia = dict(kinds=kinds, values=values, t_wide=rb(3, 64), U_wide=rb(3, 64), rng_seed=rb(3, 32), positions=[4, 0, 2])
    ib = dict(kinds=kflip, values=values, t_wide=rb(3, 64), U_wide=rb(3, 64), rng_seed=rb(3, 32), positions=[1, 3, 5])
    outs_m, st_m = batch.issue_mixed(ctx, [ia, ib])
    assert st_m.tolist() == [0x5a] * 6 and len(outs_m) == 2 and outs_m[1]["responses"].shape == (n + 5, 3, 32)
    for bad_pos in ([1, 3, 4], [1, 1, 5]):
        try:
            batch.issue_mixed(ctx, [ia, dict(ib, positions=bad_pos)])
            raise SystemExit("bad positions accepted")
        except (afx.AfxError, AssertionError) as ex:
            assert getattr(ex, "rc", afx.E_BAD_ARGS) == afx.E_BAD_ARGS
    st_m = batch.verify_issuances_mixed(ctx, [dict(kinds=kinds, values=values, issuance=iss, positions=[5, 4, 3]),
                                              dict(kinds=kflip, values=values, issuance=iss, n_responses=2, positions=[0, 1, 2])])
    assert st_m.tolist() == [1, 1, 1, 0x5a, 0x5a, 0x5a]   # the group with the wrong response count fails whole, on the host
    sa = dict(kinds=k2, values=values, t=iss["t"], U=iss["U"], V=iss["V"], keypairs=kp, z_wide=rb(3, 64), rng_seed=rb(3, 32), enc_seeds=rb(max(nsp, 1), 3, 32),
              M2=values, m3=values, positions=[0, 2, 4])
    sb = dict(sa, kinds=list(kinds), enc_seeds=None, positions=[1, 3, 5])
    outs_s, st_s = batch.show_mixed(ctx, [sa, sb])
    assert st_s.tolist() == [0x5a] * 6 and bytes(outs_s[0][1]) == bytes(shape) and outs_s[1][1].n_enc_proofs == 0
    # mis-shaped requests take the fail-all path
    bad = afx.Shape.from_buffer_copy(bytes(shape))
    bad.n_responses += 1   # claims a row the arrays do not have: must be rejected without reading them
    batch.verify_presentations(ctx, bad, pres)
    bad = afx.Shape.from_buffer_copy(bytes(shape))
    bad.n_attributes = 40
    batch.verify_presentations(ctx, bad, pres)
    batch.multiscalar_mul(ctx, rb(24, 3, 32), rb(24, 3, 32))
    ms = rb(3, 64)
    o = [rb(3, 32) for _ in range(4)]
    assert afx.lib().afx_keypairs_derive(ctx.h, ms.ctypes.data, 3, *(x.ctypes.data for x in o)) == 0
    # the cold-path entry points that build and destroy a context of their own
    ks = d["key"][:4 + 32 * (4 + n)]
    W, ipb = C.create_string_buffer(32), C.create_string_buffer(64)
    assert afx.lib().afx_issuer_keygen(0, d["params"], len(d["params"]), ks, len(ks), W, ipb) == 0
    assert afx.lib().afx_issuer_keygen(0, d["params"], len(d["params"]), ks[:-1], len(ks) - 1, W, ipb) == afx.E_BAD_PARAMS
    stream = bytes(range(256)) * 8
    pout, used = C.create_string_buffer(8192), C.c_size_t(0)
    afx.lib().afx_system_parameters_generate(0, n, stream, len(stream), pout, 8192, C.byref(used))   # stub kernels: any rc, no bad access
    kpsoa = afx.KeypairsSoA(*(x.ctypes.data for x in o))
    e = [rb(3, 32) for _ in range(8)]
    stx = rb(3)
    assert afx.lib().afx_encrypt(ctx.h, C.byref(kpsoa), e[0].ctypes.data, e[1].ctypes.data, e[2].ctypes.data, 3, e[3].ctypes.data, e[4].ctypes.data, stx.ctypes.data) == 0
    assert afx.lib().afx_decrypt(ctx.h, C.byref(kpsoa), e[3].ctypes.data, e[4].ctypes.data, 3, e[5].ctypes.data, e[6].ctypes.data, e[7].ctypes.data, None, stx.ctypes.data) == 0
    # the key-independent schedule: no NAF jobs, every launch list still assembles
    assert afx.lib().afx_ctx_set_fixed_key_schedule(ctx.h, 1) == 0
    batch.verify_presentations(ctx, shape, pres)
    batch.issue(ctx, kinds, values, rb(3, 64), rb(3, 64), rb(3, 32))
    assert afx.lib().afx_ctx_set_fixed_key_schedule(ctx.h, 0) == 0
    # the plan of large passes (one chain per job, NAF schedules for the key) at this size too: the default here is the latency plan
    ctx.set_small_batch_items(0)
    batch.verify_presentations(ctx, shape, pres)
    big_jobs = ctx.plan_stats()["msm_jobs"]
    batch.issue(ctx, kinds, values, rb(3, 64), rb(3, 64), rb(3, 32))
    batch.show(ctx, k2, values, iss["t"], iss["U"], iss["V"], kp, rb(3, 64), rb(3, 32), rb(max(nsp, 1), 3, 32), values, values)
    ctx.set_small_batch_items(2048)
    batch.verify_presentations(ctx, shape, pres)
    assert ctx.plan_stats()["msm_jobs"] >= big_jobs   # one chain per term
    # secret-independent addressing: prover plans mark every term, the verifier's plan the key's terms; the fake launcher checks
    # that each launch's flag is the OR of its terms' flags and that the 4-bit tables exist
    ctx.set_secret_independent_addressing(True)
    # the verifier's secrets: every term of Z - the key's own and the key-derived y_i * m_i of revealed scalars - in both plans
    # (marked before a small pass splits Z into one chain per term); nothing else of Issuer::verify
    z_terms = 2 + shape.n_attributes + sum(1 for k in list(shape.kinds)[:shape.n_attributes] if k == 0)
    for thr in (0, 2048):
        ctx.set_small_batch_items(thr)
        batch.verify_presentations(ctx, shape, pres)
        assert shape.n_attributes != n or ctx.plan_stats()["secret_terms"] == z_terms, (ctx.plan_stats(), z_terms)
    batch.issue(ctx, kinds, values, rb(3, 64), rb(3, 64), rb(3, 32))
    assert ctx.plan_stats()["secret_terms"] > 3 * n
    batch.show(ctx, k2, values, iss["t"], iss["U"], iss["V"], kp, rb(3, 64), rb(3, 32), rb(max(nsp, 1), 3, 32), values, values)
    assert afx.lib().afx_encrypt(ctx.h, C.byref(kpsoa), e[0].ctypes.data, e[1].ctypes.data, e[2].ctypes.data, 3, e[3].ctypes.data, e[4].ctypes.data, stx.ctypes.data) == 0
    assert ctx.plan_stats()["fixed_additions"] == 0   # E2 = a*E1 + M1: one variable base
    batch.issue(ctx, kinds, values, rb(3, 64), rb(3, 64), rb(3, 32))
    sec_small = ctx.plan_stats()
    ctx.set_small_batch_items(0)
    batch.issue(ctx, kinds, values, rb(3, 64), rb(3, 64), rb(3, 32))
    sec_fixed = ctx.plan_stats()["fixed_additions"]
    ctx.set_secret_independent_addressing(False)
    batch.issue(ctx, kinds, values, rb(3, 64), rb(3, 64), rb(3, 32))
    assert sec_fixed > 2 * ctx.plan_stats()["fixed_additions"] > 0   # 43 additions per secret fixed-base term instead of 20
    assert ctx.plan_stats()["secret_terms"] == 0
    # ... but a SMALL prover pass takes the secret-independent plan in every mode: its segmented chains are the faster ones
    ctx.set_small_batch_items(2048)
    batch.issue(ctx, kinds, values, rb(3, 64), rb(3, 64), rb(3, 32))
    if not (afx.DEFAULT_PLAN_VARIANTS & afx.VARIANT_SEGMENTS_1):   # (whole chains forced: then mode 0 means the fast tables at every size)
        assert ctx.plan_stats() == sec_small and sec_small["secret_terms"] > 3 * n
    mhz = C.c_double(-1)
    assert afx.lib().afx_ctx_get_core_clock_mhz(ctx.h, C.byref(mhz)) == 0 and mhz.value >= 0
    # a range of a batch, and the same batch over a two-member group (two fake devices)
    st_r = batch.verify_presentations(ctx, shape, pres, first=1, n=2)
    assert st_r[0] == 255 and len(st_r) == 3
    batch.issue(ctx, kinds, values, rb(3, 64), rb(3, 64), rb(3, 32), first=2, n=1)
    os.environ["AFX_FAKE_HIP_DEVICES"] = "2"
    grp = afx.Group(d["params"], d["key"], d["ip"], [0, 1])
    assert len(batch.verify_presentations(grp, shape, pres)) == 3   # a small call: one member takes it whole
    grp.member(0).set_small_batch_items(0)                            # ... and from here on every call is split over the members
    assert len(grp) == 2 and grp.member(1).n == n
    assert len(batch.verify_presentations(grp, shape, pres)) == 3
    assert [g.tolist() for g in batch.verify_mixed(grp, [(sh2, pres2), (shape, pres)])] == [[0x5a] * 3] * 2
    assert batch.issue_mixed(grp, [ia, ib])[1].tolist() == [0x5a] * 6
    assert batch.show_mixed(grp, [sa, sb])[1].tolist() == [0x5a] * 6
    assert batch.verify_issuances_mixed(grp, [dict(kinds=kinds, values=values, issuance=iss)]).tolist() == [0x5a] * 3
    assert wire.verify_wire(grp, blob).tolist() == [0x5a] * 3 and wire.verify_mixed_wire(grp, mixed_stream).tolist() == [0x5a] * 9
    assert wire.verify_wire(ctx, blob, first=1, n=2).tolist() == [255, 0x5a, 0x5a]
    try:
        wire.verify_wire(ctx, blob, first=2, n=2)
        raise SystemExit("a range outside the blob's batch was accepted")
    except afx.AfxError as ex:
        assert ex.rc == afx.E_BAD_ARGS
    o2, st2 = batch.issue(grp, kinds, values, rb(3, 64), rb(3, 64), rb(3, 32))
    assert len(st2) == 3
    assert len(batch.verify_issuances(grp, kinds, values, iss)) == 3
    assert batch.verify_issuances(ctx, kinds, values, iss, first=1, n=1)[0] == 255
    pg, shg, stg = batch.show(grp, k2, values, iss["t"], iss["U"], iss["V"], kp, rb(3, 64), rb(3, 32), rb(max(nsp, 1), 3, 32), values, values)
    assert bytes(shg) == bytes(shape) and len(stg) == 3
    pr, shr, str_ = batch.show(ctx, k2, values, iss["t"], iss["U"], iss["V"], kp, rb(3, 64), rb(3, 32), rb(max(nsp, 1), 3, 32), values, values, first=2, n_items=1)
    assert str_[0] == 255 and str_[1] == 255 and bytes(shr) == bytes(shape)
    pe, she, ste = batch.show(ctx, k2, values, iss["t"], iss["U"], iss["V"], kp, rb(3, 64), rb(3, 32), rb(max(nsp, 1), 3, 32), values, values, first=3, n_items=0)
    assert bytes(she) == bytes(shape) and (ste == 255).all()
    try:
        afx.Group(d["params"], d["key"], d["ip"], [0, 2])
        raise SystemExit("group on a missing device accepted")
    except afx.AfxError as ex:
        assert ex.rc == afx.E_NO_DEVICE and b"member 1" in afx.lib().afx_last_error()
    grp.close()
    del os.environ["AFX_FAKE_HIP_DEVICES"]
    ctx.close()
# several slices on alternating lanes (the host-pointer pipeline): 3 slices of 256 items + a tail
d = make_credentials(4, "SSPE", 1, b"hostsim-pipe")
ctx = afx.Context(d["params"], d["key"], d["ip"])
ctx.set_chunk_items(256)
cntp = 3 * 256 + 17
zero = lambda *s: np.zeros(s, np.uint8)
shape = afx.Shape()
shape.n_attributes, shape.n_responses, shape.n_hidden_scalars, shape.n_enc_proofs = 4, 4, 1, 1
for i, k in enumerate((1, 0, 2, 3)):
    shape.kinds[i] = k
shape.hidden_scalar_indices[0], shape.enc_indices[0] = 0, 3
bigp = {"challenge": zero(cntp, 32), "responses": zero(4, cntp, 32), "C_x_0": zero(cntp, 32), "C_x_1": zero(cntp, 32), "C_V": zero(cntp, 32),
        "C_y": zero(4, cntp, 32), "attr_values": zero(4, cntp, 32),
        "enc": [{f: (zero(6, cntp, 32) if f == "responses" else zero(cntp, 32)) for f in batch.ENC_FIELDS}]}
# a *_dev call with a null array that the shape needs is refused on the host (it would fault the GPU)
null_soa, stn = afx.PresentationSoA(), np.zeros(4, np.uint8)
assert afx.lib().afx_verify_presentations_dev(ctx.h, C.byref(shape), C.byref(null_soa), 4, stn.ctypes.data) == afx.E_BAD_ARGS
null_enc = afx.EncProofSoA()
assert afx.lib().afx_verify_encryption_proofs_dev(ctx.h, 3, C.byref(null_enc), 4, stn.ctypes.data) == afx.E_BAD_ARGS
stp = batch.verify_presentations(ctx, shape, bigp)
assert len(stp) == cntp and (stp == 0x5a).all(), stp[:8]      # every status byte came back through the pinned path
stp = batch.verify_presentations(ctx, shape, bigp, first=300, n=400)
assert (stp[:300] == 255).all() and (stp[300:700] == 0x5a).all() and (stp[700:] == 255).all()
ctx.close()
# a device too small for the default pass: the engine halves the pass size until the workspace fits
d = make_credentials(4, "SSPE", 1, b"hostsim-oom")
ctx = afx.Context(d["params"], d["key"], d["ip"])
cnt = 20000
zero = lambda *s: np.zeros(s, np.uint8)
shape = afx.Shape()
shape.n_attributes, shape.n_responses, shape.n_hidden_scalars, shape.n_enc_proofs = 4, 4, 1, 1
for i, k in enumerate((1, 0, 2, 3)):
    shape.kinds[i] = k
shape.hidden_scalar_indices[0], shape.enc_indices[0] = 0, 3
big = {"challenge": zero(cnt, 32), "responses": zero(4, cnt, 32), "C_x_0": zero(cnt, 32), "C_x_1": zero(cnt, 32), "C_V": zero(cnt, 32),
       "C_y": zero(4, cnt, 32), "attr_values": zero(4, cnt, 32),
       "enc": [{f: (zero(6, cnt, 32) if f == "responses" else zero(cnt, 32)) for f in batch.ENC_FIELDS}]}
os.environ["AFX_FAKE_HIP_MAX_ALLOC"] = str(256 << 20)
assert len(batch.verify_presentations(ctx, shape, big)) == cnt
os.environ["AFX_FAKE_HIP_MAX_ALLOC"] = str(8 << 20)      # not even the smallest pass fits: a clean error
try:
    batch.verify_presentations(ctx, shape, big)
    raise SystemExit("out-of-memory not reported")
except afx.AfxError as e:
    assert e.rc == afx.E_HIP, e.rc
del os.environ["AFX_FAKE_HIP_MAX_ALLOC"]
# a scripted merlin transcript (afx_merlin_challenges): compiled like the statements' own, fields staged; malformed scripts refused
mf = np.arange(32 * 5, dtype=np.uint8).reshape(5, 32)
mo = ctx.merlin_challenges(b"hostsim", [("append", b"big", bytes(1000)), ("append_field", b"f", 0), ("append", b"", b""), ("append_field", b"g", 1), ("challenge", b"c", 32)], [mf, mf], 5)
assert mo.shape == (5, 64)
for bad_script in (b"", bytes([1]), bytes([1, 9, 0, 0, 0]) + b"xy", bytes([1, 1, 0, 0, 0]) + b"x", bytes([1, 1, 0, 0, 0]) + b"x" + bytes([7, 0, 0, 0, 0]),
                   bytes([1, 1, 0, 0, 0]) + b"x" + bytes([4, 1, 0, 0, 0]) + b"c" + bytes([0, 0, 0, 0]),
                   bytes([1, 1, 0, 0, 0]) + b"x" + bytes([4, 1, 0, 0, 0]) + b"c" + bytes([8, 0, 0, 0]) + bytes([2, 0, 0, 0, 0, 0, 0, 0, 0])):
    assert afx.lib().afx_merlin_challenges(ctx.h, bad_script, len(bad_script), None, 0, 1, mo.ctypes.data) == afx.E_BAD_ARGS, bad_script
ctx.close()
# bad parameters / keys are rejected on the host
d = make_credentials(2, "SP", 1, b"hostsim-bad")
for params, key in ((d["params"][:-1], d["key"]), (d["params"], d["key"][:-1]), (b"\x09" + d["params"][1:], d["key"]), (d["params"], d["key"][:4] + b"\xff" * 32 + d["key"][36:])):
    try:
        afx.Context(params, key, d["ip"])
        raise SystemExit("accepted bad parameters")
    except afx.AfxError as e:
        assert e.rc == afx.E_BAD_PARAMS, e.rc
print("hostsim ok")
"""


# the plans of small prover passes come in three forms (engine.cpp Assembler::segments, msm_list): segments over kept CACHED tables (the
# default), segments over kept affine tables (no four-wave chains: AFX_VARIANT_ONE_WAVE_CHAINS | AFX_VARIANT_SEGMENTS_4), and whole
# chains (AFX_VARIANT_SEGMENTS_1) - afx_ctx_set_plan_variants, applied to every context the script makes by the python mirror's
# AFX_TEST_PLAN_VARIANTS hook (aeonflux_amd.DEFAULT_PLAN_VARIANTS)
@pytest.mark.parametrize("plan_env", [{}, {"AFX_TEST_PLAN_VARIANTS": "0x0c"}, {"AFX_TEST_PLAN_VARIANTS": "0x01"}], ids=["default", "affine-segments", "whole-chains"])
def test_every_entry_point_assembles_cleanly_under_asan(hostsim_lib, tmp_path, plan_env):
    script = tmp_path / "drive.py"
    script.write_text(DRIVER % {"root": ROOT, "lib": hostsim_lib})
    asan = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    # AFX_PLAN_SELFCHECK: every plan is assembled twice against different provisional addresses and the two copies, relocated to
    # the same place, must be byte-identical; a reused plan must equal a freshly assembled one (engine.hpp afx::Plan)
    env = dict(os.environ, LD_PRELOAD=asan, ASAN_OPTIONS="detect_leaks=0", AFX_PLAN_SELFCHECK="1", **plan_env)
    r = subprocess.run([sys.executable, str(script)], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "hostsim ok" in r.stdout, (r.stdout[-2000:], r.stderr[-3000:])


TSAN_DRIVER = r"""
import os, sys
sys.path.insert(0, %(root)r)
import numpy as np
import aeonflux_amd as afx
afx.LIB_PATH = %(lib)r
from aeonflux_amd import batch
import bench
os.environ["AFX_FAKE_HIP_DEVICES"] = "3"
params, key, ip = bench.load_fixture("c3_8attrs_SSPPeeee")
grp = afx.Group(params, key, ip, [0, 1, 2])
grp.member(0).set_small_batch_items(0)   # 1500-item calls: split over the members' threads (what this run is about)
shape = afx.Shape()
shape.n_attributes, shape.n_responses, shape.n_hidden_scalars, shape.n_enc_proofs = 8, 3, 0, 4
for i, k in enumerate((0, 0, 2, 2, 3, 3, 3, 3)):
    shape.kinds[i] = k
for e in range(4):
    shape.enc_indices[e] = 4 + e
cnt = 1500
z = lambda *s: np.zeros(s, np.uint8)
pres = {"challenge": z(cnt, 32), "responses": z(3, cnt, 32), "C_x_0": z(cnt, 32), "C_x_1": z(cnt, 32), "C_V": z(cnt, 32), "C_y": z(8, cnt, 32),
        "attr_values": z(8, cnt, 32), "enc": [{f: (z(6, cnt, 32) if f == "responses" else z(cnt, 32)) for f in batch.ENC_FIELDS} for _ in range(4)]}
for m in range(3):
    grp.member(m).set_chunk_items(256)
for _ in range(3):
    assert (batch.verify_presentations(grp, shape, pres) == 0x5a).all()
kinds = [0] * 8
o, st = batch.issue(grp, kinds, z(8, cnt, 32), z(cnt, 64), z(cnt, 64), z(cnt, 32))
assert len(st) == cnt
assert len(batch.verify_issuances(grp, kinds, z(8, cnt, 32), o)) == cnt
kp = {f: z(cnt, 32) for f in ("a", "a0", "a1", "pk")}
k2 = [1, 0, 2, 2, 4, 4, 3, 3]
pg, shg, stg = batch.show(grp, k2, z(8, cnt, 32), o["t"], o["U"], o["V"], kp, z(cnt, 64), z(cnt, 32), z(2, cnt, 32), z(8, cnt, 32), z(8, cnt, 32))
assert len(stg) == cnt and shg.n_enc_proofs == 2
# serialized batches and a stream of mixed shapes over the members' threads
from aeonflux_amd import wire
blob = wire.pack_presentations(shape, pres)
assert (wire.verify_wire(grp, blob) == 0x5a).all()
sh2 = afx.Shape.from_buffer_copy(bytes(shape))
sh2.n_enc_proofs = 0
half = {f: np.ascontiguousarray(pres[f][..., :700, :]) for f in batch.PRES_FIELDS}
assert (wire.verify_mixed_wire(grp, wire.pack_presentations(sh2, dict(half, enc=[])) + blob + wire.pack_presentations(sh2, dict(half, enc=[]))) == 0x5a).all()
assert all((g == 0x5a).all() for g in batch.verify_mixed(grp, [(shape, pres), (sh2, dict(half, enc=[]))]))
grp.close()
# eight members (what a node has): ragged splits (1501 items over 8), a request of mixed layouts dealt out to the members' threads,
# and a member that fails in the middle of a call: the call reports that member and the others finish
os.environ["AFX_FAKE_HIP_DEVICES"] = "8"
grp8 = afx.Group(params, key, ip, list(range(8)))
for m in range(8):
    grp8.member(m).set_chunk_items(256)
rag = 1501
pr = {f: np.ascontiguousarray(np.concatenate([pres[f], pres[f][..., :1, :]], axis=-2)) for f in batch.PRES_FIELDS}
pr["enc"] = [{f: np.ascontiguousarray(np.concatenate([d[f], d[f][..., :1, :]], axis=-2)) for f in batch.ENC_FIELDS} for d in pres["enc"]]
grp8.member(0).set_small_batch_items(0)
for m in range(1, 8):
    grp8.member(m).set_small_batch_items(0)
assert (batch.verify_presentations(grp8, shape, pr) == 0x5a).all() and len(batch.verify_presentations(grp8, shape, pr)) == rag
assert len(batch.verify_presentations(grp8, shape, {f: (pr[f][..., :5, :] if f != "enc" else None) for f in batch.PRES_FIELDS} | {"enc": [{f: d[f][..., :5, :] for f in batch.ENC_FIELDS} for d in pr["enc"]]})) == 5   # fewer items than members
for m in range(8):
    grp8.member(m).set_small_batch_items(4096)
small = lambda p, n: dict({f: np.ascontiguousarray(p[f][..., :n, :]) for f in batch.PRES_FIELDS}, enc=[{f: np.ascontiguousarray(d[f][..., :n, :]) for f in batch.ENC_FIELDS} for d in p["enc"]])
items = []
for g in range(20):
    items.append((shape, small(pres, 3 + g)) if g %% 2 == 0 else (sh2, dict(small(pres, 2 + g), enc=[])))
assert all((g == 0x5a).all() for g in batch.verify_mixed(grp8, items))
ia = [dict(kinds=kinds, values=z(8, 4 + g, 32), t_wide=z(4 + g, 64), U_wide=z(4 + g, 64), rng_seed=z(4 + g, 32)) for g in range(11)]
outs8, st8 = batch.issue_mixed(grp8, ia)
assert (st8 == 0x5a).all() and len(st8) == sum(4 + g for g in range(11))
os.environ["AFX_FAKE_HIP_FAIL_DEVICE"] = "5"
for m in range(8):
    grp8.member(m).set_small_batch_items(0)
try:
    batch.verify_presentations(grp8, shape, pr)
    raise SystemExit("a failing member went unnoticed")
except afx.AfxError as e:
    assert e.rc == afx.E_HIP and "member 5" in str(e), str(e)
del os.environ["AFX_FAKE_HIP_FAIL_DEVICE"]
assert (batch.verify_presentations(grp8, shape, pr) == 0x5a).all()   # and the group is usable afterwards
grp8.close()
print("tsan drive ok")
"""


def test_group_calls_are_race_free_under_tsan(tmp_path):
    """the multi-device path (one host thread per member, slices on two lanes each) under ThreadSanitizer, on the fake runtime"""
    tsan = subprocess.run(["gcc", "-print-file-name=libtsan.so"], capture_output=True, text=True).stdout.strip()
    if not os.path.isabs(tsan) or not os.path.exists(tsan):
        pytest.skip("no libtsan")
    out = str(tmp_path / "libafx_tsan.so")
    srcs = [os.path.join(CSRC, f) for f in ("engine.cpp", "plans.cpp", "statements.cpp", "statements_prove.cpp", "statements_setup.cpp", "group.cpp", "mixed.cpp", "wire.cpp")]
    srcs.append(os.path.join(ROOT, "tests", "hostsim", "fake_hip.cpp"))
    r = subprocess.run(["g++", "-g", "-O1", "-fsanitize=thread", "-fPIC", "-std=c++17", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", "-shared",
                        "-pthread", "-o", out] + srcs, capture_output=True, text=True)
    if r.returncode != 0:
        pytest.skip("cannot build with -fsanitize=thread: " + r.stderr[-300:])
    script = tmp_path / "drive.py"
    script.write_text(TSAN_DRIVER % {"root": ROOT, "lib": out})
    env = dict(os.environ, LD_PRELOAD=tsan, TSAN_OPTIONS="report_bugs=1 halt_on_error=0 exitcode=0")
    r = subprocess.run([sys.executable, str(script)], env=env, capture_output=True, text=True, timeout=900)
    assert "tsan drive ok" in r.stdout, (r.stdout[-1000:], r.stderr[-3000:])
    assert "WARNING: ThreadSanitizer" not in r.stderr, r.stderr[-4000:]


CO_DRIVER = r"""
# Concurrent small calls on ONE context (afx_ctx::co, plans.cpp coalesced_call): threads of every front end at once, calls that
# take the context alone in between, and launch sets that fail with calls of several threads in them.  The fake runtime makes a
# launch set "compute" for 300 us (so that calls arrive meanwhile and are collected) and echoes the first byte of every item's
# challenge (issuances: t) as its status: every caller must get the answer to ITS rows.
import os, sys, threading, ctypes as C
sys.path.insert(0, %(root)r)
import numpy as np
import aeonflux_amd as afx
afx.LIB_PATH = %(lib)r
from aeonflux_amd import batch, wire
import bench
L = afx.lib()
L.afx_fake_set.argtypes = [C.c_char_p, C.c_int]
params, key, ip = bench.load_fixture("c3_8attrs_SSPPeeee")
ctx = afx.Context(params, key, ip)
shape = afx.Shape()
shape.n_attributes, shape.n_responses, shape.n_hidden_scalars, shape.n_enc_proofs = 8, 3, 0, 4
for i, k in enumerate((0, 0, 2, 2, 3, 3, 3, 3)):
    shape.kinds[i] = k
for e in range(4):
    shape.enc_indices[e] = 4 + e
sh2 = afx.Shape.from_buffer_copy(bytes(shape))
sh2.n_enc_proofs = 0
z = lambda *s: np.zeros(s, np.uint8)
def mk(cnt, tag, enc=4):
    p = {"challenge": z(cnt, 32), "responses": z(3, cnt, 32), "C_x_0": z(cnt, 32), "C_x_1": z(cnt, 32), "C_V": z(cnt, 32), "C_y": z(8, cnt, 32),
         "attr_values": z(8, cnt, 32), "enc": [{f: (z(6, cnt, 32) if f == "responses" else z(cnt, 32)) for f in batch.ENC_FIELDS} for _ in range(enc)]}
    p["challenge"][:, 0] = [(tag + i) %% 251 for i in range(cnt)]
    return p
kinds = [0] * 8
k2 = [1, 0, 2, 2, 4, 4, 3, 3]
L.afx_fake_set(b"echo", 1)
L.afx_fake_set(b"sync_us", 300)
K, reps = 16, 24
errs, failed = [], []
def work(t):
    try:
        for r in range(reps):
            cnt, tag = 1 + (t + r) %% 5, 7 * t + r
            want = [(tag + i) %% 251 for i in range(cnt)]
            try:
                kind = t %% 8
                if kind == 6:      # a request of two shapes: its groups join the collecting sessions (afx::DeferScope) beside the others' calls
                    a, b = mk(cnt, tag), mk(2, tag + 100, 0)
                    got = batch.verify_mixed(ctx, [(shape, a), (sh2, b), (shape, a)])
                    st = np.concatenate(got)
                    want = want + [(tag + 100 + i) %% 251 for i in range(2)] + want
                elif kind == 7:    # ... and as a stream of serialized sections, the two of one shape apart from each other
                    a, b = mk(cnt, tag), mk(2, tag + 100, 0)
                    st = wire.verify_mixed_wire(ctx, wire.pack_presentations(shape, a) + wire.pack_presentations(sh2, b) + wire.pack_presentations(shape, a))
                    want = want + [(tag + 100 + i) %% 251 for i in range(2)] + want
                elif kind == 0:
                    st = batch.verify_presentations(ctx, shape, mk(cnt, tag))
                elif kind == 1:
                    st = batch.verify_presentations(ctx, sh2, mk(cnt, tag, 0))
                elif kind == 2:
                    st = wire.verify_wire(ctx, wire.pack_presentations(shape, mk(cnt, tag)))
                elif kind == 3:
                    o, st = batch.issue(ctx, kinds, z(8, cnt, 32), z(cnt, 64), z(cnt, 64), z(cnt, 32))
                    want = [0] * cnt        # (zero inputs: the echoed byte of the first scalar array the pass checks)
                elif kind == 4:
                    iss = {"t": z(cnt, 32), "U": z(cnt, 32), "V": z(cnt, 32), "challenge": z(cnt, 32), "responses": z(13, cnt, 32)}
                    st = batch.verify_issuances(ctx, kinds, z(8, cnt, 32), iss) if r %% 2 else ctx.verify_issuances_wire(wire.pack_issuances(kinds, z(8, cnt, 32), iss))
                    want = [0x5a] * cnt     # (the fake runtime's negated generators encode as zeros: the statement fails whole, before any check is laid out)
                else:
                    kp = {f: z(cnt, 32) for f in ("a", "a0", "a1", "pk")}
                    pg, shg, st = batch.show(ctx, k2, z(8, cnt, 32), z(cnt, 32), z(cnt, 32), z(cnt, 32), kp, z(cnt, 64), z(cnt, 32), z(2, cnt, 32), z(8, cnt, 32), z(8, cnt, 32))
                    want = [0] * cnt
                    assert shg.n_enc_proofs == 2
                if st.tolist() != want:
                    errs.append((t, r, st.tolist(), want))
            except afx.AfxError as e:
                if e.rc != afx.E_HIP:
                    errs.append((t, r, repr(e)))
                failed.append((t, r))
    except Exception as e:
        errs.append((t, repr(e)))
def alone():
    # calls that take the context for themselves while the others keep coming: a batch too large to be collected, setters, readers
    try:
        for r in range(6):
            big = mk(600, r)
            st = batch.verify_presentations(ctx, shape, big)
            assert st.tolist() == [(r + i) %% 251 for i in range(600)]
            ctx.set_small_batch_items(4096)
            ctx.plan_stats()
            assert len(batch.multiscalar_mul(ctx, z(2, 3, 32), z(2, 3, 32))[0]) == 3
            if r == 2:
                L.afx_fake_set(b"fail_next", 3)   # the next three launch sets fail, whoever is in them
    except afx.AfxError as e:
        if e.rc != afx.E_HIP:
            errs.append(("alone", repr(e)))
    except Exception as e:
        errs.append(("alone", repr(e)))
ths = [threading.Thread(target=work, args=(t,)) for t in range(K)] + [threading.Thread(target=alone)]
[t.start() for t in ths]
[t.join() for t in ths]
cs = ctx.coalescing_stats()
assert not errs, errs[:3]
assert cs["calls"] >= K * reps - 8 and cs["sessions"] < cs["calls"] and cs["appended_calls"] > 0 and cs["max_calls"] > 1, cs
assert 1 <= len(failed) <= 3 * K, failed          # the failed launch sets took their callers with them - and only those
# the collector's other ways out of a session: it holds all it may (max_items), its images are full (calls of hundreds of items), the
# bounded wait runs out while an earlier session still computes, a call too large for the item slots on offer opens a group of its own
def phase(threads, rounds, items_of):
    errs2 = []
    def w(t):
        try:
            for r in range(rounds):
                cnt, tag = items_of(t, r), 11 * t + r
                st = batch.verify_presentations(ctx, shape if t %% 2 else sh2, mk(cnt, tag, 4 if t %% 2 else 0))
                if st.tolist() != [(tag + i) %% 251 for i in range(cnt)]:
                    errs2.append((t, r, cnt, st.tolist()[:4]))
        except Exception as e:
            errs2.append((t, repr(e)))
    ths2 = [threading.Thread(target=w, args=(t,)) for t in range(threads)]
    [t.start() for t in ths2]
    [t.join() for t in ths2]
    assert not errs2, errs2[:3]
before = ctx.coalescing_stats()
ctx.set_coalescing(2000, 8)                 # eight items and the session goes, whatever is in flight
phase(10, 12, lambda t, r: 1 + (t + r) %% 3)
mid = ctx.coalescing_stats()
assert mid["sessions"] - before["sessions"] >= (mid["items"] - before["items"]) // 12, (before, mid)
ctx.set_coalescing(1, 4096)                 # a wait bound of a microsecond: nobody lingers behind the 300 us "kernels"
phase(10, 8, lambda t, r: 1 + (t + r) %% 3)
ctx.set_coalescing(2000, 4096)
phase(12, 4, lambda t, r: 300 + 17 * ((t + r) %% 5))      # ~1 MB of rows a call: the 8 MB images fill up
phase(10, 6, lambda t, r: (1, 70, 2, 130, 5)[(t + r) %% 5])   # calls beyond the 64 item slots a group starts with
# the context is usable afterwards, collected or alone
last = ctx.coalescing_stats()["calls"]
assert batch.verify_presentations(ctx, shape, mk(3, 9)).tolist() == [9, 10, 11]
ctx.set_coalescing(0, 0)
assert batch.verify_presentations(ctx, shape, mk(3, 9)).tolist() == [9, 10, 11]
assert ctx.coalescing_stats()["calls"] == last + 1   # switched off: the second call ran alone
pc = ctx.plan_cache_stats()
assert pc["hits"] > 0 and pc["misses"] > 0 and pc["entries"] > 0 and pc["bytes"] > 0, pc
ctx.close()
print("coalescing drive ok", cs, len(failed))
"""


def _build_hostsim(out, flags):
    srcs = [os.path.join(CSRC, f) for f in ("engine.cpp", "plans.cpp", "statements.cpp", "statements_prove.cpp", "statements_setup.cpp", "group.cpp", "mixed.cpp", "wire.cpp")]
    srcs.append(os.path.join(ROOT, "tests", "hostsim", "fake_hip.cpp"))
    return subprocess.run(["g++", "-g", "-O1"] + flags + ["-fPIC", "-std=c++17", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", "-shared", "-pthread", "-o", out] + srcs,
                          capture_output=True, text=True)


def test_concurrent_calls_on_one_context_under_tsan(tmp_path):
    """join / flush / a failing flush / calls that need the context alone, from 13 threads on one context, under ThreadSanitizer"""
    tsan = subprocess.run(["gcc", "-print-file-name=libtsan.so"], capture_output=True, text=True).stdout.strip()
    if not os.path.isabs(tsan) or not os.path.exists(tsan):
        pytest.skip("no libtsan")
    out = str(tmp_path / "libafx_tsan.so")
    r = _build_hostsim(out, ["-fsanitize=thread"])
    if r.returncode != 0:
        pytest.skip("cannot build with -fsanitize=thread: " + r.stderr[-300:])
    script = tmp_path / "drive.py"
    script.write_text(CO_DRIVER % {"root": ROOT, "lib": out})
    env = dict(os.environ, LD_PRELOAD=tsan, TSAN_OPTIONS="report_bugs=1 halt_on_error=0 exitcode=0")
    r = subprocess.run([sys.executable, str(script)], env=env, capture_output=True, text=True, timeout=900)
    assert "coalescing drive ok" in r.stdout, (r.stdout[-1000:], r.stderr[-3000:])
    assert "WARNING: ThreadSanitizer" not in r.stderr, r.stderr[-4000:]
    # ... and the copy pool of large host-pointer calls (the same build)
    script.write_text(POOL_DRIVER % {"root": ROOT, "lib": out})
    r = subprocess.run([sys.executable, str(script)], env=env, capture_output=True, text=True, timeout=900)
    assert "pool drive ok" in r.stdout, (r.stdout[-1000:], r.stderr[-3000:])
    assert "WARNING: ThreadSanitizer" not in r.stderr, r.stderr[-4000:]


def test_concurrent_calls_on_one_context_under_asan(hostsim_lib, tmp_path):
    """the same drive under ASan/UBSan with the plan self-check on: joined calls write only the item slots they were given"""
    script = tmp_path / "drive.py"
    script.write_text(CO_DRIVER % {"root": ROOT, "lib": hostsim_lib})
    asan = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    env = dict(os.environ, LD_PRELOAD=asan, ASAN_OPTIONS="detect_leaks=0", AFX_PLAN_SELFCHECK="1")
    r = subprocess.run([sys.executable, str(script)], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "coalescing drive ok" in r.stdout, (r.stdout[-2000:], r.stderr[-3000:])


POOL_DRIVER = r"""
# Large host-pointer calls: slices of more than 16 MB of rows are gathered into the lane's pinned image by the context's copy pool
# (plans.cpp Stager::upload, afx::CopyPool) and their results scattered by it.  The fake runtime echoes the first byte of every
# item's challenge as its status, so every row must land where the plan reads it - with 0 (the runtime's copies), 1, 4 threads, on
# column arrays, on a sub-range, on a serialized batch (whose staging area has a scratch hole between its runs), and on issue
# (800 bytes of results per item come back through the pool).
import os, sys, ctypes as C
sys.path.insert(0, %(root)r)
import numpy as np
import aeonflux_amd as afx
afx.LIB_PATH = %(lib)r
from aeonflux_amd import batch, wire
import bench
L = afx.lib()
L.afx_fake_set.argtypes = [C.c_char_p, C.c_int]
L.afx_fake_set(b"echo", 1)
params, key, ip = bench.load_fixture("c3_8attrs_SSPPeeee")
ctx = afx.Context(params, key, ip)
ctx.set_chunk_items(32768)          # slices of 32768 items x 704 bytes = 23 MB
shape = afx.Shape()
shape.n_attributes, shape.n_responses, shape.n_hidden_scalars, shape.n_enc_proofs = 8, 3, 0, 0
for i, k in enumerate((0, 0, 2, 2, 2, 2, 2, 2)):
    shape.kinds[i] = k
cnt = 70001
rng = np.random.default_rng(5)
r = lambda *s: rng.integers(0, 256, size=s, dtype=np.uint8)
pres = {"challenge": r(cnt, 32), "responses": r(3, cnt, 32), "C_x_0": r(cnt, 32), "C_x_1": r(cnt, 32), "C_V": r(cnt, 32), "C_y": r(8, cnt, 32), "attr_values": r(8, cnt, 32), "enc": []}
want = pres["challenge"][:, 0].copy()
blob = wire.pack_presentations(shape, pres)
before = os.sched_getaffinity(0)
for threads in (4, 0, 1, 3):
    ctx.set_host_copy_threads(threads)
    for _ in range(2):
        assert np.array_equal(batch.verify_presentations(ctx, shape, pres), want), threads
    soa, keep = batch.presentation_soa(pres)
    st = np.full(cnt, 0xEE, np.uint8)
    afx.check(L.afx_verify_presentations_range(ctx.h, C.byref(shape), C.byref(soa), cnt, 1234, 40000, st.ctypes.data))
    assert np.array_equal(st[1234:41234], want[1234:41234]) and (st[:1234] == 0xEE).all() and (st[41234:] == 0xEE).all(), threads
    assert np.array_equal(wire.verify_wire(ctx, blob), want), threads
    assert os.sched_getaffinity(0) == before, "the caller's mask was not restored"
# issue: the fake kernels write nothing, so the outputs are the zeros the staging area starts from - what matters is that
# 100 MB of result rows come back through the pool without touching memory that is not theirs
kinds = [0] * 8
ctx.set_host_copy_threads(4)
o, st = batch.issue(ctx, kinds, r(8, cnt, 32), r(cnt, 64), r(cnt, 64), r(cnt, 32))
assert len(st) == cnt and o["responses"].shape == (13, cnt, 32) and not o["t"].any()
ctx.close()
print("pool drive ok")
"""


def test_large_host_calls_through_the_copy_pool_under_asan(hostsim_lib, tmp_path):
    script = tmp_path / "drive.py"
    script.write_text(POOL_DRIVER % {"root": ROOT, "lib": hostsim_lib})
    asan = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    env = dict(os.environ, LD_PRELOAD=asan, ASAN_OPTIONS="detect_leaks=0")
    r = subprocess.run([sys.executable, str(script)], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "pool drive ok" in r.stdout, (r.stdout[-2000:], r.stderr[-3000:])


ADVICE_DRIVER = r"""
# Round-5 advisor findings on the collector (plans.cpp coalesced_call, Session::launch), on the fake runtime with echoed statuses:
# (1) a mixed request of many shapes stages its groups one after the other (0.3 ms of plan assembly per new shape, several ms under
#     the sanitizer) and used to LEAD the session it opened until its last group was staged: another thread's single call that had
#     joined that session slept the whole time.  Now the first ordinary caller that joins takes the session over, and a deferring leader
#     launches a session whose deadline has passed on its next way through.
# (2) a launch set whose plans do not fit the device side by side failed every call it carried; now it runs in halves.
import os, sys, time, threading, ctypes as C
sys.path.insert(0, %(root)r)
import numpy as np
import aeonflux_amd as afx
afx.LIB_PATH = %(lib)r
from aeonflux_amd import batch
import bench
L = afx.lib()
L.afx_fake_set.argtypes = [C.c_char_p, C.c_int]
L.afx_fake_set(b"echo", 1)
params, key, ip = bench.load_fixture("c3_8attrs_SSPPeeee")
ctx = afx.Context(params, key, ip)
z = lambda *s: np.zeros(s, np.uint8)
def shape_of(pub, enc):
    # (pub: which of the four revealed attributes are points rather than scalars; enc: how many proofs of encryption are attached)
    sh = afx.Shape()
    sh.n_attributes, sh.n_responses, sh.n_hidden_scalars, sh.n_enc_proofs = 8, 3, 0, enc
    for i in range(4):
        sh.kinds[i] = 2 if (pub >> i) & 1 else 0
    for i in range(4, 8):
        sh.kinds[i] = 3
    for e in range(enc):
        sh.enc_indices[e] = 4 + e
    return sh
def mk(cnt, tag, enc=4):
    p = {"challenge": z(cnt, 32), "responses": z(3, cnt, 32), "C_x_0": z(cnt, 32), "C_x_1": z(cnt, 32), "C_V": z(cnt, 32), "C_y": z(8, cnt, 32),
         "attr_values": z(8, cnt, 32), "enc": [{f: (z(6, cnt, 32) if f == "responses" else z(cnt, 32)) for f in batch.ENC_FIELDS} for _ in range(enc)]}
    p["challenge"][:, 0] = [(tag + i) %% 251 for i in range(cnt)]
    return p
# ---- (1) many DIFFERENT shapes in one request (distinct response counts / proofs of encryption: each is a plan of its own, none cached)
def request(tag0):
    return [(shape_of(g %% 16, g %% 5), mk(3, tag0 + g, g %% 5)) for g in range(48)]    # (g -> (g mod 16, g mod 5) is one-to-one below 80)
single_shape, single = shape_of(12, 4), mk(2, 200)
batch.verify_presentations(ctx, single_shape, single)          # (the single call's plan is kept from here on)
ctx.set_coalescing(200000, 4096)                               # a long max_wait_us: only the hand-over can make the single call fast
L.afx_fake_set(b"sync_us", 2000)
t_req, t_single, errs = [0.0], [], []
def run_request():
    t0 = time.perf_counter()
    got = batch.verify_mixed(ctx, request(0))
    t_req[0] = time.perf_counter() - t0
    for g, st in enumerate(got):
        if st.tolist() != [(g + i) %% 251 for i in range(3)]:
            errs.append(("request", g, st.tolist()))
def run_singles():
    time.sleep(0.02)                                            # the request has opened its session and is assembling
    for _ in range(3):
        t0 = time.perf_counter()
        st = batch.verify_presentations(ctx, single_shape, single)
        t_single.append(time.perf_counter() - t0)
        if st.tolist() != [200, 201]:
            errs.append(("single", st.tolist()))
a, b = threading.Thread(target=run_request), threading.Thread(target=run_singles)
a.start(); b.start(); a.join(); b.join()
assert not errs, errs[:3]
assert t_req[0] > 0.1, t_req                                    # (the request really is long: 48 plans under the sanitizer)
assert min(t_single) < 0.35 * t_req[0], (t_single, t_req)       # the single calls did not wait for the request to finish staging
cs = ctx.coalescing_stats()
# ... and alone, past the deadline: a deferring leader launches on its way through instead of at the end
ctx.set_coalescing(1000, 4096)
s0 = ctx.coalescing_stats()["sessions"]
got = batch.verify_mixed(ctx, request(50))
assert all(st.tolist() == [(50 + g + i) %% 251 for i in range(3)] for g, st in enumerate(got))
assert ctx.coalescing_stats()["sessions"] - s0 >= 2, "a 48-shape request that outlives max_wait_us goes out in more than one launch set"
# ---- (2) a launch set that does not fit the device side by side: in halves
L.afx_fake_set(b"sync_us", 0)
ctx.set_coalescing(0, 0)                                        # collection off: the request runs its own session of merged plans
items = [(shape_of(g, 4), mk(64, 10 * g, 4)) for g in range(8)]
got = batch.verify_mixed(ctx, items)                            # (sizes the buffers once without a limit)
ctx2 = afx.Context(params, key, ip)
ctx2.set_coalescing(0, 0)
lo = None
for mb in (4, 6, 8, 12, 16, 24, 32, 48, 64, 96, 128):          # the smallest limit ONE such plan runs under
    os.environ["AFX_FAKE_HIP_MAX_ALLOC"] = str(mb << 20)
    try:
        batch.verify_mixed(ctx2, items[:1])
        lo = mb
        break
    except afx.AfxError as e:
        assert e.rc == afx.E_HIP, e.rc
assert lo is not None
ctx2.close()
del os.environ["AFX_FAKE_HIP_MAX_ALLOC"]
ctx3 = afx.Context(params, key, ip)                             # a fresh context: no buffer is large already
ctx3.set_coalescing(0, 0)
os.environ["AFX_FAKE_HIP_MAX_ALLOC"] = str((2 * lo) << 20)      # two plans fit side by side, eight do not
got = batch.verify_mixed(ctx3, items)
del os.environ["AFX_FAKE_HIP_MAX_ALLOC"]
for g, st in enumerate(got):
    assert st.tolist() == [(10 * g + i) %% 251 for i in range(64)], (g, st.tolist()[:8])
ctx3.close()
ctx.close()
print("advice drive ok", round(t_req[0], 3), [round(x, 3) for x in t_single], lo)
"""


def test_round5_advisor_findings_on_the_collector(hostsim_lib, tmp_path):
    script = tmp_path / "drive.py"
    script.write_text(ADVICE_DRIVER % {"root": ROOT, "lib": hostsim_lib})
    asan = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    env = dict(os.environ, LD_PRELOAD=asan, ASAN_OPTIONS="detect_leaks=0", AFX_PLAN_SELFCHECK="1")
    r = subprocess.run([sys.executable, str(script)], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "advice drive ok" in r.stdout, (r.stdout[-2000:], r.stderr[-3000:])


NUMA_DRIVER = r"""
# The host threads of a group's members run on the CPUs of their device's NUMA node (group.cpp cpus_of_device / PinScope).  The fake
# runtime names device d "0000:0d:00.0"; AFX_SYSFS_ROOT points at a two-socket tree: devices 0, 1 on node 0 (CPUs 0-1), devices 2, 3 on
# node 1 (CPUs 2-3).  The fake finishing kernel records the affinity of the thread that launched it.
import os, sys, ctypes as C
sys.path.insert(0, %(root)r)
import numpy as np
import aeonflux_amd as afx
afx.LIB_PATH = %(lib)r
from aeonflux_amd import batch
import bench
tree = %(tree)r
allowed = sorted(os.sched_getaffinity(0))
assert len(allowed) >= 4, allowed
node_cpus = [allowed[:2], allowed[2:4]]
for d in range(4):
    p = os.path.join(tree, "sys/bus/pci/devices/0000:%%02x:00.0" %% d)
    os.makedirs(p)
    open(os.path.join(p, "numa_node"), "w").write("%%d\n" %% (d // 2))
for node, cpus in enumerate(node_cpus):
    p = os.path.join(tree, "sys/devices/system/node/node%%d" %% node)
    os.makedirs(p)
    open(os.path.join(p, "cpulist"), "w").write(",".join(str(c) for c in cpus) + ",4000\n")   # (a CPU this process may not use is dropped)
L = afx.lib()
L.afx_fake_affinity.restype = C.c_ulonglong
L.afx_fake_affinity.argtypes = [C.c_int]
mask = lambda cpus: sum(1 << c for c in cpus if c < 64)
params, key, ip = bench.load_fixture("c3_8attrs_SSPPeeee")
shape = afx.Shape()
shape.n_attributes, shape.n_responses, shape.n_hidden_scalars, shape.n_enc_proofs = 8, 3, 0, 0
for i, k in enumerate((0, 0, 2, 2, 2, 2, 2, 2)):
    shape.kinds[i] = k
cnt = 900
z = lambda *s: np.zeros(s, np.uint8)
pres = {"challenge": z(cnt, 32), "responses": z(3, cnt, 32), "C_x_0": z(cnt, 32), "C_x_1": z(cnt, 32), "C_V": z(cnt, 32), "C_y": z(8, cnt, 32), "attr_values": z(8, cnt, 32), "enc": []}
os.environ["AFX_FAKE_HIP_DEVICES"] = "4"
def run(expect):
    grp = afx.Group(params, key, ip, [0, 1, 2, 3])
    for m in range(4):
        grp.member(m).set_small_batch_items(0)     # every call is split over the members' threads
    before = os.sched_getaffinity(0)
    assert (batch.verify_presentations(grp, shape, pres) == 0x5a).all()
    assert os.sched_getaffinity(0) == before, "the caller's mask was not restored"      # member 0 runs on the caller's thread
    got = [L.afx_fake_affinity(d) for d in range(4)]
    assert got == expect, (got, expect)
    # a small call goes whole to one member, on the caller's thread, pinned for the call and restored
    for m in range(4):
        grp.member(m).set_small_batch_items(4096)
    small = {f: (v[..., :3, :] if f != "enc" else v) for f, v in pres.items()}
    small = {f: (np.ascontiguousarray(v) if f != "enc" else v) for f, v in small.items()}
    for _ in range(4):
        assert len(batch.verify_presentations(grp, shape, small)) == 3
    assert os.sched_getaffinity(0) == before
    assert [L.afx_fake_affinity(d) for d in range(4)] == expect
    grp.close()
os.environ["AFX_SYSFS_ROOT"] = tree
run([mask(node_cpus[0])] * 2 + [mask(node_cpus[1])] * 2)
# a topology that is not exposed (containers: no numa_node files): nobody is pinned
os.environ["AFX_SYSFS_ROOT"] = os.path.join(tree, "nothing-here")
run([mask(allowed)] * 4)
# a device on a node that lists no usable CPU: that member is left alone, the others are pinned
os.environ["AFX_SYSFS_ROOT"] = tree
open(os.path.join(tree, "sys/devices/system/node/node1/cpulist"), "w").write("4000-4003\n")
run([mask(node_cpus[0])] * 2 + [mask(allowed)] * 2)
print("numa drive ok")
"""


def test_group_threads_follow_their_devices_numa_nodes(hostsim_lib, tmp_path):
    if len(os.sched_getaffinity(0)) < 4:
        pytest.skip("needs 4 usable CPUs")
    script = tmp_path / "drive.py"
    tree = tmp_path / "fake_root"
    tree.mkdir()
    script.write_text(NUMA_DRIVER % {"root": ROOT, "lib": hostsim_lib, "tree": str(tree)})
    asan = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    env = dict(os.environ, LD_PRELOAD=asan, ASAN_OPTIONS="detect_leaks=0")
    r = subprocess.run([sys.executable, str(script)], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "numa drive ok" in r.stdout, (r.stdout[-2000:], r.stderr[-3000:])
