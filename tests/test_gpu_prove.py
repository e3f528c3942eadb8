"""GPU parity for Issuer::issue, CredentialIssuance::verify and AnonymousCredential::show (SURVEY.md §8a rows
I0-I3, S0-S2).  Every random draw is an explicit input, so the HIP path must reproduce the ORACLE's outputs
byte for byte: (t, U, V), both proofs' (challenge, responses), every commitment and ciphertext."""
import ctypes as C
import hashlib

import numpy as np
import pytest

from tests.helpers import make_credentials

pytestmark = pytest.mark.gpu


def rows(list_of_lists):
    """[k][count] 32-byte values -> contiguous uint8 array [k][count][32]"""
    return np.frombuffer(b"".join(b"".join(r) for r in list_of_lists), dtype=np.uint8).copy()


def gpu_issue(afx, ctx, kinds, values_rows, t_wide, U_wide, seeds, n_out=None):
    cnt = len(t_wide)
    n = ctx.n if n_out is None else n_out
    req = afx.AttributesSoA()
    req.n_attributes = len(kinds)
    for i, k in enumerate(kinds):
        req.kinds[i] = k
    vals = rows(values_rows) if values_rows else np.zeros(1, np.uint8)
    req.values = vals.ctypes.data
    tw, uw, sd = (np.frombuffer(b"".join(x), dtype=np.uint8).copy() for x in (t_wide, U_wide, seeds))
    rnd = afx.IssueRandomness(tw.ctypes.data, uw.ctypes.data, sd.ctypes.data)
    o = {k: np.zeros(32 * cnt, np.uint8) for k in ("t", "U", "V", "challenge")}
    o["responses"] = np.zeros(32 * cnt * (n + 5), np.uint8)
    out = afx.IssuanceSoA(*(o[k].ctypes.data for k in ("t", "U", "V", "challenge", "responses")))
    status = np.full(cnt, 9, np.uint8)
    afx.check(afx.lib().afx_issue(ctx.h, C.byref(req), C.byref(rnd), cnt, C.byref(out), status.ctypes.data))
    return o, status


# secret: afx_ctx_set_secret_independent_addressing - every table entry read, 4-bit positional tables for the generators; same bytes
@pytest.mark.parametrize("secret", [False, True])
@pytest.mark.parametrize("n,layout,count", [(4, "SSPE", 70), (16, "SSSSSSSSPPPPEEEE", 20), (1, "P", 3), (3, "SSP", 65)])
def test_issue_matches_oracle_bytes_and_verifies(n, layout, count, secret):
    import aeonflux_amd as afx
    d = make_credentials(n, layout, count, b"gpu-issue-%d" % n)
    creds = d["creds"]
    ctx = afx.Context(d["params"], d["key"], d["ip"])
    ctx.set_secret_independent_addressing(secret)
    kinds = creds[0]["kinds"]
    values_rows = [[c["values"][i][:32] for c in creds] for i in range(n)]
    o, status = gpu_issue(afx, ctx, kinds, values_rows, [c["rnd"][0] for c in creds], [c["rnd"][1] for c in creds], [c["rnd"][2] for c in creds])
    assert status.tolist() == [0] * count
    for i, c in enumerate(creds):
        assert bytes(o["t"][32 * i:32 * i + 32]) == c["t"]
        assert bytes(o["U"][32 * i:32 * i + 32]) == c["U"]
        assert bytes(o["V"][32 * i:32 * i + 32]) == c["V"]
        assert bytes(o["challenge"][32 * i:32 * i + 32]) == c["challenge"]
        for k in range(n + 5):
            off = 32 * (k * count + i)
            assert bytes(o["responses"][off:off + 32]) == c["responses"][k], (i, k)
    # CredentialIssuance::verify on the GPU: honest ones pass, tampered ones fail, same as the oracle
    req = afx.AttributesSoA()
    req.n_attributes = n
    for i, k in enumerate(kinds):
        req.kinds[i] = k
    vals = rows(values_rows)
    if count > 4:
        o["V"][32 * 1] ^= 1
        o["responses"][32 * (2 * count + 3) + 5] ^= 8
        vals[32 * (0 * count + 4) + 1] ^= 2
    req.values = vals.ctypes.data
    iss = afx.IssuanceSoA(*(o[k].ctypes.data for k in ("t", "U", "V", "challenge", "responses")))
    status = np.full(count, 9, np.uint8)
    user = afx.Context(d["params"], None, d["ip"])
    afx.check(afx.lib().afx_verify_issuances(user.h, C.byref(req), C.byref(iss), n + 5, count, status.ctypes.data))
    want = []
    for i, c in enumerate(creds):
        v = [bytes(vals[32 * (k * count + i):32 * (k * count + i) + 32]) + c["values"][k][32:] for k in range(n)]
        resp = [bytes(o["responses"][32 * (k * count + i):32 * (k * count + i) + 32]) for k in range(n + 5)]
        want.append(d["user"].issuance_verify(kinds, v, bytes(o["t"][32 * i:32 * i + 32]), bytes(o["U"][32 * i:32 * i + 32]),
                                              bytes(o["V"][32 * i:32 * i + 32]), bytes(o["challenge"][32 * i:32 * i + 32]), resp))
    assert status.tolist() == want
    if count > 4:
        assert want[1] == 1 and want[3] == 1 and want[4] == 1 and sum(want) == 3
    afx.check(afx.lib().afx_verify_issuances(user.h, C.byref(req), C.byref(iss), n + 4, count, status.ctypes.data))
    assert status.tolist() == [1] * count   # wrong response count: zkp rejects every proof
    ctx.close()
    user.close()


def test_issue_wrong_attribute_count_and_identity_plaintext(flows):
    import aeonflux_amd as afx
    d = make_credentials(3, "SSP", 2, b"gpu-issue-len")
    ctx = afx.Context(d["params"], d["key"], d["ip"])
    creds = d["creds"]
    values_rows = [[c["values"][i][:32] for c in creds] for i in range(2)]
    o, status = gpu_issue(afx, ctx, creds[0]["kinds"][:2], values_rows, [c["rnd"][0] for c in creds], [c["rnd"][1] for c in creds],
                          [c["rnd"][2] for c in creds])
    assert status.tolist() == [afx.ST_MAC_CREATION] * 2   # amacs.rs:285-287
    ctx.close()
    # issuance.rs:272-295: an all-zero plaintext encodes to the identity; issuance succeeds, the user's verify fails
    r = next(f for f in flows if f["name"] == "issuance_proof_identity_plaintext")
    H = bytes.fromhex
    ctx = afx.Context(H(r["params"]), H(r["key"]), H(r["issuer_params"]))
    i = r["issue"]
    vals = [[H(v)[:32]] for v in i["values"]]
    o, status = gpu_issue(afx, ctx, i["kinds"], vals, [H(i["t_wide"])], [H(i["U_wide"])], [H(i["rng_seed"])])
    assert status.tolist() == [0]
    assert bytes(o["V"]).hex() == i["V"] and bytes(o["challenge"]).hex() == i["challenge"]
    req = afx.AttributesSoA()
    req.n_attributes = len(i["kinds"])
    for k, x in enumerate(i["kinds"]):
        req.kinds[k] = x
    va = rows(vals)
    req.values = va.ctypes.data
    iss = afx.IssuanceSoA(*(o[k].ctypes.data for k in ("t", "U", "V", "challenge", "responses")))
    st = np.full(1, 9, np.uint8)
    afx.check(afx.lib().afx_verify_issuances(ctx.h, C.byref(req), C.byref(iss), len(i["kinds"]) + 5, 1, st.ctypes.data))
    assert st.tolist() == [1] and r["issuance_verify"] == 1
    ctx.close()


def gpu_show(afx, ctx, kinds, creds, keypairs, z_wide, seeds, enc_seeds):
    """creds: list of dicts (values 96-byte records, t, U, V).  Returns (arrays dict, shape, status)"""
    n, cnt = len(kinds), len(creds)
    nsp = sum(1 for k in kinds if k == 4)
    hs = sum(1 for k in kinds if k == 1)
    cs = afx.CredentialsSoA()
    cs.n_attributes = n
    for i, k in enumerate(kinds):
        cs.kinds[i] = k
    a = {
        "values": rows([[c["values"][i][:32] for c in creds] for i in range(n)]),
        "M2": rows([[c["values"][i][32:64] for c in creds] for i in range(n)]),
        "m3": rows([[c["values"][i][64:96] for c in creds] for i in range(n)]),
        "t": rows([[c["t"] for c in creds]]), "U": rows([[c["U"] for c in creds]]), "V": rows([[c["V"] for c in creds]]),
    }
    for k, v in a.items():
        setattr(cs, k, v.ctypes.data)
    kp = None
    if keypairs is not None:
        ka = {f: rows([[k[32 * j:32 * j + 32] for k in keypairs]]) for j, f in enumerate(("a", "a0", "a1", "pk"))}
        kp = afx.KeypairsSoA(*(ka[f].ctypes.data for f in ("a", "a0", "a1", "pk")))
    zw = np.frombuffer(b"".join(z_wide), dtype=np.uint8).copy()
    sd = np.frombuffer(b"".join(seeds), dtype=np.uint8).copy()
    es = rows([[e[32 * j:32 * j + 32] for e in enc_seeds] for j in range(nsp)]) if nsp else np.zeros(1, np.uint8)
    rnd = afx.ShowRandomness(zw.ctypes.data, sd.ctypes.data, es.ctypes.data)
    o = {k: np.zeros(32 * cnt, np.uint8) for k in ("challenge", "C_x_0", "C_x_1", "C_V")}
    o["responses"] = np.zeros(32 * cnt * (3 + hs), np.uint8)
    o["C_y"] = np.zeros(32 * cnt * n, np.uint8)
    o["attr_values"] = np.zeros(32 * cnt * n, np.uint8)
    eouts = (afx.EncProofOut * max(1, nsp))()
    o["enc"] = []
    for e in range(nsp):
        d = {f: np.zeros(32 * cnt * (6 if f == "responses" else 1), np.uint8)
             for f in ("challenge", "responses", "pk", "E1", "E2", "C_y_1", "C_y_2", "C_y_3", "C_y_2p")}
        for f, v in d.items():
            setattr(eouts[e], f, v.ctypes.data)
        o["enc"].append(d)
    out = afx.PresentationOut()
    for f in ("challenge", "responses", "C_x_0", "C_x_1", "C_V", "C_y", "attr_values"):
        setattr(out, f, o[f].ctypes.data)
    out.enc = C.cast(eouts, C.POINTER(afx.EncProofOut))
    shape = afx.Shape()
    status = np.full(cnt, 9, np.uint8)
    afx.check(afx.lib().afx_show(ctx.h, C.byref(cs), C.byref(kp) if kp is not None else None, C.byref(rnd), cnt, C.byref(out),
                                 C.byref(shape), status.ctypes.data))
    return o, shape, status


@pytest.mark.parametrize("n,layout,hide,count", [
    (4, "SSPE", [0, 3], 70),
    (8, "SSPPEEEE", [4, 5, 6, 7], 20),
    (6, "SSSPSE", [0, 2, 4, 5], 33),
    (1, "S", [], 2),
    (3, "ESS", [0], 5),
    (32, "E" * 32, list(range(32)), 2),     # 32 proofs of encryption per presentation
    (32, "S" * 32, list(range(32)), 2),     # 32 hidden scalars
])
@pytest.mark.parametrize("secret", [False, True])   # afx_ctx_set_secret_independent_addressing: same bytes
def test_show_matches_oracle_bytes_and_gpu_verifies(n, layout, hide, count, secret):
    import aeonflux_amd as afx
    d = make_credentials(n, layout, count, b"gpu-show-%d" % n)
    take, user, issuer = d["take"], d["user"], d["issuer"]
    kinds = list(d["creds"][0]["kinds"])
    for i in hide:
        kinds[i] = 1 if kinds[i] == 0 else 4
    nsp = sum(1 for k in kinds if k == 4)
    kps = [user.keypair_derive(take(64)) for _ in range(count)]
    zw = [take(64) for _ in range(count)]
    sd = [take(32) for _ in range(count)]
    es = [take(32 * nsp) for _ in range(count)]
    want = []
    for c, kp, z, s, e in zip(d["creds"], kps, zw, sd, es):
        st, p = user.show(kinds, c["values"], c["t"], c["U"], c["V"], kp, z, s, e)
        assert st == 0
        want.append(p)
    uctx = afx.Context(d["params"], None, d["ip"])
    uctx.set_secret_independent_addressing(secret)
    o, shape, status = gpu_show(afx, uctx, kinds, d["creds"], kps, zw, sd, es)
    uctx.close()
    assert status.tolist() == [0] * count
    p0 = want[0]
    assert (shape.n_attributes, shape.n_responses, shape.n_hidden_scalars, shape.n_enc_proofs) == (n, p0.n_responses, p0.n_hidden_scalars, p0.n_enc_proofs)
    assert list(shape.kinds[:n]) == list(p0.kinds[:n])
    assert list(shape.hidden_scalar_indices[:p0.n_hidden_scalars]) == list(p0.hidden_scalar_indices[:p0.n_hidden_scalars])
    assert list(shape.enc_indices[:nsp]) == [p0.enc[e].index for e in range(nsp)]

    def cell(arr, k, i):
        return bytes(arr[32 * (k * count + i):32 * (k * count + i) + 32])
    for i, p in enumerate(want):
        assert cell(o["challenge"], 0, i) == bytes(p.challenge)
        for k in range(p.n_responses):
            assert cell(o["responses"], k, i) == bytes(p.responses[k])
        assert cell(o["C_x_0"], 0, i) == bytes(p.C_x_0) and cell(o["C_x_1"], 0, i) == bytes(p.C_x_1) and cell(o["C_V"], 0, i) == bytes(p.C_V)
        for k in range(n):
            assert cell(o["C_y"], k, i) == bytes(p.C_y[k])
            if p.kinds[k] in (0, 2):
                assert cell(o["attr_values"], k, i) == bytes(p.attr_values[k])
        for e in range(nsp):
            q, g = p.enc[e], o["enc"][e]
            assert cell(g["challenge"], 0, i) == bytes(q.challenge)
            for k in range(6):
                assert cell(g["responses"], k, i) == bytes(q.responses[k])
            for f in ("pk", "E1", "E2", "C_y_1", "C_y_2", "C_y_3", "C_y_2p"):
                assert cell(g[f], 0, i) == bytes(getattr(q, f)), f
    # and the issuer's GPU verify agrees with the oracle on the oracle-made twins
    from tests.helpers import gpu_verify
    ictx = afx.Context(d["params"], d["key"], d["ip"])
    ictx.set_secret_independent_addressing(secret)   # the verifier's side of the mode: the key's terms of Z
    assert gpu_verify(afx, ictx, want) == [issuer.verify_presentation(p) for p in want]
    ictx.close()


def test_show_without_key_is_no_symmetric_key():
    import aeonflux_amd as afx
    d = make_credentials(2, "ES", 3, b"gpu-show-nokey")
    kinds = [4, 0]
    uctx = afx.Context(d["params"], None, d["ip"])
    o, shape, status = gpu_show(afx, uctx, kinds, d["creds"], None, [bytes(64)] * 3, [bytes(32)] * 3, [bytes(32)] * 3)
    uctx.close()
    assert status.tolist() == [afx.ST_NO_SYMMETRIC_KEY] * 3   # presentation.rs:150-157


def test_issuer_keygen_matches_oracle():
    import oracle
    import aeonflux_amd as afx
    st = hashlib.shake_256(b"gpu-keygen").digest(1 << 15)
    for n in (1, 2, 4, 16):
        params, used = oracle.system_parameters_generate(n, st)
        key, ip = oracle.issuer_new(params, st[used:used + 64 * (4 + n)])
        W = C.create_string_buffer(32)
        out = C.create_string_buffer(64)
        afx.check(afx.lib().afx_issuer_keygen(0, params, len(params), key[:-32], len(key) - 32, W, out))
        assert W.raw == key[-32:] and out.raw == ip


def test_maximum_attribute_count_full_cycle():
    """AFX_MAX_ATTRIBUTES = 32 attributes (mixed kinds, hidden scalars and trailing hidden points): issue, the user's
    issuance check, show and the issuer's verify on the GPU, all byte-identical to the oracle."""
    import aeonflux_amd as afx
    from aeonflux_amd import batch
    n, layout, count = 32, "SSSSSSSSSSSSPPPPPPPPPPPPSSSSEEEE", 5
    hide = [0, 5, 11, 24, 28, 29, 30, 31]
    d = make_credentials(n, layout, count, b"gpu-max-attrs")
    creds, user, issuer = d["creds"], d["user"], d["issuer"]
    ctx = afx.Context(d["params"], d["key"], d["ip"])
    kinds = creds[0]["kinds"]
    col = lambda f: np.stack([np.frombuffer(f(c), np.uint8) for c in creds])
    values = np.stack([col(lambda c, i=i: c["values"][i][:32]) for i in range(n)])
    iss, st = batch.issue(ctx, kinds, values, col(lambda c: c["rnd"][0]), col(lambda c: c["rnd"][1]), col(lambda c: c["rnd"][2]))
    assert st.tolist() == [0] * count
    for i, c in enumerate(creds):
        assert (iss["t"][i].tobytes(), iss["U"][i].tobytes(), iss["V"][i].tobytes(), iss["challenge"][i].tobytes()) == (c["t"], c["U"], c["V"], c["challenge"])
        assert [iss["responses"][k, i].tobytes() for k in range(n + 5)] == c["responses"]
    assert batch.verify_issuances(ctx, kinds, values, iss).tolist() == [0] * count
    iss["responses"][n + 4, 2, 0] ^= 1
    assert batch.verify_issuances(ctx, kinds, values, iss).tolist() == [0, 0, 1, 0, 0]
    skinds = list(kinds)
    for i in hide:
        skinds[i] = 1 if skinds[i] == 0 else 4
    take = d["take"]
    kps = [user.keypair_derive(take(64)) for _ in range(count)]
    zw, sd, es = [take(64) for _ in range(count)], [take(32) for _ in range(count)], [take(32 * 4) for _ in range(count)]
    want = [user.show(skinds, c["values"], c["t"], c["U"], c["V"], kp, z, s, e)[1] for c, kp, z, s, e in zip(creds, kps, zw, sd, es)]
    M2 = np.stack([col(lambda c, i=i: c["values"][i][32:64]) for i in range(n)])
    m3 = np.stack([col(lambda c, i=i: c["values"][i][64:96]) for i in range(n)])
    kpd = {f: np.stack([np.frombuffer(k[32 * j:32 * j + 32], np.uint8) for k in kps]) for j, f in enumerate(("a", "a0", "a1", "pk"))}
    esr = np.stack([np.stack([np.frombuffer(e[32 * j:32 * j + 32], np.uint8) for e in es]) for j in range(4)])
    pres, shape, st = batch.show(ctx, skinds, values, col(lambda c: c["t"]), col(lambda c: c["U"]), col(lambda c: c["V"]), kpd,
                                 np.stack([np.frombuffer(z, np.uint8) for z in zw]), np.stack([np.frombuffer(s, np.uint8) for s in sd]), esr, M2, m3)
    assert st.tolist() == [0] * count
    for i, p in enumerate(want):
        assert pres["challenge"][i].tobytes() == bytes(p.challenge)
        assert all(pres["responses"][k, i].tobytes() == bytes(p.responses[k]) for k in range(p.n_responses))
        assert all(pres["C_y"][k, i].tobytes() == bytes(p.C_y[k]) for k in range(n))
        assert all(pres["enc"][e]["E2"][i].tobytes() == bytes(p.enc[e].E2) and pres["enc"][e]["challenge"][i].tobytes() == bytes(p.enc[e].challenge) for e in range(4))
    assert [issuer.verify_presentation(p) for p in want] == [0] * count
    assert batch.verify_presentations(ctx, shape, pres).tolist() == [0] * count
    pres["C_y"][17, 3, 4] ^= 2
    assert batch.verify_presentations(ctx, shape, pres).tolist() == [0, 0, 0, 1, 0]
    ctx.close()


def test_issuance_verification_recomputes_the_oracles_challenge():
    """CredentialIssuance::verify: recomputed challenge equal to the oracle's, also for tampered issuances"""
    import oracle
    import aeonflux_amd as afx
    from aeonflux_amd import batch
    n, cnt = 4, 24
    d = make_credentials(n, "SSPE", cnt, b"gpu-iss-trace")
    user = d["user"]
    kinds = list(d["creds"][0]["kinds"])
    vals = np.zeros((n, cnt, 32), np.uint8)
    iss = {k: np.zeros((cnt, 32), np.uint8) for k in ("t", "U", "V", "challenge")}
    iss["responses"] = np.zeros((n + 5, cnt, 32), np.uint8)
    want, want_status = [], []
    for i, cr in enumerate(d["creds"]):
        t, resp = bytearray(cr["t"]), [bytearray(r) for r in cr["responses"]]
        if i % 4 == 1:
            resp[i % (n + 5)][3] ^= 2
        if i % 4 == 2:
            t[5] ^= 1
        oracle.debug_reset()
        want_status.append(user.issuance_verify(kinds, cr["values"], bytes(t), cr["U"], cr["V"], cr["challenge"], [bytes(r) for r in resp]))
        commits, c2 = oracle.debug_last()
        want.append(c2 if commits else None)
        for k in range(n):
            vals[k, i] = np.frombuffer(cr["values"][k][:32], np.uint8)
        for name, v in (("t", bytes(t)), ("U", cr["U"]), ("V", cr["V"]), ("challenge", cr["challenge"])):
            iss[name][i] = np.frombuffer(v, np.uint8)
        for k in range(n + 5):
            iss["responses"][k, i] = np.frombuffer(bytes(resp[k]), np.uint8)
    ctx = afx.Context(d["params"], None, d["ip"])
    ctx.set_challenge_trace(1, cnt)
    st = batch.verify_issuances(ctx, kinds, vals, iss)
    got = ctx.get_challenge_trace()
    ctx.close()
    assert st.tolist() == want_status and 0 in want_status and 1 in want_status
    reached = 0
    for i in range(cnt):
        if want[i] is not None:
            assert bytes(got[0, i]) == want[i], i
            reached += 1
    assert reached >= cnt - 6


@pytest.mark.parametrize("secret", ["prover", False])
def test_a_wide_issue_pass_takes_several_terms_per_chain_and_issues_the_same_bytes(secret):
    """20 000 requests of 16 attributes through the latency plan (afx_ctx_set_small_batch_items raised above them): its stages would
    put more than four waves of one-term chains on every SIMD, so the jobs take two or more terms per chain (engine.cpp
    Assembler::msm; with secrets: narrow chains of several terms, and the lane exchange beside them).  Every output byte is the
    oracle's (20 distinct requests tiled: equal inputs, equal outputs), and the plan has fewer chains than a small call's."""
    import aeonflux_amd as afx
    n, layout, distinct, reps = 16, "SSSSSSSSPPPPEEEE", 20, 1000
    d = make_credentials(n, layout, distinct, b"gpu-issue-%d" % n)
    creds = d["creds"] * reps
    count = len(creds)
    ctx = afx.Context(d["params"], d["key"], d["ip"])
    ctx.set_secret_independent_addressing(secret)
    kinds = creds[0]["kinds"]
    values_rows = [[c["values"][i][:32] for c in creds] for i in range(n)]
    args = (kinds, values_rows, [c["rnd"][0] for c in creds], [c["rnd"][1] for c in creds], [c["rnd"][2] for c in creds])
    gpu_issue(afx, ctx, kinds, [r[:distinct] for r in values_rows], *[a[:distinct] for a in args[2:]])
    small_jobs = ctx.plan_stats()["msm_jobs"]
    ctx.set_small_batch_items(32768)
    o, status = gpu_issue(afx, ctx, *args)
    assert not status.any()
    assert ctx.plan_stats()["msm_jobs"] < small_jobs, (ctx.plan_stats()["msm_jobs"], small_jobs)
    for f in ("t", "U", "V", "challenge"):
        got = np.frombuffer(bytes(o[f]), np.uint8).reshape(count, 32)
        want = np.stack([np.frombuffer(c[f], np.uint8) for c in creds])
        assert np.array_equal(got, want), f
    got = np.frombuffer(bytes(o["responses"]), np.uint8).reshape(n + 5, count, 32)
    want = np.stack([np.stack([np.frombuffer(c["responses"][k], np.uint8) for c in creds]) for k in range(n + 5)])
    assert np.array_equal(got, want)
    ctx.close()
