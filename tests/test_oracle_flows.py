"""The reference's own tests (all randomised round trips) restated against the ORACLE with
seeded randomness, plus the committed flows.json regression fixtures.  CPU only."""
import hashlib

import pytest

import oracle
from oracle import (ATTR_EITHER_POINT, ATTR_PUBLIC_POINT, ATTR_PUBLIC_SCALAR, ATTR_SECRET_POINT,
                    ATTR_SECRET_SCALAR, Ctx)

H = bytes.fromhex


def pad96(b):
    return b + bytes(96 - len(b))


class Stream:
    def __init__(self, seed):
        self.b = hashlib.shake_256(seed).digest(1 << 16)
        self.pos = 0

    def take(self, n):
        out = self.b[self.pos:self.pos + n]
        self.pos += n
        return out


def setup(n, seed):
    s = Stream(seed)
    params, used = oracle.system_parameters_generate(n, s.b)
    s.pos = used
    key, ip = oracle.issuer_new(params, s.take(64 * (4 + n)))
    return s, params, key, ip


def test_flows_fixture_reproduces(flows):
    """every committed flow is reproduced bit for bit from its recorded inputs"""
    for r in flows:
        issuer = Ctx(H(r["params"]), H(r["key"]), H(r["issuer_params"]))
        user = Ctx(H(r["params"]), None, H(r["issuer_params"]))
        i = r["issue"]
        st, t, U, V, ch, resp = issuer.issue(i["kinds"], [H(v) for v in i["values"]], H(i["t_wide"]), H(i["U_wide"]), H(i["rng_seed"]))
        assert st == i["status"]
        assert (t.hex(), U.hex(), V.hex(), ch.hex()) == (i["t"], i["U"], i["V"], i["challenge"])
        assert [x.hex() for x in resp] == i["responses"]
        assert user.issuance_verify(i["kinds"], [H(v) for v in i["values"]], t, U, V, ch, resp) == r["issuance_verify"]
        s = r["show"]
        st, p = user.show(s["kinds"], [H(v) for v in s["values"]], t, U, V, H(s["keypair"]) if s["keypair"] else None,
                          H(s["z_wide"]), H(s["rng_seed"]), H(s["enc_seeds"]))
        assert st == s["status"]
        if st != 0:
            continue
        assert bytes(p.challenge).hex() == r["presentation"]["challenge"]
        assert [bytes(p.C_y[k]).hex() for k in range(p.n_attributes)] == r["presentation"]["C_y"]
        assert issuer.verify_presentation(p) == r["verify"]


def test_reference_test_outcomes(flows):
    """accept/reject outcomes the reference's tests assert (file:line in tests/gen_golden.py)"""
    by = {r["name"]: r for r in flows}
    for name in ("readme_4attrs_sSPe", "credential_proof_10_attributes", "credential_proof_10_attributes_with_plaintext",
                 "credential_proof_1_plaintext_hidden", "credential_proof_1_scalar_revealed", "switch_scalar_point",
                 "switch_point_scalar", "c3_8attrs_SSPPeeee", "hidden_scalars_mixed"):
        assert by[name]["issuance_verify"] == 0 and by[name]["verify"] == 0, name
    assert by["bad_credential_proof_1_scalar_revealed"]["verify"] == 1          # presentation.rs:618-638
    assert by["issuance_proof_identity_plaintext"]["issuance_verify"] == 1      # issuance.rs:272-295
    assert by["leading_hidden_point_fails"]["verify"] == 1
    assert by["no_symmetric_key"]["show"]["status"] == oracle.ST_NO_SYMMETRIC_KEY  # presentation.rs:150-157
    assert by["credential_proof_1_plaintext"]["show"]["status"] == 0            # presentation.rs:528-542


def test_sizes_and_serialisation():
    # amacs.rs:344-353, parameters.rs:34-40,384-392
    lib = oracle.lib()
    assert lib.afxo_sizeof_secret_key(2) == 32 * 7 + 4
    assert lib.afxo_sizeof_system_parameters(2) == 32 * (5 + 3 + 2 + 4) + 4
    assert lib.afxo_sizeof_system_parameters(8) == 32 * (5 + 16 + 4) + 4
    s, params, key, ip = setup(2, b"sizes")
    assert len(params) == lib.afxo_sizeof_system_parameters(2) and len(key) == lib.afxo_sizeof_secret_key(2)
    Ctx(params, key, ip)
    with pytest.raises(ValueError):
        Ctx(params[:-1], key, ip)
    with pytest.raises(ValueError):
        Ctx(params, key[:-32] + b"\xff" * 32, ip)


def test_wrong_attribute_count_is_mac_creation():
    # amacs.rs:285-287 -> errors.rs:141-142
    s, params, key, ip = setup(3, b"maclen")
    issuer = Ctx(params, key, ip)
    kinds = [ATTR_PUBLIC_SCALAR, ATTR_PUBLIC_SCALAR]
    vals = [pad96(oracle.scalar_reduce_wide(s.take(64))) for _ in range(2)]
    st = issuer.issue(kinds, vals, s.take(64), s.take(64), s.take(32))[0]
    assert st == oracle.ST_MAC_CREATION


def test_encrypt_decrypt_roundtrip_and_encryption_proof():
    # symmetric.rs:299-310, encryption.rs:222-244, encoding.rs:92-103
    s, params, key, ip = setup(5, b"encproof")
    user = Ctx(params, None, ip)
    issuer = Ctx(params, key, ip)
    kp = user.keypair_derive(s.take(64))
    for msg in (bytes(30), b"This is a tsunami alert test..", s.take(30)):
        pl, ctr = oracle.plaintext_from_bytes(msg)
        ct = oracle.encrypt(kp, pl)
        rc, dec = oracle.decrypt(kp, ct)
        assert rc == 0 and dec == pl
        data, ctr2 = oracle.decode_from_group(dec[:32])
        assert (data, ctr2) == (msg, ctr)
    # a presentation with one hidden point carries one encryption proof that verifies alone
    kinds = [ATTR_PUBLIC_SCALAR, ATTR_EITHER_POINT, ATTR_PUBLIC_SCALAR, ATTR_PUBLIC_SCALAR, ATTR_PUBLIC_SCALAR]
    pl, _ = oracle.plaintext_from_bytes(b"This is a tsunami alert test..")
    vals = [pad96(oracle.scalar_reduce_wide(s.take(64))) if k == ATTR_PUBLIC_SCALAR else pl for k in kinds]
    st, t, U, V, ch, resp = issuer.issue(kinds, vals, s.take(64), s.take(64), s.take(32))
    kinds[1] = ATTR_SECRET_POINT
    st, p = user.show(kinds, vals, t, U, V, kp, s.take(64), s.take(32), s.take(32))
    assert st == 0 and p.n_enc_proofs == 1 and p.enc[0].index == 1
    assert issuer.verify_encryption_proof(p.enc[0]) == 0
    rc, dec = oracle.decrypt(kp, bytes(p.enc[0].E1) + bytes(p.enc[0].E2))
    assert rc == 0 and dec == pl
    p.enc[0].C_y_3[0] ^= 2
    assert issuer.verify_encryption_proof(p.enc[0]) == 1


def test_tampering_rejects_every_field():
    s, params, key, ip = setup(4, b"tamper")
    issuer, user = Ctx(params, key, ip), Ctx(params, None, ip)
    kinds = [ATTR_PUBLIC_SCALAR, ATTR_PUBLIC_SCALAR, ATTR_PUBLIC_POINT, ATTR_EITHER_POINT]
    vals = [pad96(oracle.scalar_reduce_wide(s.take(64))), pad96(oracle.scalar_reduce_wide(s.take(64))),
            pad96(oracle.point_from_uniform(s.take(64))), oracle.plaintext_from_bytes(s.take(30))[0]]
    st, t, U, V, ch, resp = issuer.issue(kinds, vals, s.take(64), s.take(64), s.take(32))
    kp = user.keypair_derive(s.take(64))
    skinds = [ATTR_SECRET_SCALAR, ATTR_PUBLIC_SCALAR, ATTR_PUBLIC_POINT, ATTR_SECRET_POINT]
    st, p = user.show(skinds, vals, t, U, V, kp, s.take(64), s.take(32), s.take(32))
    assert issuer.verify_presentation(p) == 0
    other = oracle.point_from_uniform(s.take(64))
    osc = oracle.scalar_reduce_wide(s.take(64))

    def with_field(get, new):
        arr = get(p)
        old = bytes(arr)
        for i in range(32):
            arr[i] = new[i]
        rc = issuer.verify_presentation(p)
        for i in range(32):
            arr[i] = old[i]
        return rc
    fields = [lambda q: q.C_x_0, lambda q: q.C_x_1, lambda q: q.C_V, lambda q: q.C_y[0], lambda q: q.C_y[1], lambda q: q.C_y[2],
              lambda q: q.C_y[3], lambda q: q.attr_values[2], lambda q: q.enc[0].pk, lambda q: q.enc[0].E1, lambda q: q.enc[0].E2,
              lambda q: q.enc[0].C_y_1, lambda q: q.enc[0].C_y_2, lambda q: q.enc[0].C_y_3, lambda q: q.enc[0].C_y_2p]
    for f in fields:
        assert with_field(f, other) == 1          # a different valid point
        assert with_field(f, bytes(32)) == 1      # identity
        assert with_field(f, b"\xff" * 32) == 1   # undecodable
    for f in [lambda q: q.challenge, lambda q: q.responses[0], lambda q: q.responses[3], lambda q: q.attr_values[1],
              lambda q: q.enc[0].challenge, lambda q: q.enc[0].responses[5]]:
        assert with_field(f, osc) == 1
        assert with_field(f, b"\xff" * 32) == 1   # non-canonical scalar
    assert issuer.verify_presentation(p) == 0
    # structural mismatches the reference would panic on -> failure, never a fault
    p.n_responses = 3
    assert issuer.verify_presentation(p) == 1
    p.n_responses = 4
    p.hidden_scalar_indices[0] = 1
    assert issuer.verify_presentation(p) == 1
    p.hidden_scalar_indices[0] = 9
    assert issuer.verify_presentation(p) == 1
    p.hidden_scalar_indices[0] = 0
    p.enc[0].index = 7
    assert issuer.verify_presentation(p) == 1


def test_batch_soa_matches_per_item():
    import ctypes as C
    import numpy as np
    s, params, key, ip = setup(4, b"batch")
    issuer, user = Ctx(params, key, ip), Ctx(params, None, ip)
    kinds = [ATTR_PUBLIC_SCALAR, ATTR_PUBLIC_SCALAR, ATTR_PUBLIC_POINT, ATTR_EITHER_POINT]
    skinds = [ATTR_SECRET_SCALAR, ATTR_PUBLIC_SCALAR, ATTR_PUBLIC_POINT, ATTR_SECRET_POINT]
    pres = []
    for i in range(6):
        vals = [pad96(oracle.scalar_reduce_wide(s.take(64))), pad96(oracle.scalar_reduce_wide(s.take(64))),
                pad96(oracle.point_from_uniform(s.take(64))), oracle.plaintext_from_bytes(s.take(30))[0]]
        st, t, U, V, ch, resp = issuer.issue(kinds, vals, s.take(64), s.take(64), s.take(32))
        kp = user.keypair_derive(s.take(64))
        st, p = user.show(skinds, vals, t, U, V, kp, s.take(64), s.take(32), s.take(32))
        pres.append(p)
    pres[2].C_V[5] ^= 1
    pres[4].enc[0].responses[2][0] ^= 1
    from tests.soa import pack_presentations
    shape, soa, keep = pack_presentations(pres)
    status = np.full(len(pres), 9, dtype=np.uint8)
    for threads in (1, 3):
        oracle.lib().afxo_verify_presentations_soa(issuer.h, C.byref(shape), C.byref(soa), len(pres), status.ctypes.data, threads)
        assert status.tolist() == [issuer.verify_presentation(p) for p in pres] == [0, 0, 1, 0, 1, 0]


def test_batch_forms_equal_the_per_item_calls():
    """The oracle's threaded struct-of-arrays forms (oracle/batch.c: statuses + recomputed challenges for presentations and
    issuances, batch issue), which the full-size GPU tests use on 2^16-item samples, against the per-item entry points the
    golden flows pin."""
    import numpy as np
    import oracle
    from tests.helpers import corrupt, make_batch, make_credentials
    from tests.soa import pack_presentations
    params, key, ip, issuer, pres = make_batch(4, "SSPE", [0, 3], 14, b"oracle-batch-forms")
    corrupt(pres, b"oracle-batch-corrupt")
    shape, soa, keep = pack_presentations(pres)
    st, trace, reached = oracle.verify_presentations_traced(issuer, shape, soa, len(pres), threads=3)
    assert st.tolist() == [issuer.verify_presentation(p) for p in pres] and 0 < st.sum() < len(pres)
    for i, p in enumerate(pres):
        q = oracle.Presentation.from_buffer_copy(bytes(p))
        q.n_enc_proofs = 0
        oracle.debug_reset()
        issuer.verify_presentation(q)
        commits, ch = oracle.debug_last()
        assert bool(reached[0, i]) == bool(commits) and (not commits or bytes(trace[0, i]) == ch)
        oracle.debug_reset()
        issuer.verify_encryption_proof(p.enc[0])
        commits, ch = oracle.debug_last()
        assert bool(reached[1, i]) == bool(commits) and (not commits or bytes(trace[1, i]) == ch)
    assert reached.sum() >= 2 * len(pres) - 6
    d = make_credentials(5, "SSPES", 9, b"oracle-batch-issue")
    creds, iss_ctx, user = d["creds"], d["issuer"], d["user"]
    kinds = creds[0]["kinds"]
    values = np.stack([np.stack([np.frombuffer(c["values"][k][:32], np.uint8) for c in creds]) for k in range(5)])
    rnd = [np.stack([np.frombuffer(c["rnd"][j], np.uint8) for c in creds]) for j in range(3)]
    o, st = oracle.issue_soa(iss_ctx, kinds, values, *rnd, threads=2)
    assert not st.any()
    for i, c in enumerate(creds):
        assert (bytes(o["t"][i]), bytes(o["U"][i]), bytes(o["V"][i]), bytes(o["challenge"][i])) == (c["t"], c["U"], c["V"], c["challenge"])
        assert [bytes(o["responses"][k, i]) for k in range(10)] == c["responses"]
    o["V"][2, 5] ^= 1
    o["responses"][3, 6, 0] ^= 2
    st, trace, reached = oracle.verify_issuances_traced(user, kinds, values, o, threads=4)
    assert st.tolist() == [0, 0, 1, 0, 0, 0, 1, 0, 0] and reached.sum() >= 8 and reached[6]   # item 2's V may no longer decode
    for i in (0, 2, 6):
        oracle.debug_reset()
        vals = [bytes(values[k, i]) + bytes(64) for k in range(5)]
        assert user.issuance_verify(kinds, vals, bytes(o["t"][i]), bytes(o["U"][i]), bytes(o["V"][i]), bytes(o["challenge"][i]),
                                    [bytes(o["responses"][k, i]) for k in range(10)]) == int(st[i])
        assert not reached[i] or oracle.debug_last()[1] == bytes(trace[i])
    assert (bytes(trace[0]) == bytes(o["challenge"][0])) and bytes(trace[6]) != bytes(o["challenge"][6])
