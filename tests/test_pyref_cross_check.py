"""CPU-only: tests/golden/flows.json (computed by the C oracle) replayed through the second, independently written
restatement tests/pyref (pure Python).  Every output byte of issue and show - (t, U, V), challenges, responses, the
prover's commitments - and every accept / reject decision with its recomputed commitments must be identical.
The reference pins none of this (no vectors, not buildable here), so parity stays "unpinned by the
reference"; what this test adds is that two restatements written apart from the Rust sources agree on all of it."""
import pytest

from tests.pyref import keccak, ristretto, statements as S

H = bytes.fromhex


def test_pyref_primitives_against_third_party_vectors(kat, primitives):
    base = ristretto.decode(H("e2f2ae0a6abc4e71a884a961c500515f58e30b6aa582dd8db6a65945e08d2d76"))
    assert ristretto.encode(ristretto.add(base, base)).hex() == "6a493210f7499cd17fecb510ae0cea23a110e8d5b901f8acadd3095c73a3b919"
    t = keccak.Transcript(b"test protocol")
    t.append_message(b"some label", b"some data")
    assert t.challenge_bytes(b"challenge", 32).hex() == "d5a21972d0d5fe320c0d263fac7fffb8145aa640af6e9bca177c03c7efcf0615"
    # libsodium-computed vectors (tests/gen_golden.py): hash-to-group, scalar multiplication, validity
    n = 0
    for v in primitives["from_uniform"]:
        assert ristretto.encode(ristretto.from_uniform_bytes(H(v["in"]))).hex() == v["out"]
        n += 1
    for v in primitives["scalarmult"]:
        p = ristretto.decode(H(v["p"]))
        k = int.from_bytes(H(v["s"]), "little")
        if p is not None and v.get("out") and k < 2**255:   # libsodium clears bit 255 of the scalar
            assert ristretto.encode(ristretto.mul(k, p)).hex() == v["out"]
            n += 1
    for v in primitives["add"]:
        assert ristretto.encode(ristretto.add(ristretto.decode(H(v["p"])), ristretto.decode(H(v["q"])))).hex() == v["out"]
    for v in primitives["sub"]:
        assert ristretto.encode(ristretto.sub(ristretto.decode(H(v["p"])), ristretto.decode(H(v["q"])))).hex() == v["out"]
    for v in primitives["validity"]:
        assert (ristretto.decode(H(v["in"])) is not None) == bool(v["valid"]), v["in"]
        n += 1
    for v in primitives["scalar_reduce_wide"]:
        assert ristretto.sc_bytes(ristretto.sc_from_wide(H(v["in"]))).hex() == v["out"]
    assert n > 600


def test_pyref_rfc9496_appendix_a_and_merlins_multi_block_vector(kat):
    base = ristretto.decode(H(kat["rfc9496_B"]))
    acc = None
    for k in range(1, 16):
        acc = base if acc is None else ristretto.add(acc, base)
        assert ristretto.encode(acc).hex() == kat["rfc9496_generator_multiples"][k]
        assert ristretto.encode(ristretto.mul(k, base)).hex() == kat["rfc9496_generator_multiples"][k]
    assert ristretto.encode(ristretto.sub(base, base)).hex() == kat["rfc9496_generator_multiples"][0]
    for reason, encs in kat["rfc9496_bad_encodings"].items():
        for e in encs:
            assert ristretto.decode(H(e)) is None, (reason, e)
    for v in kat["rfc9496_from_uniform_bytes"]:
        assert ristretto.encode(ristretto.from_uniform_bytes(H(v["in"]))).hex() == v["out"]
    v = kat["merlin_equivalence_complex"]
    t = keccak.Transcript(v["label"].encode())
    t.append_message(v["first_label"].encode(), v["first_data"].encode())
    for _ in range(v["rounds"]):
        chl = t.challenge_bytes(v["challenge_label"].encode(), 32)
        t.append_message(v["big_label"].encode(), bytes([v["big_byte"]]) * v["big_len"])
        t.append_message(v["feedback_label"].encode(), chl)
    assert chl.hex() == v["last_challenge32"]


def _values(rec):
    return [H(v) for v in rec["values"]]


def test_every_flow_replays_byte_for_byte(flows):
    checked = {"issue": 0, "issuance_verify": 0, "show": 0, "verify": 0}
    for f in flows:
        params, key, ip = H(f["params"]), H(f["key"]), H(f["issuer_params"])
        iss = f["issue"]
        st, o = S.issue(params, key, ip, iss["kinds"], _values(iss), H(iss["t_wide"]), H(iss["U_wide"]), H(iss["rng_seed"]))
        assert st == iss["status"], f["name"]
        if st == 0:
            for k in ("t", "U", "V", "challenge"):
                assert o[k].hex() == iss[k], (f["name"], k)
            assert [r.hex() for r in o["responses"]] == iss["responses"], f["name"]
            assert [c.hex() for c in o["commitments"]] == f["issuance_commitments"], f["name"]
            checked["issue"] += 1
            vst, coms = S.issuance_verify(params, ip, iss["kinds"], _values(iss), H(iss["t"]), H(iss["U"]), H(iss["V"]), H(iss["challenge"]),
                                          [H(r) for r in iss["responses"]])
            assert vst == f["issuance_verify"], f["name"]
            if vst == 0:
                assert [c.hex() for c in coms] == f["issuance_commitments"], f["name"]   # the verifier recomputes the prover's commitments
            checked["issuance_verify"] += 1
        sh = f.get("show")
        if not sh:
            continue
        kp = H(sh["keypair"]) if sh.get("keypair") else None
        st, p = S.show(params, ip, sh["kinds"], _values(sh), H(iss["t"]), H(iss["U"]), H(iss["V"]), kp, H(sh["z_wide"]), H(sh["rng_seed"]),
                       H(sh["enc_seeds"]))
        assert st == sh["status"], f["name"]
        checked["show"] += 1
        if st != 0:
            continue
        want = f["presentation"]
        for k in ("challenge", "C_x_0", "C_x_1", "C_V"):
            assert p[k].hex() == want[k], (f["name"], k)
        assert [r.hex() for r in p["responses"]] == want["responses"] and [c.hex() for c in p["C_y"]] == want["C_y"], f["name"]
        assert p["kinds"] == want["kinds"] and p["hidden_scalar_indices"] == want["hidden_scalar_indices"], f["name"]
        assert [v.hex() for v in p["attr_values"]] == want["attr_values"], f["name"]
        assert len(p["enc"]) == len(want["enc"])
        for e, we in zip(p["enc"], want["enc"]):
            for k in ("challenge", "pk", "E1", "E2", "C_y_1", "C_y_2", "C_y_3", "C_y_2p"):
                assert e[k].hex() == we[k], (f["name"], k)
            assert [r.hex() for r in e["responses"]] == we["responses"] and e["index"] == we["index"], f["name"]
        # Issuer::verify on the fixture's presentation (some fixtures are tampered with after show)
        pres = dict(kinds=want["kinds"], attr_values=[H(v) for v in want["attr_values"]], hidden_scalar_indices=want["hidden_scalar_indices"],
                    challenge=H(want["challenge"]), responses=[H(r) for r in want["responses"]], C_x_0=H(want["C_x_0"]), C_x_1=H(want["C_x_1"]),
                    C_V=H(want["C_V"]), C_y=[H(c) for c in want["C_y"]],
                    enc=[dict(index=we["index"], challenge=H(we["challenge"]), responses=[H(r) for r in we["responses"]],
                              **{k: H(we[k]) for k in ("pk", "E1", "E2", "C_y_1", "C_y_2", "C_y_3", "C_y_2p")}) for we in want["enc"]])
        vst, coms = S.verify_presentation(params, key, ip, pres)
        assert vst == f["verify"], f["name"]
        if vst == 0:
            assert [c.hex() for c in coms] == f["verify_last_commitments"], f["name"]
        checked["verify"] += 1
    assert checked["issue"] >= 16 and checked["show"] >= 16 and checked["verify"] >= 15, checked


def test_libsodium_recomputes_every_group_operation_of_every_flow(flows):
    """A third implementation under the statement layer: every scalar multiplication, multiscalar sum, addition, subtraction and
    negation that tests/pyref performs while replaying the golden flows - tags, messages, both provers' commitments, ciphertexts,
    Z, every commitment a verifier recomputes - is recomputed with libsodium's ristretto255 from the recorded scalars and point
    encodings.  With test_every_flow_replays_byte_for_byte (oracle == pyref on every byte) this pins the GROUP VALUES of the
    flows to libsodium; what stays pinned only by the two restatements agreeing is the framing: transcript labels, the order of
    allocations and constraints, which terms a constraint has.  libsodium exists in the build container only."""
    from tests import sodium_replay
    sod = sodium_replay.load()
    if sod is None:
        pytest.skip("no libsodium on this machine (it lives in the build container)")
    n = sodium_replay.replay_flows(sod, flows)
    assert n["msm"] >= 150 and n["mul"] >= 200 and n["sub"] >= 40 and n["neg"] >= 40, n


def test_double_and_compress_equals_encode_of_the_double():
    """the formula behind the engine's k_compress2x (kernels.hip): the encoding of 2P without a square root, batched over one
    inversion, against encode(P + P) - random points, random projective scalings, every representative of a coset (P + E[4]),
    identities in between, and the halved-scalar identity the engine relies on: 2 * ((s/2 mod l) P) encodes like s P."""
    import hashlib
    R = ristretto
    tors = [R.IDENTITY, (0, R.P - 1, 1, 0), (R.SQRT_M1, 0, 1, 0), ((-R.SQRT_M1) % R.P, 0, 1, 0)]
    pts = []
    for i in range(40):
        p = R.from_uniform_bytes(hashlib.sha512(b"c2x-%d" % i).digest())
        lam = int.from_bytes(hashlib.sha256(b"lam-%d" % i).digest(), "little") % R.P or 1
        p = R.add(p, tors[i % 4])
        pts.append(tuple(c * lam % R.P for c in p))
        if i % 7 == 3:
            pts.append(tors[(i // 7) % 4])          # a representative of the identity in the middle of the batch
    got = R.double_and_compress(pts)
    assert got == [R.encode(R.add(p, p)) for p in pts]
    assert bytes(32) in got
    inv2 = (R.L + 1) // 2
    for i in range(8):
        p = R.from_uniform_bytes(hashlib.sha512(b"half-%d" % i).digest())
        s = int.from_bytes(hashlib.sha512(b"s-%d" % i).digest(), "little") % R.L
        half = R.mul(s * inv2 % R.L, R.add(p, tors[i % 4]))
        assert R.double_and_compress([half])[0] == R.encode(R.mul(s, p))


def test_strict_mode_statement_agrees_with_the_oracle():
    """the engine's opt-in strict mode (own-position constraint #3, one proof of encryption per hidden group element, the DLEQ with
    its C_y_1) is not the reference's statement, so no reference pins it either: the two restatements must agree on it as well"""
    from tests.helpers import make_credentials
    for n, layout, hide in ((3, "ESS", [0]), (6, "SESPSE", [0, 1, 4, 5]), (4, "SSPE", [0, 3])):
        d = make_credentials(n, layout, 3, b"pyref-strict-%d" % n)
        user, issuer, take = d["user"], d["issuer"], d["take"]
        user.set_strict(True)
        issuer.set_strict(True)
        kinds = list(d["creds"][0]["kinds"])
        for i in hide:
            kinds[i] = 1 if kinds[i] == 0 else 4
        nsp = sum(1 for k in kinds if k == 4)
        made = []
        for c in d["creds"]:
            kp, z, sd, es = user.keypair_derive(take(64)), take(64), take(32), take(32 * nsp)
            st, p = user.show(kinds, c["values"], c["t"], c["U"], c["V"], kp, z, sd, es)
            st2, q = S.show(d["params"], d["ip"], kinds, c["values"], c["t"], c["U"], c["V"], kp, z, sd, es, strict=True)
            assert st == st2 == 0
            assert q["challenge"] == bytes(p.challenge) and q["responses"] == [bytes(p.responses[k]) for k in range(p.n_responses)], layout
            assert [e["challenge"] for e in q["enc"]] == [bytes(p.enc[e].challenge) for e in range(nsp)]
            made.append((p, {k: q[k] for k in ("kinds", "attr_values", "hidden_scalar_indices", "challenge", "responses", "C_x_0", "C_x_1", "C_V", "C_y", "enc")}))
        for p, q in made:
            assert S.verify_presentation(d["params"], d["key"], d["ip"], q, strict=True)[0] == issuer.verify_presentation(p) == 0
        issuer.set_strict(False)
        for p, q in made:   # under the reference's statement a strict-mode proof is just a wrong proof
            assert S.verify_presentation(d["params"], d["key"], d["ip"], q, strict=False)[0] == issuer.verify_presentation(p) == 1
        issuer.set_strict(True)
        # another presentation's (valid) proof of encryption in place of the own one: the DLEQ rejects it
        if nsp:
            q = dict(made[0][1], enc=[dict(made[1][1]["enc"][0])] + made[0][1]["enc"][1:])
            assert S.verify_presentation(d["params"], d["key"], d["ip"], q, strict=True)[0] == 1
