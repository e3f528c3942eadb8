#!/usr/bin/env python3
"""Generate tests/golden/*.json.

Runs ONLY in the build container (needs /opt/conda/lib/libsodium.so.23, libsodium 1.0.18, an
independent ristretto255 implementation).  The reference (/root/reference) is Rust and cannot be
built or imported here, and holds no golden vectors of its own, so:

  primitives.json   inputs + outputs computed by LIBSODIUM (and hashlib) — independent pins for the
                    oracle's and the HIP kernels' ristretto255 / scalar / SHA-512 arithmetic.
  kat.json          third-party known-answer values (RFC 9496 basepoint multiples + hash-to-group
                    vector re-derived through libsodium, merlin `equivalence_simple` transcript).
  flows.json        whole issue -> show -> verify transcripts for the reference's test layouts
                    (presentation.rs:461-638, issuance.rs:233-295, encryption.rs:222-244) computed by
                    the ORACLE after it passed the two files above: regression + GPU-parity inputs,
                    NOT an independent pin ("parity unpinned" for the statement layer).  Before it is written, every
                    flow is replayed through tests/pyref (a second restatement in pure Python, written
                    independently of oracle/): all bytes and decisions must agree, or generation fails.
"""
import ctypes as C
import hashlib
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import oracle  # noqa: E402

sod = C.CDLL("/opt/conda/lib/libsodium.so.23")
assert sod.sodium_init() >= 0
L = 2**252 + 27742317777372353535851937790883648493
P = 2**255 - 19


def rng(seed, n):
    return hashlib.shake_256(b"afx-golden/" + seed.encode()).digest(n)


def s_base(k):
    o = C.create_string_buffer(32)
    rc = sod.crypto_scalarmult_ristretto255_base(o, k)
    return o.raw if rc == 0 else bytes(32)


def s_mult(k, p):
    o = C.create_string_buffer(32)
    rc = sod.crypto_scalarmult_ristretto255(o, k, p)
    return o.raw if rc == 0 else None


def s_add(p, q):
    o = C.create_string_buffer(32)
    assert sod.crypto_core_ristretto255_add(o, p, q) == 0
    return o.raw


def s_sub(p, q):
    o = C.create_string_buffer(32)
    assert sod.crypto_core_ristretto255_sub(o, p, q) == 0
    return o.raw


def s_from_hash(h):
    o = C.create_string_buffer(32)
    assert sod.crypto_core_ristretto255_from_hash(o, h) == 0
    return o.raw


def s_valid(p):
    return sod.crypto_core_ristretto255_is_valid_point(p) == 1


def s_reduce(w):
    o = C.create_string_buffer(32)
    sod.crypto_core_ristretto255_scalar_reduce(o, w)
    return o.raw


def sc(x):
    return (x % L).to_bytes(32, "little")


def gen_primitives():
    out = {"_source": "libsodium 1.0.18 /opt/conda/lib/libsodium.so.23 + hashlib; see tests/gen_golden.py"}
    # basepoint multiples 0..16
    out["base_multiples"] = [s_base(sc(k)).hex() for k in range(17)]
    # from_uniform_bytes (crypto_core_ristretto255_from_hash == dalek from_uniform_bytes)
    r = rng("uniform", 64 * 32)
    out["from_uniform"] = [{"in": r[64 * i:64 * i + 64].hex(), "out": s_from_hash(r[64 * i:64 * i + 64]).hex()} for i in range(32)]
    edge = [bytes(64), b"\xff" * 64, b"\x01" + bytes(63), bytes(32) + b"\x01" + bytes(31)]
    out["from_uniform"] += [{"in": e.hex(), "out": s_from_hash(e).hex()} for e in edge]
    # scalar mult / add / sub on random points (clamp scalars below 2^255: libsodium masks bit 255)
    r = rng("arith", 64 * 64 + 32 * 64)
    pts = [s_from_hash(r[64 * i:64 * i + 64]) for i in range(64)]
    scs = [sc(int.from_bytes(r[4096 + 32 * i:4096 + 32 * i + 32], "little")) for i in range(64)]
    out["scalarmult"] = [{"s": scs[i].hex(), "p": pts[i].hex(), "out": s_mult(scs[i], pts[i]).hex()} for i in range(32)]
    for k in (0, 1, 2, L - 1, L - 2, 2**252, 8, 15, 16, 2**128):
        o = s_mult(sc(k), pts[0])
        out["scalarmult"].append({"s": sc(k).hex(), "p": pts[0].hex(), "out": (o or bytes(32)).hex()})
    out["add"] = [{"p": pts[i].hex(), "q": pts[i + 1].hex(), "out": s_add(pts[i], pts[i + 1]).hex()} for i in range(16)]
    out["add"].append({"p": pts[0].hex(), "q": pts[0].hex(), "out": s_add(pts[0], pts[0]).hex()})
    out["add"].append({"p": pts[0].hex(), "q": bytes(32).hex(), "out": s_add(pts[0], bytes(32)).hex()})
    out["sub"] = [{"p": pts[i].hex(), "q": pts[i + 1].hex(), "out": s_sub(pts[i], pts[i + 1]).hex()} for i in range(16)]
    out["sub"].append({"p": pts[0].hex(), "q": pts[0].hex(), "out": s_sub(pts[0], pts[0]).hex()})
    # small MSMs assembled from libsodium mult+add
    msm = []
    for n in (1, 2, 3, 4, 7, 19):
        acc = bytes(32)
        for k in range(n):
            acc = s_add(acc, s_mult(scs[k], pts[k]))
        msm.append({"s": [x.hex() for x in scs[:n]], "p": [x.hex() for x in pts[:n]], "out": acc.hex()})
    out["msm"] = msm
    # validity of encodings: crafted + random
    cand = []
    for v in (0, 1, 2, 3, P - 1, P, P + 1, P + 2, 2**255 - 1, 2**255 - 20, 2**256 - 1, 2**255, 2**255 + 2):
        cand.append((v % 2**256).to_bytes(32, "little"))
    r = rng("valid", 32 * 400)
    cand += [r[32 * i:32 * i + 32] for i in range(400)]
    cand += [bytes([b[0] & 0xFE]) + b[1:31] + bytes([b[31] & 0x7F]) for b in cand[-200:]]
    cand += pts[:8] + [s_base(sc(k)) for k in range(1, 9)]
    # libsodium 1.0.18 ignores bit 255 in its canonicity test; RFC 9496 §4.3.1 and dalek (whose
    # decompress re-encodes the masked field element and compares all 32 bytes) reject it.
    out["validity"] = [{"in": c.hex(), "valid": bool(s_valid(c)) and not (c[31] & 0x80)} for c in cand]
    # scalars
    r = rng("scalars", 64 * 40)
    wides = [r[64 * i:64 * i + 64] for i in range(40)] + [bytes(64), b"\xff" * 64, sc(L - 1) + bytes(32),
                                                         L.to_bytes(32, "little") + bytes(32), (L + 1).to_bytes(32, "little") + bytes(32)]
    out["scalar_reduce_wide"] = [{"in": w.hex(), "out": s_reduce(w).hex()} for w in wides]
    out["scalar_muladd"] = []
    for i in range(0, 39, 3):
        a, b, c = (int.from_bytes(s_reduce(wides[i + j]), "little") for j in range(3))
        out["scalar_muladd"].append({"a": sc(a).hex(), "b": sc(b).hex(), "c": sc(c).hex(), "out": sc(a * b + c).hex()})
    # sha512 -> hash_to_scalar / hash_to_group (symmetric.rs:138-139)
    msgs = [b"", b"abc", b"This is a tsunami alert test..", bytes(30), bytes(range(30)), b"x" * 200]
    out["sha512"] = [{"msg": m.hex(), "digest": hashlib.sha512(m).hexdigest(), "to_scalar": s_reduce(hashlib.sha512(m).digest()).hex(),
                      "to_group": s_from_hash(hashlib.sha512(m).digest()).hex()} for m in msgs]
    # encode_to_group (encoding.rs:56-70) through libsodium's validity test
    enc = []
    for m in [b"This is a tsunami alert test..", bytes(30), bytes([1]) + bytes(29), bytes(range(30)), b"\xff" * 30, b"short"]:
        b = bytearray(32)
        b[1:1 + len(m)] = m
        found = None
        for j in range(64):
            b[31] = j
            for i in range(128):
                b[0] = 2 * i
                if s_valid(bytes(b)):
                    found = (bytes(b), i + 128 * j)
                    break
            if found:
                break
        enc.append({"msg": m.hex(), "point": found[0].hex(), "counter": found[1]})
    out["encode_to_group"] = enc
    return out


def gen_kat():
    """Third-party known answers, written out as their sources print them (nothing here is computed): RFC 9496 Appendix A.1 (multiples
    of the generator 0 .. 15), A.2 (invalid encodings, by the reason the RFC gives) and A.3 (group elements from uniform byte
    strings: the inputs are the SHA-512 digests the RFC lists, the first one of the sentence it names), and merlin's two conformance
    tests (merlin src/transcript.rs tests::equivalence_simple and equivalence_complex [3P]).  tests/test_oracle_primitives.py,
    test_pyref_cross_check.py and test_gpu_primitives.py run all of them through the oracle, the Python restatement and the GPU."""
    return {
        "_source": "third-party KATs: RFC 9496 App. A.1-A.3, merlin tests::equivalence_simple / equivalence_complex",
        "rfc9496_B": "e2f2ae0a6abc4e71a884a961c500515f58e30b6aa582dd8db6a65945e08d2d76",
        "rfc9496_2B": "6a493210f7499cd17fecb510ae0cea23a110e8d5b901f8acadd3095c73a3b919",
        "rfc9496_hash_to_group": {
            "msg": "Ristretto is traditionally a short shot of espresso coffee",
            "out": "3066f82a1a747d45120d1740f14358531a8f04bbffe6a819f86dfe50f44a0a46"},
        "rfc9496_generator_multiples": [
         "0000000000000000000000000000000000000000000000000000000000000000",
         "e2f2ae0a6abc4e71a884a961c500515f58e30b6aa582dd8db6a65945e08d2d76",
         "6a493210f7499cd17fecb510ae0cea23a110e8d5b901f8acadd3095c73a3b919",
         "94741f5d5d52755ece4f23f044ee27d5d1ea1e2bd196b462166b16152a9d0259",
         "da80862773358b466ffadfe0b3293ab3d9fd53c5ea6c955358f568322daf6a57",
         "e882b131016b52c1d3337080187cf768423efccbb517bb495ab812c4160ff44e",
         "f64746d3c92b13050ed8d80236a7f0007c3b3f962f5ba793d19a601ebb1df403",
         "44f53520926ec81fbd5a387845beb7df85a96a24ece18738bdcfa6a7822a176d",
         "903293d8f2287ebe10e2374dc1a53e0bc887e592699f02d077d5263cdd55601c",
         "02622ace8f7303a31cafc63f8fc48fdc16e1c8c8d234b2f0d6685282a9076031",
         "20706fd788b2720a1ed2a5dad4952b01f413bcf0e7564de8cdc816689e2db95f",
         "bce83f8ba5dd2fa572864c24ba1810f9522bc6004afe95877ac73241cafdab42",
         "e4549ee16b9aa03099ca208c67adafcafa4c3f3e4e5303de6026e3ca8ff84460",
         "aa52e000df2e16f55fb1032fc33bc42742dad6bd5a8fc0be0167436c5948501f",
         "46376b80f409b29dc2b5f6f0c52591990896e5716f41477cd30085ab7f10301e",
         "e0c418f7c8d9c4cdd7395b93ea124f3ad99021bb681dfc3302a9d99a2e53e64e"
        ],
        "rfc9496_bad_encodings": {
         "non-canonical field encodings": [
          "00ffffffffffffffffffffffffffffffffffffffffffffffffffffffffffffff",
          "ffffffffffffffffffffffffffffffffffffffffffffffffffffffffffffff7f",
          "f3ffffffffffffffffffffffffffffffffffffffffffffffffffffffffffff7f",
          "edffffffffffffffffffffffffffffffffffffffffffffffffffffffffffff7f"
         ],
         "negative field elements": [
          "0100000000000000000000000000000000000000000000000000000000000000",
          "01ffffffffffffffffffffffffffffffffffffffffffffffffffffffffffff7f",
          "ed57ffd8c914fb201471d1c3d245ce3c746fcbe63a3679d51b6a516ebebe0e20",
          "c34c4e1826e5d403b78e246e88aa051c36ccf0aafebffe137d148a2bf9104562",
          "c940e5a4404157cfb1628b108db051a8d439e1a421394ec4ebccb9ec92a8ac78",
          "47cfc5497c53dc8e61c91d17fd626ffb1c49e2bca94eed052281b510b1117a24",
          "f1c6165d33367351b0da8f6e4511010c68174a03b6581212c71c0e1d026c3c72",
          "87260f7a2f12495118360f02c26a470f450dadf34a413d21042b43b9d93e1309"
         ],
         "non-square x^2": [
          "26948d35ca62e643e26a83177332e6b6afeb9d08e4268b650f1f5bbd8d81d371",
          "4eac077a713c57b4f4397629a4145982c661f48044dd3f96427d40b147d9742f",
          "de6a7b00deadc788eb6b6c8d20c0ae96c2f2019078fa604fee5b87d6e989ad7b",
          "bcab477be20861e01e4a0e295284146a510150d9817763caf1a6f4b422d67042",
          "2a292df7e32cababbd9de088d1d1abec9fc0440f637ed2fba145094dc14bea08",
          "f4a9e534fc0d216c44b218fa0c42d99635a0127ee2e53c712f70609649fdff22",
          "8268436f8c4126196cf64b3c7ddbda90746a378625f9813dd9b8457077256731",
          "2810e5cbc2cc4d4eece54f61c6f69758e289aa7ab440b3cbeaa21995c2f4232b"
         ],
         "negative xy value": [
          "3eb858e78f5a7254d8c9731174a94f76755fd3941c0ac93735c07ba14579630e",
          "a45fdc55c76448c049a1ab33f17023edfb2be3581e9c7aade8a6125215e04220",
          "d483fe813c6ba647ebbfd3ec41adca1c6130c2beeee9d9bf065c8d151c5f396e",
          "8a2e1d30050198c65a54483123960ccc38aef6848e1ec8f5f780e8523769ba32",
          "32888462f8b486c68ad7dd9610be5192bbeaf3b443951ac1a8118419d9fa097b",
          "227142501b9d4355ccba290404bde41575b037693cef1f438c47f8fbf35d1165",
          "5c37cc491da847cfeb9281d407efc41e15144c876e0170b499a96a22ed31e01e",
          "445425117cb8c90edcbc7c1cc0e74f747f2c1efa5630a967c64f287792a48a4b"
         ],
         "s = -1, which causes y = 0": [
          "ecffffffffffffffffffffffffffffffffffffffffffffffffffffffffffff7f"
         ]
        },
        "rfc9496_from_uniform_bytes": [
         {
          "in": "5d1be09e3d0c82fc538112490e35701979d99e06ca3e2b5b54bffe8b4dc772c14d98b696a1bbfb5ca32c436cc61c16563790306c79eaca7705668b47dffe5bb6",
          "out": "3066f82a1a747d45120d1740f14358531a8f04bbffe6a819f86dfe50f44a0a46"
         },
         {
          "in": "f116b34b8f17ceb56e8732a60d913dd10cce47a6d53bee9204be8b44f6678b270102a56902e2488c46120e9276cfe54638286b9e4b3cdb470b542d46c2068d38",
          "out": "f26e5b6f7d362d2d2a94c5d0e7602cb4773c95a2e5c31a64f133189fa76ed61b"
         },
         {
          "in": "8422e1bbdaab52938b81fd602effb6f89110e1e57208ad12d9ad767e2e25510c27140775f9337088b982d83d7fcf0b2fa1edffe51952cbe7365e95c86eaf325c",
          "out": "006ccd2a9e6867e6a2c5cea83d3302cc9de128dd2a9a57dd8ee7b9d7ffe02826"
         },
         {
          "in": "ac22415129b61427bf464e17baee8db65940c233b98afce8d17c57beeb7876c2150d15af1cb1fb824bbd14955f2b57d08d388aab431a391cfc33d5bafb5dbbaf",
          "out": "f8f0c87cf237953c5890aec3998169005dae3eca1fbb04548c635953c817f92a"
         },
         {
          "in": "165d697a1ef3d5cf3c38565beefcf88c0f282b8e7dbd28544c483432f1cec7675debea8ebb4e5fe7d6f6e5db15f15587ac4d4d4a1de7191e0c1ca6664abcc413",
          "out": "ae81e7dedf20a497e10c304a765c1767a42d6e06029758d2d7e8ef7cc4c41179"
         },
         {
          "in": "a836e6c9a9ca9f1e8d486273ad56a78c70cf18f0ce10abb1c7172ddd605d7fd2979854f47ae1ccf204a33102095b4200e5befc0465accc263175485f0e17ea5c",
          "out": "e2705652ff9f5e44d3e841bf1c251cf7dddb77d140870d1ab2ed64f1a9ce8628"
         },
         {
          "in": "2cdc11eaeb95daf01189417cdddbf95952993aa9cb9c640eb5058d09702c74622c9965a697a3b345ec24ee56335b556e677b30e6f90ac77d781064f866a3c982",
          "out": "80bd07262511cdde4863f8a7434cef696750681cb9510eea557088f76d9e5065"
         }
        ],
        "merlin_equivalence_simple": {
            "label": "test protocol", "append_label": "some label", "append_data": "some data", "challenge_label": "challenge",
            "challenge32": "d5a21972d0d5fe320c0d263fac7fffb8145aa640af6e9bca177c03c7efcf0615"},
        # Transcript::new("test protocol"); append_message("step1", "some data"); then 32 times: challenge_bytes("challenge", 32 bytes),
        # append_message("bigdata", 1024 bytes of 99), append_message("challengedata", that challenge); the LAST challenge is the vector
        # (1024-byte appends cross six blocks of the 166-byte rate; each challenge is absorbed again)
        "merlin_equivalence_complex": {
            "label": "test protocol", "first_label": "step1", "first_data": "some data", "rounds": 32, "challenge_label": "challenge",
            "big_label": "bigdata", "big_byte": 99, "big_len": 1024, "feedback_label": "challengedata",
            "last_challenge32": "a8c933f54fae76e3f9bea93648c1308e7dfa2152dd51674ff3ca438351cf003c"},
    }


# ---- flows (oracle-generated) ----
PS, SS, PP, EP, SP = range(5)


def pad96(b):
    return b + bytes(96 - len(b))


def flow(name, n, layout, hide, with_key=True, tamper=None, seed=None):
    """layout: list of 'S' (scalar), 'P' (point), 'E' (plaintext) at issuance; hide: indices hidden for show."""
    st = hashlib.shake_256(b"afx-flow/" + (seed or name).encode()).digest(1 << 16)
    pos = [0]

    def take(k):
        b = st[pos[0]:pos[0] + k]
        pos[0] += k
        return b
    params, used = oracle.system_parameters_generate(n, st)
    pos[0] = used
    key, ip = oracle.issuer_new(params, take(64 * (4 + n)))
    issuer = oracle.Ctx(params, key, ip)
    user = oracle.Ctx(params, None, ip)
    kinds, values = [], []
    for c in layout:
        if c == "S":
            kinds.append(PS); values.append(pad96(oracle.scalar_reduce_wide(take(64))))
        elif c == "P":
            kinds.append(PP); values.append(pad96(oracle.point_from_uniform(take(64))))
        elif c == "E":
            kinds.append(EP); values.append(oracle.plaintext_from_bytes(take(30))[0])
        elif c == "0":  # the all-zero message: M1 is the identity (issuance.rs:272-295)
            kinds.append(EP); values.append(oracle.plaintext_from_bytes(bytes(30))[0])
    t_wide, U_wide, iseed = take(64), take(64), take(32)
    ist, t, U, V, ch, resp = issuer.issue(kinds, values, t_wide, U_wide, iseed)
    rec = {"name": name, "n": n, "params": params.hex(), "key": key.hex(), "issuer_params": ip.hex(),
           "issue": {"kinds": kinds, "values": [v.hex() for v in values], "t_wide": t_wide.hex(), "U_wide": U_wide.hex(),
                     "rng_seed": iseed.hex(), "status": ist, "t": t.hex(), "U": U.hex(), "V": V.hex(), "challenge": ch.hex(),
                     "responses": [r.hex() for r in resp]}}
    if ist != 0:
        return rec
    rec["issuance_verify"] = user.issuance_verify(kinds, values, t, U, V, ch, resp)
    commits, c2 = oracle.debug_last()
    rec["issuance_commitments"] = [c.hex() for c in commits]
    skinds = list(kinds)
    for i in hide:
        if skinds[i] == PS:
            skinds[i] = SS
        elif skinds[i] == EP:
            skinds[i] = SP
    master = take(64)
    kp = user.keypair_derive(master)
    if tamper == "swap_attr0":  # presentation.rs:618-638
        values = list(values)
        values[0] = pad96(oracle.scalar_reduce_wide(take(64)))
    z_wide, sseed = take(64), take(32)
    nsp = sum(1 for k in skinds if k == SP)
    eseeds = take(32 * nsp)
    sst, p = user.show(skinds, values, t, U, V, kp if with_key else None, z_wide, sseed, eseeds)
    show = {"kinds": skinds, "values": [v.hex() for v in values], "master_secret": master.hex(), "keypair": kp.hex() if with_key else None,
            "z_wide": z_wide.hex(), "rng_seed": sseed.hex(), "enc_seeds": eseeds.hex(), "status": sst}
    rec["show"] = show
    if sst != 0:
        return rec
    pres = {"n_attributes": p.n_attributes, "n_responses": p.n_responses, "challenge": bytes(p.challenge).hex(),
            "responses": [bytes(p.responses[i]).hex() for i in range(p.n_responses)], "C_x_0": bytes(p.C_x_0).hex(),
            "C_x_1": bytes(p.C_x_1).hex(), "C_V": bytes(p.C_V).hex(), "C_y": [bytes(p.C_y[i]).hex() for i in range(p.n_attributes)],
            "kinds": list(p.kinds[:p.n_attributes]), "attr_values": [bytes(p.attr_values[i]).hex() for i in range(p.n_attributes)],
            "hidden_scalar_indices": list(p.hidden_scalar_indices[:p.n_hidden_scalars]), "enc": []}
    for e in range(p.n_enc_proofs):
        q = p.enc[e]
        pres["enc"].append({"challenge": bytes(q.challenge).hex(), "responses": [bytes(q.responses[i]).hex() for i in range(6)],
                            "pk": bytes(q.pk).hex(), "E1": bytes(q.E1).hex(), "E2": bytes(q.E2).hex(), "C_y_1": bytes(q.C_y_1).hex(),
                            "C_y_2": bytes(q.C_y_2).hex(), "C_y_3": bytes(q.C_y_3).hex(), "C_y_2p": bytes(q.C_y_2p).hex(), "index": q.index})
    rec["presentation"] = pres
    rec["verify"] = issuer.verify_presentation(p)
    commits, c2 = oracle.debug_last()
    rec["verify_last_commitments"] = [c.hex() for c in commits]
    return rec


def gen_flows():
    flows = [
        flow("readme_4attrs_sSPe", 4, "SSPE", [0, 3]),                                # README.md:61-116 (BASELINE config 1/2 shape)
        flow("credential_proof_10_attributes", 10, "PPSSPSPSSP", []),                 # presentation.rs:461-489
        flow("credential_proof_10_attributes_with_plaintext", 10, "EPSSPSPSSP", [2]),  # presentation.rs:492-525
        flow("credential_proof_1_plaintext", 1, "E", []),                             # presentation.rs:528-542
        flow("credential_proof_1_plaintext_hidden", 1, "E", [0]),                     # presentation.rs:545-566
        flow("credential_proof_1_scalar_revealed", 1, "S", [], with_key=False),       # presentation.rs:569-584
        flow("switch_scalar_point", 2, "SP", [], with_key=False),                     # presentation.rs:587-616
        flow("switch_point_scalar", 2, "PS", [], with_key=False),
        flow("bad_credential_proof_1_scalar_revealed", 1, "S", [], with_key=False, tamper="swap_attr0"),  # presentation.rs:618-638
        flow("issuance_proof", 3, "SSP", []),                                         # issuance.rs:233-248
        flow("issuance_proof_with_plaintext", 5, "SSPES", []),                        # issuance.rs:251-269
        flow("issuance_proof_identity_plaintext", 6, "0SSPPS", []),                   # issuance.rs:272-295 (must FAIL issuance verify)
        flow("c3_8attrs_SSPPeeee", 8, "SSPPEEEE", [4, 5, 6, 7]),                      # BASELINE config 3/4 shape
        flow("leading_hidden_point_fails", 3, "ESS", [0]),                            # honest proof rejected
        flow("no_symmetric_key", 2, "ES", [0], with_key=False),                       # presentation.rs:150-157
        flow("c5_16attrs", 16, "SSSSSSSSPPPPEEEE", []),                               # BASELINE config 5 shape (issue)
        flow("hidden_scalars_mixed", 6, "SSSPSE", [0, 2, 4, 5]),
    ]
    return {"_source": "computed by the ORACLE (oracle/*.c) after primitives.json/kat.json passed; self-consistency "
                       "fixtures and GPU parity inputs, not an independent pin", "flows": flows}


if __name__ == "__main__":
    os.makedirs(os.path.join(HERE, "golden"), exist_ok=True)
    for name, fn in (("primitives", gen_primitives), ("kat", gen_kat), ("flows", gen_flows)):
        data = fn()
        if name == "flows":
            # the second restatement must reproduce every byte and decision before the fixture is accepted
            from tests.test_pyref_cross_check import test_every_flow_replays_byte_for_byte
            test_every_flow_replays_byte_for_byte(data["flows"])
            data["_source"] += "; every flow cross-checked byte for byte against tests/pyref (independent pure-Python restatement)"
            # ... and libsodium must recompute every group operation of that replay (tests/sodium_replay.py): the flows' group
            # values are then pinned by a third implementation; labels / order / constraint lists by the two restatements only
            from tests import sodium_replay
            ops = sodium_replay.replay_flows(sod, data["flows"])
            data["_source"] += "; every group operation of that replay recomputed with libsodium 1.0.18 (%d multiscalar sums, %d scalar " \
                               "multiplications, %d additions/subtractions/negations)" % (ops["msm"], ops["mul"], ops["add"] + ops["sub"] + ops["neg"])
        with open(os.path.join(HERE, "golden", name + ".json"), "w") as f:
            json.dump(data, f, indent=1)
        print("wrote", name)
