"""Pin the ORACLE's primitives against independent values: libsodium-computed vectors (primitives.json),
RFC 9496 / merlin KATs (kat.json) and hashlib.  CPU only."""
import hashlib

import oracle

H = bytes.fromhex
L = 2**252 + 27742317777372353535851937790883648493


def test_rfc9496_kats(kat):
    assert oracle.basepoint().hex() == kat["rfc9496_B"]
    assert oracle.point_scalarmult((2).to_bytes(32, "little"), oracle.basepoint()).hex() == kat["rfc9496_2B"]
    v = kat["rfc9496_hash_to_group"]
    assert oracle.point_from_uniform(hashlib.sha512(v["msg"].encode()).digest()).hex() == v["out"]


def test_merlin_kat(kat):
    v = kat["merlin_equivalence_simple"]
    out = oracle.merlin_simple(v["label"].encode(), v["append_label"].encode(), v["append_data"].encode(),
                               v["challenge_label"].encode(), 32)
    assert out.hex() == v["challenge32"]


def test_sha512_and_keccak_against_hashlib():
    for m in (b"", b"abc", b"a" * 111, b"a" * 112, b"a" * 127, b"a" * 128, b"a" * 129, bytes(range(256)) * 5):
        assert oracle.sha512(m) == hashlib.sha512(m).digest()

    def sha3_256(msg):
        rate, st, m = 136, bytearray(200), bytearray(msg) + b"\x06"
        m += bytes(-len(m) % rate)
        m[-1] |= 0x80
        for o in range(0, len(m), rate):
            for i in range(rate):
                st[i] ^= m[o + i]
            st = bytearray(oracle.keccak_f1600(bytes(st)))
        return bytes(st[:32])
    for m in (b"", b"abc", b"q" * 135, b"q" * 136, b"q" * 500):
        assert sha3_256(m) == hashlib.sha3_256(m).digest()


def test_base_multiples(primitives):
    B = oracle.basepoint()
    for k, want in enumerate(primitives["base_multiples"]):
        assert oracle.point_scalarmult(k.to_bytes(32, "little"), B).hex() == want
        assert oracle.multiscalar([k.to_bytes(32, "little")], [B], vartime=True).hex() == want


def test_from_uniform(primitives):
    for v in primitives["from_uniform"]:
        assert oracle.point_from_uniform(H(v["in"])).hex() == v["out"]


def test_scalarmult_add_sub(primitives):
    for v in primitives["scalarmult"]:
        assert oracle.point_scalarmult(H(v["s"]), H(v["p"])).hex() == v["out"]
        assert oracle.multiscalar([H(v["s"])], [H(v["p"])], vartime=True).hex() == v["out"]
    for v in primitives["add"]:
        assert oracle.point_add(H(v["p"]), H(v["q"])).hex() == v["out"]
    for v in primitives["sub"]:
        assert oracle.point_sub(H(v["p"]), H(v["q"])).hex() == v["out"]


def test_msm_both_schedules(primitives):
    for v in primitives["msm"]:
        s, p = [H(x) for x in v["s"]], [H(x) for x in v["p"]]
        assert oracle.multiscalar(s, p, vartime=False).hex() == v["out"]
        assert oracle.multiscalar(s, p, vartime=True).hex() == v["out"]


def test_encoding_validity(primitives):
    n_valid = 0
    for v in primitives["validity"]:
        out = oracle.point_decode_encode(H(v["in"]))
        assert (out is not None) == v["valid"], v["in"]
        if out is not None:
            n_valid += 1
            assert out.hex() == v["in"]  # compress(decompress(b)) == b for every valid encoding
    assert n_valid > 20


def test_scalars(primitives):
    for v in primitives["scalar_reduce_wide"]:
        assert oracle.scalar_reduce_wide(H(v["in"])).hex() == v["out"]
    for v in primitives["scalar_muladd"]:
        assert oracle.scalar_muladd(H(v["a"]), H(v["b"]), H(v["c"])).hex() == v["out"]
    assert oracle.lib().afxo_scalar_is_canonical((L - 1).to_bytes(32, "little")) == 1
    assert oracle.lib().afxo_scalar_is_canonical(L.to_bytes(32, "little")) == 0
    assert oracle.lib().afxo_scalar_is_canonical(b"\xff" * 32) == 0
    assert oracle.scalar_neg(bytes(32)) == bytes(32)


def test_hash_to_scalar_group_and_encode(primitives):
    for v in primitives["sha512"]:
        d = oracle.sha512(H(v["msg"]))
        assert d.hex() == v["digest"]
        assert oracle.scalar_reduce_wide(d).hex() == v["to_scalar"]
        assert oracle.point_from_uniform(d).hex() == v["to_group"]
    for v in primitives["encode_to_group"]:
        pt, ctr = oracle.encode_to_group(H(v["msg"]))
        assert ctr == v["counter"]
        if any(H(v["point"])):  # the identity re-encodes as itself
            assert pt.hex() == v["point"]
        data, ctr2 = oracle.decode_from_group(H(v["point"]))
        assert ctr2 == ctr and data[:len(H(v["msg"]))] == H(v["msg"])


def test_rfc9496_appendix_a_in_full(kat):
    """RFC 9496 A.1 (0 .. 15 times the generator), A.2 (every invalid encoding is rejected) and A.3 (the seven uniform byte strings)"""
    B = oracle.basepoint()
    mult = kat["rfc9496_generator_multiples"]
    assert len(mult) == 16 and mult[0] == "00" * 32 and mult[1] == B.hex()
    acc = None
    for k in range(1, 16):
        assert oracle.point_scalarmult(k.to_bytes(32, "little"), B).hex() == mult[k], k
        acc = B if acc is None else oracle.point_add(acc, B)          # ... and by repeated addition
        assert acc.hex() == mult[k], k
        assert oracle.point_decode_encode(H(mult[k])) == H(mult[k])
    assert oracle.point_sub(B, B).hex() == mult[0]
    n = 0
    for reason, encs in kat["rfc9496_bad_encodings"].items():
        for e in encs:
            assert oracle.point_decode_encode(H(e)) is None, (reason, e)
            n += 1
    assert n == 29
    for v in kat["rfc9496_from_uniform_bytes"]:
        assert oracle.point_from_uniform(H(v["in"])).hex() == v["out"]
    assert len(kat["rfc9496_from_uniform_bytes"]) == 7


def merlin_complex_ops(v):
    ops = [("append", v["first_label"].encode(), v["first_data"].encode())]
    for _ in range(v["rounds"]):
        ops += [("challenge", v["challenge_label"].encode(), 32), ("append", v["big_label"].encode(), bytes([v["big_byte"]]) * v["big_len"]),
                ("append_last_challenge", v["feedback_label"].encode())]
    return ops


def test_merlin_multi_block_kat(kat):
    """merlin's equivalence_complex: 1024-byte appends across the 166-byte rate, 32 chained challenges (oracle/hashes.c through
    afxo_merlin_script); the simple vector through the same door"""
    v = kat["merlin_equivalence_complex"]
    chals = oracle.merlin_script(v["label"].encode(), merlin_complex_ops(v))
    assert len(chals) == 32 and len(set(chals)) == 32 and chals[-1].hex() == v["last_challenge32"]
    s = kat["merlin_equivalence_simple"]
    got = oracle.merlin_script(s["label"].encode(), [("append", s["append_label"].encode(), s["append_data"].encode()), ("challenge", s["challenge_label"].encode(), 32)])
    assert got[0].hex() == s["challenge32"]
    # per-item fields are messages like any other
    f = bytes(range(32))
    a = oracle.merlin_script(b"x", [("append_field", b"val", 0), ("challenge", b"c", 64)], [f])
    b = oracle.merlin_script(b"x", [("append", b"val", f), ("challenge", b"c", 64)])
    assert a == b and len(a[0]) == 64
