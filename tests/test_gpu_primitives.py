"""GPU parity, K* rows: the HIP field/point/scalar kernels through the C ABI against the
ORACLE and the libsodium-computed golden vectors.  Bit-exact."""
import hashlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
H = bytes.fromhex


@pytest.fixture(scope="module")
def engine():
    import oracle
    import aeonflux_amd as afx
    st = hashlib.shake_256(b"gpu-prims").digest(1 << 15)
    params, used = oracle.system_parameters_generate(4, st)
    key, ip = oracle.issuer_new(params, st[used:used + 64 * 8])
    ctx = afx.Context(params, key, ip)
    yield afx, ctx
    ctx.close()


def arr(blobs):
    return np.frombuffer(b"".join(blobs), dtype=np.uint8).copy()


def test_from_uniform_and_reduce_wide(engine, primitives):
    import oracle
    afx, ctx = engine
    ins = [H(v["in"]) for v in primitives["from_uniform"]]
    rnd = hashlib.shake_256(b"more-uniform").digest(64 * 1000)
    ins += [rnd[64 * i:64 * i + 64] for i in range(1000)]
    a = arr(ins)
    out = np.zeros(32 * len(ins), np.uint8)
    afx.check(afx.lib().afx_points_from_uniform_bytes(ctx.h, a.ctypes.data, len(ins), out.ctypes.data))
    got = [bytes(out[32 * i:32 * i + 32]) for i in range(len(ins))]
    for v, g in zip(primitives["from_uniform"], got):
        assert g.hex() == v["out"]
    for i in range(len(primitives["from_uniform"]), len(ins)):
        assert got[i] == oracle.point_from_uniform(ins[i])
    wides = [H(v["in"]) for v in primitives["scalar_reduce_wide"]] + ins
    a = arr(wides)
    out = np.zeros(32 * len(wides), np.uint8)
    afx.check(afx.lib().afx_scalars_from_wide_bytes(ctx.h, a.ctypes.data, len(wides), out.ctypes.data))
    for i, v in enumerate(primitives["scalar_reduce_wide"]):
        assert bytes(out[32 * i:32 * i + 32]).hex() == v["out"]
    for i in range(len(primitives["scalar_reduce_wide"]), len(wides)):
        assert bytes(out[32 * i:32 * i + 32]) == oracle.scalar_reduce_wide(wides[i])


def test_decode_encode_validity(engine, primitives):
    afx, ctx = engine
    ins = [H(v["in"]) for v in primitives["validity"]]
    a = arr(ins)
    ok = np.zeros(len(ins), np.uint8)
    re = np.zeros(32 * len(ins), np.uint8)
    afx.check(afx.lib().afx_points_validate(ctx.h, a.ctypes.data, len(ins), ok.ctypes.data, re.ctypes.data))
    for i, v in enumerate(primitives["validity"]):
        assert bool(ok[i]) == v["valid"], v["in"]
        if v["valid"]:
            assert bytes(re[32 * i:32 * i + 32]).hex() == v["in"]


def msm(afx, ctx, scalars_rows, points_rows):
    """rows: [n_terms][count] of 32-byte values"""
    nt, cnt = len(scalars_rows), len(scalars_rows[0])
    s = arr([x for row in scalars_rows for x in row])
    p = arr([x for row in points_rows for x in row])
    out = np.zeros(32 * cnt, np.uint8)
    ok = np.zeros(cnt, np.uint8)
    afx.check(afx.lib().afx_multiscalar_mul(ctx.h, nt, s.ctypes.data, p.ctypes.data, cnt, out.ctypes.data, ok.ctypes.data))
    return [bytes(out[32 * i:32 * i + 32]) for i in range(cnt)], ok


def test_scalarmult_add_sub_msm_golden(engine, primitives):
    afx, ctx = engine
    L = 2**252 + 27742317777372353535851937790883648493
    one, minus_one = (1).to_bytes(32, "little"), (L - 1).to_bytes(32, "little")
    sm = primitives["scalarmult"]
    got, ok = msm(afx, ctx, [[H(v["s"]) for v in sm]], [[H(v["p"]) for v in sm]])
    assert ok.all() and [g.hex() for g in got] == [v["out"] for v in sm]
    ad = primitives["add"]
    got, ok = msm(afx, ctx, [[one] * len(ad), [one] * len(ad)], [[H(v["p"]) for v in ad], [H(v["q"]) for v in ad]])
    assert ok.all() and [g.hex() for g in got] == [v["out"] for v in ad]
    sb = primitives["sub"]
    got, ok = msm(afx, ctx, [[one] * len(sb), [minus_one] * len(sb)], [[H(v["p"]) for v in sb], [H(v["q"]) for v in sb]])
    assert ok.all() and [g.hex() for g in got] == [v["out"] for v in sb]
    for v in primitives["msm"]:
        got, ok = msm(afx, ctx, [[H(x)] for x in v["s"]], [[H(x)] for x in v["p"]])
        assert ok.all() and got[0].hex() == v["out"]
    bm = primitives["base_multiples"]
    got, ok = msm(afx, ctx, [[k.to_bytes(32, "little") for k in range(len(bm))]], [[H(bm[1])] * len(bm)])
    assert [g.hex() for g in got] == bm


def test_msm_random_vs_oracle_ragged_counts(engine):
    import oracle
    afx, ctx = engine
    rnd = hashlib.shake_256(b"msm-rand").digest(1 << 20)
    pts = [oracle.point_from_uniform(rnd[64 * i:64 * i + 64]) for i in range(64)]
    pos = 4096
    for nt, cnt in ((1, 1), (2, 63), (3, 64), (5, 65), (19, 257), (24, 300)):
        S = [[oracle.scalar_reduce_wide(rnd[pos + 64 * (k * cnt + i):pos + 64 * (k * cnt + i) + 64]) for i in range(cnt)] for k in range(nt)]
        pos += 64 * nt * cnt
        P = [[pts[(k * 7 + i) % 64] for i in range(cnt)] for k in range(nt)]
        got, ok = msm(afx, ctx, S, P)
        assert ok.all()
        for i in range(0, cnt, max(1, cnt // 9)):
            want = oracle.multiscalar([S[k][i] for k in range(nt)], [P[k][i] for k in range(nt)])
            assert got[i] == want, (nt, cnt, i)
    # undecodable point / non-canonical scalar are flagged, neighbours unaffected
    five = (5).to_bytes(32, "little")
    got, ok = msm(afx, ctx, [[five] * 3], [[oracle.basepoint(), b"\xff" * 32, oracle.basepoint()]])
    assert ok.tolist() == [1, 0, 1] and got[0] == got[2] == oracle.point_scalarmult(five, oracle.basepoint())
    got, ok = msm(afx, ctx, [[b"\xff" * 32, five]], [[oracle.basepoint()] * 2])
    assert ok.tolist() == [0, 1]


def test_rfc9496_appendix_a_in_full_on_the_gpu(engine, kat):
    """RFC 9496 A.1-A.3 through the engine's primitives: k * B for k = 0 .. 15, every invalid encoding rejected (and, being the
    reference's decompress, for the RFC's reason or another: only the verdict is contractual), the seven uniform byte strings"""
    afx, ctx = engine
    mult = kat["rfc9496_generator_multiples"]
    got, ok = msm(afx, ctx, [[k.to_bytes(32, "little") for k in range(16)]], [[H(mult[1])] * 16])
    assert ok.all() and [g.hex() for g in got] == mult
    # ... and as sums: (k - 1) * B + B with both terms variable
    got, ok = msm(afx, ctx, [[(1).to_bytes(32, "little")] * 15, [(1).to_bytes(32, "little")] * 15], [[H(mult[k - 1]) if k > 1 else H(mult[0]) for k in range(1, 16)], [H(mult[1])] * 15])
    assert ok.all() and [g.hex() for g in got] == mult[1:]
    bad = [e for encs in kat["rfc9496_bad_encodings"].values() for e in encs]
    ins = bad + mult
    a = arr([H(x) for x in ins])
    okv = np.zeros(len(ins), np.uint8)
    re = np.zeros(32 * len(ins), np.uint8)
    afx.check(afx.lib().afx_points_validate(ctx.h, a.ctypes.data, len(ins), okv.ctypes.data, re.ctypes.data))
    assert not okv[:len(bad)].any() and okv[len(bad):].all() and len(bad) == 29
    assert [bytes(re[32 * i:32 * i + 32]).hex() for i in range(len(bad), len(ins))] == mult
    fu = kat["rfc9496_from_uniform_bytes"]
    a = arr([H(v["in"]) for v in fu])
    out = np.zeros(32 * len(fu), np.uint8)
    afx.check(afx.lib().afx_points_from_uniform_bytes(ctx.h, a.ctypes.data, len(fu), out.ctypes.data))
    assert [bytes(out[32 * i:32 * i + 32]).hex() for i in range(len(fu))] == [v["out"] for v in fu]


def test_merlins_conformance_vectors_on_the_gpu(engine, kat):
    """merlin's two published transcript vectors through the GPU's own STROBE-128 / Keccak path (afx_merlin_challenges: StrobeSim ->
    k_hash, the path every statement's transcript takes).  equivalence_complex chains 32 challenges, each absorbed again: one program per
    challenge, the earlier ones fed back as per-item fields - the last program absorbs 34 KB (over 200 permutations, 1024-byte appends
    across the 166-byte rate) on the device.  Three items per call, all equal (every lane runs the same program), plus the oracle's
    scripted transcript for every intermediate challenge."""
    import oracle
    from tests.test_oracle_primitives import merlin_complex_ops
    afx, ctx = engine
    s = kat["merlin_equivalence_simple"]
    out = ctx.merlin_challenges(s["label"].encode(), [("append", s["append_label"].encode(), s["append_data"].encode()), ("challenge", s["challenge_label"].encode(), 32)], [], 3)
    assert all(bytes(out[i, :32]).hex() == s["challenge32"] for i in range(3))
    v = kat["merlin_equivalence_complex"]
    want = oracle.merlin_script(v["label"].encode(), merlin_complex_ops(v))
    big = bytes([v["big_byte"]]) * v["big_len"]
    count, chals = 3, []
    for r in range(v["rounds"]):
        ops = [("append", v["first_label"].encode(), v["first_data"].encode())]
        for k in range(r):    # the earlier rounds: their challenge operation (not returned), the big append, the challenge absorbed again
            ops += [("challenge", v["challenge_label"].encode(), 32), ("append", v["big_label"].encode(), big), ("append_field", v["feedback_label"].encode(), k)]
        ops.append(("challenge", v["challenge_label"].encode(), 32))
        fields = [np.tile(np.frombuffer(c, np.uint8), (count, 1)) for c in chals]
        out = ctx.merlin_challenges(v["label"].encode(), ops, fields, count)
        c = bytes(out[0, :32])
        assert all(bytes(out[i, :32]) == c for i in range(count)) and c == want[r], r
        chals.append(c)
    assert chals[-1].hex() == v["last_challenge32"]
    # a field that differs per item gives a challenge per item, each the oracle's
    f = np.frombuffer(hashlib.shake_256(b"merlin-fields").digest(32 * 70), np.uint8).reshape(70, 32)
    out = ctx.merlin_challenges(b"per item", [("append", b"a", b"constant"), ("append_field", b"val", 0), ("append_field", b"val2", 0), ("challenge", b"c", 64)], [f], 70)
    for i in range(70):
        assert bytes(out[i]) == oracle.merlin_script(b"per item", [("append", b"a", b"constant"), ("append_field", b"val", 0), ("append_field", b"val2", 0), ("challenge", b"c", 64)], [bytes(f[i])])[0]
    # malformed scripts are refused
    L = afx.lib()
    o = np.zeros((1, 64), np.uint8)
    for script in (b"", bytes([2]) + (0).to_bytes(4, "little"), bytes([1]) + (1).to_bytes(4, "little") + b"x",           # empty, no NEW, no challenge
                   bytes([1]) + (1).to_bytes(4, "little") + b"x" + bytes([4]) + (1).to_bytes(4, "little") + b"c" + (8).to_bytes(4, "little") + bytes([2]) + bytes(8),   # ends with an append
                   bytes([1]) + (1).to_bytes(4, "little") + b"x" + bytes([4]) + (1).to_bytes(4, "little") + b"c" + (65).to_bytes(4, "little"),   # 65 bytes
                   bytes([1]) + (1).to_bytes(4, "little") + b"x" + bytes([3]) + (1).to_bytes(4, "little") + b"v" + (0).to_bytes(4, "little")):   # field 0 of none
        assert L.afx_merlin_challenges(ctx.h, script, len(script), None, 0, 1, o.ctypes.data) == afx.E_BAD_ARGS
