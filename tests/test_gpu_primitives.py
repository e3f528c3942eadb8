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
