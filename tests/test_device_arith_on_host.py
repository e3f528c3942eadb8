"""CPU-only: the engine's DEVICE arithmetic headers (aeonflux_amd/csrc/fe.cuh, sc.cuh, ge.cuh) compiled for the host
(tests/hostsim/arith_host.cpp) and compared with the oracle and with Python integers.  This is the arithmetic the
kernels run, statement for statement, so a slip in a carry chain or a bound shows up here without a GPU.  The same
build measures the field-operation counts that DESIGN.md section 3 publishes."""
import ctypes as C
import hashlib
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = 2 ** 255 - 19
L = 2 ** 252 + 27742317777372353535851937790883648493


@pytest.fixture(scope="module")
def arith(tmp_path_factory):
    out = str(tmp_path_factory.mktemp("arith") / "libarith_host.so")
    cmd = ["g++", "-O2", "-fPIC", "-shared", "-std=c++17", "-I" + os.path.join(ROOT, "tests", "hostsim", "include"), "-o", out,
           os.path.join(ROOT, "tests", "hostsim", "arith_host.cpp")]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    lib = C.CDLL(out)
    lib.arith_bounds_violations.restype = C.c_uint64
    lib.arith_last_violation.restype = C.c_char_p
    return lib


def stream(seed, n):
    return hashlib.shake_256(seed).digest(n)


def test_field_ops_against_python_integers(arith):
    s = stream(b"fe-host", 64 * 400)
    edge = [bytes(32), (1).to_bytes(32, "little"), (P - 1).to_bytes(32, "little"), (P).to_bytes(32, "little"), b"\xff" * 32,
            (2 ** 255 - 1).to_bytes(32, "little"), (2 ** 254).to_bytes(32, "little"), (19).to_bytes(32, "little")]
    cases = [(s[64 * i:64 * i + 32], s[64 * i + 32:64 * i + 64]) for i in range(400)] + [(a, b) for a in edge for b in edge]
    out = ((C.c_uint8 * 32) * 4)()
    for a, b in cases:
        arith.arith_fe(out, a, b)
        x, y = int.from_bytes(a, "little") % 2 ** 255, int.from_bytes(b, "little") % 2 ** 255
        got = [int.from_bytes(bytes(o), "little") for o in out]
        assert got[0] == x * y % P and got[1] == x * x % P
        assert got[2] == pow(x, P - 2, P) and got[3] == pow(x, (P - 5) // 8, P)


def test_lazily_added_operands_at_the_documented_bounds(arith):
    s = stream(b"fe-lazy", 192 * 300)
    big = (2 ** 255 - 20).to_bytes(32, "little")
    cases = [[s[192 * i + 32 * k:192 * i + 32 * k + 32] for k in range(6)] for i in range(300)] + [[big] * 4 + [big, bytes(32)], [big] * 4 + [bytes(32), big]]
    out = ((C.c_uint8 * 32) * 2)()
    for c in cases:
        a = ((C.c_uint8 * 32) * 4)(*[(C.c_uint8 * 32)(*x) for x in c[:4]])
        b = ((C.c_uint8 * 32) * 2)(*[(C.c_uint8 * 32)(*x) for x in c[4:]])
        arith.arith_fe_lazy(out, a, b)
        v = [int.from_bytes(x, "little") % 2 ** 255 for x in c]
        assert int.from_bytes(bytes(out[0]), "little") == (v[0] + v[1] + v[2] + v[3]) * (v[4] - v[5]) % P
        assert int.from_bytes(bytes(out[1]), "little") == (v[0] - v[1]) ** 2 % P


def _limbs_value(l):
    return sum(int(x) << (29 * i) for i, x in enumerate(l))


def test_products_at_the_limits_of_the_bounds_discipline(arith):
    """fe.cuh: a product needs |f| * |g| <= 3.8 units (1 unit = 2^29; limb 8: 2^23), a squaring |f| <= 1.9.  Operands with every limb at
    the extreme of its range, all sign patterns that maximise a column, must come out right and must not trip the checker."""
    import random
    rnd = random.Random(9)
    unit = [1 << 29] * 8 + [1 << 23]
    out = ((C.c_uint8 * 32) * 5)()
    raw, cen = (C.c_int32 * 9)(), (C.c_int32 * 9)()
    before = arith.arith_bounds_violations()

    def limbs(mag, mode):
        v = []
        for i in range(9):
            top = int(unit[i] * mag) - 1
            x = top if mode != "random" or rnd.random() < 0.5 else rnd.randrange(0, top + 1)
            sign = {"pos": 1, "neg": -1, "alt": (-1) ** i, "random": rnd.choice((1, -1))}[mode]
            v.append(sign * x)
        return v
    cases = []
    for ma, mb in ((1, 1), (2, 1), (2, 1.9), (3.8, 1), (1, 3.8), (1.9, 1.99), (3, 1.25), (1.5, 2)):
        for sa in ("pos", "neg", "alt", "random"):
            for sb in ("pos", "neg", "alt", "random"):
                cases.append((limbs(ma, sa), limbs(mb, sb)))
    for _ in range(300):
        ma = rnd.choice((1, 2, 3.8))
        cases.append((limbs(ma, "random"), limbs(3.8 / ma if ma > 1 else 1, "random")))
    for f, g in cases:
        # the squaring takes |f| <= 1.9: scale f down for it when the case is a wider operand
        fs = f if max(abs(x) / u for x, u in zip(f, unit)) <= 1.9 else [x // 2 for x in f]
        arith.arith_fe_limbs(out, (C.c_int32 * 9)(*fs), (C.c_int32 * 9)(*g), 2 | 4)
        vs = _limbs_value(fs)
        assert int.from_bytes(bytes(out[2]), "little") == vs * vs % P and bytes(out[3]) == bytes(out[2])
        assert int.from_bytes(bytes(out[4]), "little") == vs % P
        arith.arith_fe_limbs(out, (C.c_int32 * 9)(*f), (C.c_int32 * 9)(*g), 1)
        want = _limbs_value(f) * _limbs_value(g) % P
        assert int.from_bytes(bytes(out[0]), "little") == want and int.from_bytes(bytes(out[1]), "little") == want
        # ranges of the results: raw in [0, 1), centred in [-1/2, 1/2]; limb 1 takes the wrap's last carry (|.| < 2^15) on top
        arith.arith_fe_mul_limbs(raw, cen, (C.c_int32 * 9)(*f), (C.c_int32 * 9)(*g))
        for i in range(9):
            slack = (1 << 17) if i == 1 else 0
            assert -slack <= raw[i] < unit[i] + slack, (i, raw[i])
            assert -(unit[i] >> 1) - slack <= cen[i] <= (unit[i] >> 1) + slack, (i, cen[i])
    assert arith.arith_bounds_violations() == before, arith.arith_last_violation()


def test_canonical_encoding_of_edge_values(arith):
    """fe_tobytes on limb patterns around 0, p and 2^255, positive and negative, lazily added, up to the int32 range"""
    out = ((C.c_uint8 * 32) * 5)()
    p_limbs = [(1 << 29) - 19] + [(1 << 29) - 1] * 7 + [(1 << 23) - 1]
    vals = [[0] * 9, [1] + [0] * 8, [-1] + [0] * 8, [-19] + [0] * 8, [-20] + [0] * 8, p_limbs, [x + (i == 0) for i, x in enumerate(p_limbs)],
            [x - (i == 0) for i, x in enumerate(p_limbs)], [-x for x in p_limbs], [2 * x for x in p_limbs], [-2 * x for x in p_limbs],
            [(1 << 29) - 1] * 8 + [(1 << 23) - 1], [18] + [0] * 8, [19] + [0] * 8, [0] * 8 + [1 << 23], [0] * 8 + [-(1 << 23)],
            [3 * x for x in p_limbs], [(1 << 31) - 1] * 9, [-(1 << 31)] * 9, [(-1) ** i * ((1 << 31) - 1) for i in range(9)]]
    zero = (C.c_int32 * 9)(*([0] * 9))
    for v in vals:
        arith.arith_fe_limbs(out, (C.c_int32 * 9)(*v), zero, 4)
        got = int.from_bytes(bytes(out[4]), "little")
        assert got == _limbs_value(v) % P and got < P, v


def test_group_ops_against_the_oracle(arith, primitives):
    import oracle
    s = stream(b"ge-host", 64 * 3 * 40)
    o = [(C.c_uint8 * 32)() for _ in range(4)]
    for i in range(40):
        p = oracle.point_from_uniform(s[192 * i:192 * i + 64])
        q = oracle.point_from_uniform(s[192 * i + 64:192 * i + 128])
        k = oracle.scalar_reduce_wide(s[192 * i + 128:192 * i + 192])
        assert arith.arith_point_ops(o[0], o[1], o[2], o[3], k, p, q) == 1
        assert bytes(o[0]) == oracle.point_scalarmult(k, p)
        assert bytes(o[1]) == oracle.point_add(p, q) and bytes(o[2]) == oracle.point_sub(p, q)
        assert bytes(o[3]) == oracle.point_add(p, p)
        arith.arith_from_uniform(o[0], s[192 * i:192 * i + 64])
        assert bytes(o[0]) == p
    # committed fixtures (libsodium / RFC 9496): generator multiples round-trip, invalid encodings are refused,
    # hash-to-group and scalar multiplication vectors
    for enc in primitives["base_multiples"]:
        assert arith.arith_decode_encode(o[0], bytes.fromhex(enc)) == 1 and bytes(o[0]).hex() == enc
    for v in primitives["validity"]:
        assert arith.arith_decode_encode(o[0], bytes.fromhex(v["in"])) == (1 if v["valid"] else 0), v["in"]
    for v in primitives["from_uniform"]:
        arith.arith_from_uniform(o[0], bytes.fromhex(v["in"]))
        assert bytes(o[0]).hex() == v["out"]
    for v in primitives["scalarmult"]:
        p = bytes.fromhex(v["p"])
        assert arith.arith_point_ops(o[0], o[1], o[2], o[3], bytes.fromhex(v["s"]), p, p) == 1 and bytes(o[0]).hex() == v["out"]


def test_scalar_ops_against_python_integers(arith):
    s = stream(b"sc-host", 160 * 300)
    red, ma, ng = ((C.c_uint8 * 32)() for _ in range(3))
    can = C.c_int(0)
    edge = [(L - 1).to_bytes(32, "little"), L.to_bytes(32, "little"), bytes(32), b"\xff" * 32]
    for i in range(300):
        wide = s[160 * i:160 * i + 64]
        a, b, c = (s[160 * i + 64 + 32 * k:160 * i + 96 + 32 * k] for k in range(3))
        if i < 4:
            a = edge[i]
            wide = b"\xff" * 64 if i == 0 else wide
        ai = int.from_bytes(a, "little")
        # sc_muladd / sc_neg take canonical inputs (the kernels check canonicity first); reduce them here
        a_r, b_r, c_r = ((int.from_bytes(x, "little") % L).to_bytes(32, "little") for x in (a, b, c))
        arith.arith_sc(red, ma, ng, C.byref(can), wide, a_r, b_r, c_r)
        assert int.from_bytes(bytes(red), "little") == int.from_bytes(wide, "little") % L
        av, bv, cv = (int.from_bytes(x, "little") for x in (a_r, b_r, c_r))
        assert int.from_bytes(bytes(ma), "little") == (av * bv + cv) % L
        assert int.from_bytes(bytes(ng), "little") == (-av) % L
        arith.arith_sc(red, ma, ng, C.byref(can), wide, a, b_r, c_r)
        assert can.value == (1 if ai < L else 0)
        half, dbl = (C.c_uint8 * 32)(), (C.c_uint8 * 32)()
        arith.arith_sc_half_dbl(half, dbl, a_r)
        assert int.from_bytes(bytes(half), "little") * 2 % L == av and int.from_bytes(bytes(half), "little") < L
        assert int.from_bytes(bytes(dbl), "little") == 2 * av % L


def test_msm_chain_with_consumer_aware_conversions(arith):
    """k_msm's chain, step for step (raw / centred conversions chosen by what consumes them), against the oracle's
    multiscalar multiplication; every fe_mul / fe_sq of the host build asserts its operand bounds on the way"""
    import oracle
    s = stream(b"chain-host", 64 * 12 * 12)
    out = (C.c_uint8 * 32)()
    pos = 0
    for case in range(12):
        nv, nf = 1 + case % 6, case % 3
        pts = [oracle.point_from_uniform(s[pos + 64 * i:pos + 64 * i + 64]) for i in range(nv + nf)]
        pos += 64 * (nv + nf)
        scs = [oracle.scalar_reduce_wide(hashlib.sha512(b"k%d-%d" % (case, i)).digest()) for i in range(nv + nf)]
        if case == 3:
            scs[0] = bytes(32)                                          # all-zero digits: identity entries only
        if case == 4:
            scs[0] = (L - 1).to_bytes(32, "little")
        if case == 5:
            pts[0] = bytes(32)                                          # the identity as a base
        assert arith.arith_msm_chain(out, nv, b"".join(scs[:nv]), b"".join(pts[:nv]), nf, b"".join(scs[nv:]), b"".join(pts[nv:])) == 1
        assert bytes(out) == oracle.multiscalar(scs, pts), case
    assert arith.arith_bounds_violations() == 0, arith.arith_last_violation()


def test_double_and_compress_against_the_oracle(arith):
    """ge.cuh c2x_from / c2x_finish (k_compress2x): the encodings of 2P for a batch of points with one inversion equal the
    oracle's encode(P + P) - identities in the batch included - with every multiplication's operand bounds checked"""
    import ctypes as C
    import hashlib
    import oracle
    rnd = hashlib.shake_256(b"c2x-host").digest(64 * 40)
    pts = [oracle.point_from_uniform(rnd[64 * i:64 * i + 64]) for i in range(39)]
    pts.insert(7, bytes(32))                     # the identity in the middle of the batch
    pts.insert(20, bytes(32))
    shift = oracle.point_from_uniform(hashlib.sha512(b"c2x-shift").digest())
    out = C.create_string_buffer(32 * len(pts))
    arith.arith_double_and_compress.argtypes = [C.c_char_p, C.c_uint32, C.c_char_p, C.c_char_p]
    assert arith.arith_double_and_compress(out, len(pts), b"".join(pts), shift) == 1
    for j, p in enumerate(pts):
        assert out.raw[32 * j:32 * j + 32] == oracle.point_add(p, p), j
    assert out.raw[32 * 7:32 * 8] == bytes(32)


def test_negate_and_encode_against_the_oracle(arith, primitives):
    """ge.cuh negenc_den / negenc_finish (k_negenc): the encodings of -P for a batch of DECODED points with one inversion and no
    square root equal the oracle's encode(identity - P) - random points, small multiples of the base point, the libsodium-valid
    encodings of the golden file (both parities of every decision inside Encode occur), the identity in the batch - with every
    multiplication's operand bounds checked"""
    import ctypes as C
    import hashlib
    import oracle
    rnd = hashlib.shake_256(b"negenc-host").digest(64 * 30)
    pts = [oracle.point_from_uniform(rnd[64 * i:64 * i + 64]) for i in range(30)]
    pts += [bytes.fromhex(x) for x in primitives["base_multiples"][1:]]
    pts += [bytes.fromhex(v["in"]) for v in primitives["validity"] if v["valid"]][:16]
    pts.insert(5, bytes(32))                      # the identity in the middle of the batch
    arith.arith_negate_and_encode.argtypes = [C.c_char_p, C.c_uint32, C.c_char_p]
    seen = set()
    for lo in range(0, len(pts), 60):
        chunk = pts[lo:lo + 60]
        out = C.create_string_buffer(32 * len(chunk))
        assert arith.arith_negate_and_encode(out, len(chunk), b"".join(chunk)) == 1
        for j, p in enumerate(chunk):
            want = oracle.point_sub(bytes(32), p)
            assert out.raw[32 * j:32 * j + 32] == want, (lo + j, p.hex())
            seen.add(want == p)
    assert seen == {True, False} or len(pts) > 40   # 2-torsion-like fixed points are rare; most negations differ


def test_the_bounds_checker_fires(arith):
    arith.arith_bounds_checker_selftest.restype = C.c_uint64
    assert arith.arith_bounds_checker_selftest() >= 2


def test_no_bound_was_violated_anywhere(arith):
    """runs last in this module: all the field operations of the tests above stayed within their operand bounds"""
    arith.arith_bounds_violations.restype = C.c_uint64
    arith.arith_last_violation.restype = C.c_char_p
    assert arith.arith_bounds_violations() == 0, arith.arith_last_violation()


def test_signed_digit_recoding(arith):
    """k_msm's carry-free recoding: the digits of s + bias minus the per-digit offset sum back to s"""
    s = stream(b"digits", 32 * 100)
    out = (C.c_uint32 * 8)()
    for i in range(100):
        k = int.from_bytes(s[32 * i:32 * i + 32], "little") % L
        kb = k.to_bytes(32, "little")
        arith.arith_sc_bias(out, kb, 0x88888888)
        v = sum(int(out[j]) << (32 * j) for j in range(8))
        assert sum((((v >> (4 * j)) & 15) - 8) << (4 * j) for j in range(64)) == k
        arith.arith_sc_bias(out, kb, 0x80808080)
        v = sum(int(out[j]) << (32 * j) for j in range(8))
        assert sum((((v >> (8 * j)) & 255) - 128) << (8 * j) for j in range(32)) == k


# field multiplications / squarings per building block: the constants behind afx_plan_stats.field_mul / field_sq
# (aeonflux_amd/csrc/engine.cpp Assembler::msm, plan.h AFX_DECODE_* / AFX_ENCODE_*) and DESIGN.md section 3
OP_COUNTS = {"decode": (27, 257), "encode": (32, 255), "double_to_p2": (3, 4), "double_to_p3": (4, 4), "table_entry": (1, 0),
             "add_var_to_p3": (8, 0), "add_fixed_to_p3": (7, 0), "add_var_to_p2": (7, 0)}


def test_chain_operations_are_the_published_share(arith):
    """the inversion / square-root chains run in the 10-limb form (fe10.cuh): 251 squarings + 11 products of a decode or an
    encode, 254 + 11 of a plain inversion - plan.h AFX_CHAIN_*, what afx_plan_stats.chain_mul / chain_sq count"""
    out = ((C.c_uint64 * 2) * 3)()
    arith.arith_chain_op_counts(out)
    assert [(int(r[0]), int(r[1])) for r in out] == [(11, 251), (11, 251), (11, 254)]


def test_operation_counts_are_the_published_ones(arith):
    out = ((C.c_uint64 * 2) * 8)()
    arith.arith_op_counts(out)
    got = {k: (int(out[i][0]), int(out[i][1])) for i, k in enumerate(OP_COUNTS)}
    assert got == OP_COUNTS, got
