"""ristretto255 over Python integers (RFC 9496 §4) and scalars mod l.  Points are extended coordinates (X, Y, Z, T)."""
P = 2**255 - 19
L = 2**252 + 27742317777372353535851937790883648493
D = (-121665 * pow(121666, P - 2, P)) % P
SQRT_M1 = pow(2, (P - 1) // 4, P)


def _neg(x):
    return x % P & 1 == 1


def _abs(x):
    x %= P
    return P - x if x & 1 else x


def sqrt_ratio_m1(u, v):
    u %= P
    v %= P
    v3 = v * v % P * v % P
    v7 = v3 * v3 % P * v % P
    r = u * v3 % P * pow(u * v7 % P, (P - 5) // 8, P) % P
    check = v * r % P * r % P
    correct = check == u
    flipped = check == (-u) % P
    flipped_i = check == (-u) * SQRT_M1 % P
    if flipped or flipped_i:
        r = r * SQRT_M1 % P
    return (correct or flipped), _abs(r)


def _const_sqrt(v, want_square=True):
    ok, r = sqrt_ratio_m1(v, 1)
    assert ok
    return r


# a = -1.  RFC 9496 §4.1 constants, derived instead of typed in; the signs RFC 9496 lists are checked by the KATs in tests/
ONE_MINUS_D_SQ = (1 - D * D) % P
D_MINUS_ONE_SQ = (D - 1) * (D - 1) % P
_ok, _r = sqrt_ratio_m1(1, (-1 - D) % P)
assert _ok
INVSQRT_A_MINUS_D = _r                      # 1/sqrt(a - d), the non-negative root: ...7578 in RFC 9496
SQRT_AD_MINUS_ONE = (P - _const_sqrt((-D - 1) % P)) % P   # sqrt(a*d - 1), RFC 9496 lists the NEGATIVE root (...0235)
assert INVSQRT_A_MINUS_D == 54469307008909316920995813868745141605393597292927456921205312896311721017578
assert SQRT_AD_MINUS_ONE == 25063068953384623474111414158702152701244531502492656460079210482610430750235
assert SQRT_M1 == 19681161376707505956807079304988542015446066515923890162744021073123829784752

IDENTITY = (0, 1, 1, 0)


def decode(b):
    """32 bytes -> point or None"""
    s = int.from_bytes(b, "little")
    if s >= P or s & 1:
        return None
    ss = s * s % P
    u1 = (1 - ss) % P
    u2 = (1 + ss) % P
    u2_sqr = u2 * u2 % P
    v = (-(D * u1 % P * u1) - u2_sqr) % P
    was_square, invsqrt = sqrt_ratio_m1(1, v * u2_sqr % P)
    den_x = invsqrt * u2 % P
    den_y = invsqrt * den_x % P * v % P
    x = _abs(2 * s * den_x % P)
    y = u1 * den_y % P
    t = x * y % P
    if not was_square or _neg(t) or y == 0:
        return None
    return (x, y, 1, t)


def encode(pt):
    X, Y, Z, T = pt
    u1 = (Z + Y) * (Z - Y) % P
    u2 = X * Y % P
    _, invsqrt = sqrt_ratio_m1(1, u1 * u2 % P * u2 % P)
    den1 = invsqrt * u1 % P
    den2 = invsqrt * u2 % P
    z_inv = den1 * den2 % P * T % P
    ix = X * SQRT_M1 % P
    iy = Y * SQRT_M1 % P
    enchanted = den1 * INVSQRT_A_MINUS_D % P
    if _neg(T * z_inv % P):
        x, y, den_inv = iy, ix, enchanted
    else:
        x, y, den_inv = X, Y, den2
    if _neg(x * z_inv % P):
        y = (-y) % P
    return _abs(den_inv * (Z - y) % P).to_bytes(32, "little")


# TRACE: when a list, every top-level group operation of the statement code is recorded as (op, [scalars], [encodings of the
# input points], encoding of the result) - tests/gen_golden.py and tests/test_pyref_cross_check.py replay the records through
# libsodium, which pins the GROUP VALUES of every flow (tags, commitments, ciphertexts, recomputed commitments) to a third
# implementation; the operations inside mul / msm / sub are not recorded on their own.
TRACE = None
_depth = 0


def _traced(op):
    def wrap(fn):
        def inner(*args):
            global _depth
            if TRACE is None or _depth:
                return fn(*args)
            _depth += 1
            try:
                out = fn(*args)
            finally:
                _depth -= 1
            if op == "mul":
                TRACE.append((op, [args[0] % L], [encode(args[1])], encode(out)))
            elif op == "msm":
                TRACE.append((op, [k % L for k in args[0]], [encode(q) for q in args[1]], encode(out)))
            else:
                TRACE.append((op, [], [encode(q) for q in args], encode(out)))
            return out
        return inner
    return wrap


def _add(p, q):
    X1, Y1, Z1, T1 = p
    X2, Y2, Z2, T2 = q
    A = (Y1 - X1) * (Y2 - X2) % P
    B = (Y1 + X1) * (Y2 + X2) % P
    C = 2 * D * T1 % P * T2 % P
    Dd = 2 * Z1 * Z2 % P
    E, F, G, H = B - A, Dd - C, Dd + C, B + A
    return (E * F % P, G * H % P, F * G % P, E * H % P)


add = _traced("add")(_add)


@_traced("neg")
def neg(p):
    return ((-p[0]) % P, p[1], p[2], (-p[3]) % P)


@_traced("sub")
def sub(p, q):
    return _add(p, ((-q[0]) % P, q[1], q[2], (-q[3]) % P))


def _mul(k, p):
    k %= L
    acc = IDENTITY
    for bit in bin(k)[2:] if k else "":
        acc = _add(acc, acc)
        if bit == "1":
            acc = _add(acc, p)
    return acc


mul = _traced("mul")(_mul)


@_traced("msm")
def msm(scalars, points):
    acc = IDENTITY
    for k, p in zip(scalars, points):
        acc = _add(acc, _mul(k, p))
    return acc


def _elligator(r0):
    r = SQRT_M1 * r0 % P * r0 % P
    u = (r + 1) * ONE_MINUS_D_SQ % P
    c = P - 1
    v = (c - r * D) % P * ((r + D) % P) % P
    was_square, s = sqrt_ratio_m1(u, v)
    s_prime = (-_abs(s * r0 % P)) % P
    if not was_square:
        s, c = s_prime, r
    N = (c * ((r - 1) % P) % P * D_MINUS_ONE_SQ - v) % P
    w0 = 2 * s * v % P
    w1 = N * SQRT_AD_MINUS_ONE % P
    w2 = (1 - s * s) % P
    w3 = (1 + s * s) % P
    return (w0 * w3 % P, w2 * w1 % P, w1 * w3 % P, w0 * w2 % P)


def from_uniform_bytes(b64):
    """RistrettoPoint::from_uniform_bytes: two field elements (bit 255 of each half ignored), MAP each, add"""
    r0 = int.from_bytes(b64[:32], "little") & ((1 << 255) - 1)
    r1 = int.from_bytes(b64[32:], "little") & ((1 << 255) - 1)
    return add(_elligator(r0 % P), _elligator(r1 % P))


def sc_from_wide(b64):
    return int.from_bytes(b64, "little") % L


def sc_canonical(b32):
    """Scalar::from_canonical_bytes: int or None"""
    v = int.from_bytes(b32, "little")
    return v if v < L else None


def sc_bytes(k):
    return (k % L).to_bytes(32, "little")


def double_and_compress(pts):
    """encodings of 2*P for a list of points with ONE field inversion (curve25519-dalek's double_and_compress_batch [3P]; what the
    engine's k_compress2x does per item over the commitments of its proofs).  e*f*g*h = 0 only for representatives of the identity."""
    st = []
    for X, Y, Z, T in pts:
        XX, YY, ZZ, dTT = X * X % P, Y * Y % P, Z * Z % P, T * T % P * D % P
        e, f, g, h = X * (2 * Y) % P, (ZZ + dTT) % P, (YY + XX) % P, (ZZ - dTT) % P
        st.append((e, f, g, h, e * g % P, f * h % P))
    prod, pre = 1, []
    for e, f, g, h, eg, fh in st:
        pre.append(prod)
        w = eg * fh % P
        prod = prod * (w if w else 1) % P
    inv = pow(prod, P - 2, P)
    out = [None] * len(st)
    for j in range(len(st) - 1, -1, -1):
        e, f, g, h, eg, fh = st[j]
        w = eg * fh % P
        if w == 0:
            out[j] = bytes(32)
            continue
        inv_j = inv * pre[j] % P
        inv = inv * w % P
        zinv, tinv = eg * inv_j % P, fh * inv_j % P
        magic = INVSQRT_A_MINUS_D
        if _neg(eg * zinv % P):
            e, g, h, magic = g, (-e) % P, f * SQRT_M1 % P, SQRT_M1
        if _neg(h * e % P * zinv % P):
            g = (-g) % P
        out[j] = _abs((h - g) * (magic * (g * tinv % P) % P) % P).to_bytes(32, "little")
    return out
