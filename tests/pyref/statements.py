"""The aeonflux statements, restated from /root/reference/src (second restatement; see __init__.py):
  Amac::tag + ProofOfIssuance::prove          src/amacs.rs:225-294, src/nizk/issuance.rs:40-129   -> issue()
  ProofOfIssuance::verify                     src/nizk/issuance.rs:132-218                        -> issuance_verify()
  ProofOfValidCredential::prove               src/nizk/presentation.rs:139-321                    -> show()
  ProofOfEncryption::prove / verify           src/nizk/encryption.rs:58-142, 154-210
  ProofOfValidCredential::verify              src/nizk/presentation.rs:324-443                    -> verify_presentation()
Byte layouts: SystemParameters::to_bytes (src/parameters.rs:155-184), amacs::SecretKey::to_bytes (src/amacs.rs:110-125),
IssuerParameters as C_W || I.  Status codes are those of include/aeonflux_gpu.h.  Every random draw is an argument."""
from . import ristretto as R
from .keccak import Transcript
from .zkp import ProofError, Prover, Verifier

PUBLIC_SCALAR, SECRET_SCALAR, PUBLIC_POINT, EITHER_POINT, SECRET_POINT = range(5)      # amacs::Attribute
E_PUBLIC_SCALAR, E_SECRET_SCALAR, E_PUBLIC_POINT, E_SECRET_POINT = range(4)            # amacs::EncryptedAttribute
OK, VERIFICATION_FAILURE, MAC_CREATION, NO_SYMMETRIC_KEY = 0, 1, 2, 3


class Params:
    def __init__(self, b):
        self.n = int.from_bytes(b[:4], "little")
        g = max(3, self.n)
        assert len(b) == 4 + 32 * (5 + g + self.n + 4)
        pts = [b[4 + 32 * i:36 + 32 * i] for i in range(5 + g + self.n + 4)]
        dec = [R.decode(x) for x in pts]
        assert all(p is not None for p in dec)
        self.G, self.G_w, self.G_w_prime, self.G_x_0, self.G_x_1 = dec[:5]
        self.G_y = dec[5:5 + g]
        self.G_m = dec[5 + g:5 + g + self.n]
        self.G_V, self.G_a, self.G_a0, self.G_a1 = dec[5 + g + self.n:]


class Key:
    def __init__(self, b):
        n = int.from_bytes(b[:4], "little")
        assert len(b) == 4 + 32 * (5 + n)
        sc = [int.from_bytes(b[4 + 32 * i:36 + 32 * i], "little") for i in range(4 + n)]
        assert all(s < R.L for s in sc)
        self.w, self.w_prime, self.x_0, self.x_1 = sc[:4]
        self.y = sc[4:]
        self.W = R.decode(b[4 + 32 * (4 + n):])


def _issuer_params(ip):
    return R.decode(ip[:32]), R.decode(ip[32:])       # C_W, I


def _messages(sp, kinds, values):
    """Messages::from_attributes (amacs.rs:225-243); values[i] = 32-byte scalar, or point M1; None on a bad encoding"""
    out = []
    for i, (k, v) in enumerate(zip(kinds, values)):
        if k in (PUBLIC_SCALAR, SECRET_SCALAR):
            m = R.sc_canonical(v[:32])
            if m is None:
                return None
            out.append(R.mul(m, sp.G_m[i]))
        else:
            p = R.decode(v[:32])
            if p is None:
                return None
            out.append(p)
    return out


def _issuance_statement(cs, sp, C_W, I, U, V, tU, M, point):
    """shared by prover and verifier: allocation order and constraints of issuance.rs.  `point(label, P)` allocates."""
    G_V = point(b"G_V", sp.G_V)
    G_w = point(b"G_w", sp.G_w)
    G_w_prime = point(b"G_w_prime", sp.G_w_prime)
    neg_G_x_0 = point(b"-G_x_0", R.neg(sp.G_x_0))
    neg_G_x_1 = point(b"-G_x_1", R.neg(sp.G_x_1))
    neg_G_y = [point(b"-G_y", R.neg(g)) for g in sp.G_y]
    vC_W = point(b"C_W", C_W)
    vI = point(b"I", I)
    vU = point(b"U", U)
    vV = point(b"V", V)
    vtU = point(b"tU", tU)
    vM = [point(b"M", m) for m in M]
    return G_V, G_w, G_w_prime, neg_G_x_0, neg_G_x_1, neg_G_y, vC_W, vI, vU, vV, vtU, vM


def issue(params, key, ip, kinds, values, t_wide, U_wide, seed):
    sp, sk = Params(params), Key(key)
    C_W, I = _issuer_params(ip)
    if len(kinds) != sp.n:
        return MAC_CREATION, None
    M = _messages(sp, kinds, values)
    if M is None:
        return MAC_CREATION, None
    t = R.sc_from_wide(t_wide)
    U = R.from_uniform_bytes(U_wide)
    V = R.add(R.add(sk.W, R.mul(sk.x_0, U)), R.mul(sk.x_1 * t, U))
    V = R.add(V, R.msm(sk.y, M))
    tr = Transcript(b"2019/1416 anonymous credential")
    pr = Prover(b"2019/1416 issuance proof", tr)
    w = pr.allocate_scalar(b"w", sk.w)
    w_prime = pr.allocate_scalar(b"w'", sk.w_prime)
    x_0 = pr.allocate_scalar(b"x_0", sk.x_0)
    x_1 = pr.allocate_scalar(b"x_1", sk.x_1)
    y = [pr.allocate_scalar(b"y", yi) for yi in sk.y]
    one = pr.allocate_scalar(b"1", 1)
    G_V, G_w, G_w_prime, nGx0, nGx1, nGy, vC_W, vI, vU, vV, vtU, vM = _issuance_statement(pr, sp, C_W, I, U, V, R.mul(t, U), M, pr.allocate_point)
    pr.constrain(vC_W, [(w, G_w), (w_prime, G_w_prime)])
    pr.constrain(vI, [(one, G_V), (x_0, nGx0), (x_1, nGx1)] + list(zip(y, nGy)))       # zip truncates to n terms
    pr.constrain(vV, [(w, G_w), (x_0, vU), (x_1, vtU)] + list(zip(y, vM)))
    ch, resp, coms = pr.prove_compact(seed)
    return OK, dict(t=R.sc_bytes(t), U=R.encode(U), V=R.encode(V), challenge=R.sc_bytes(ch), responses=[R.sc_bytes(r) for r in resp], commitments=coms)


def issuance_verify(params, ip, kinds, values, t, U, V, challenge, responses):
    sp = Params(params)
    C_W, I = _issuer_params(ip)
    try:
        tt, ch = R.sc_canonical(t), R.sc_canonical(challenge)
        rs = [R.sc_canonical(r) for r in responses]
        pU, pV = R.decode(U), R.decode(V)
        M = _messages(sp, kinds, values)
        if tt is None or ch is None or None in rs or pU is None or pV is None or M is None or len(kinds) > sp.n:
            raise ProofError("malformed")
        tr = Transcript(b"2019/1416 anonymous credential")
        ve = Verifier(b"2019/1416 issuance proof", tr)
        w = ve.allocate_scalar(b"w")
        w_prime = ve.allocate_scalar(b"w'")
        x_0 = ve.allocate_scalar(b"x_0")
        x_1 = ve.allocate_scalar(b"x_1")
        y = [ve.allocate_scalar(b"y") for _ in range(sp.n)]
        one = ve.allocate_scalar(b"1")
        G_V, G_w, G_w_prime, nGx0, nGx1, nGy, vC_W, vI, vU, vV, vtU, vM = _issuance_statement(
            ve, sp, C_W, I, pU, pV, R.mul(tt, pU), M, lambda label, P: ve.allocate_point(label, R.encode(P)))
        ve.constrain(vC_W, [(w, G_w), (w_prime, G_w_prime)])
        ve.constrain(vI, [(one, G_V), (x_0, nGx0), (x_1, nGx1)] + list(zip(y, nGy)))
        ve.constrain(vV, [(w, G_w), (x_0, vU), (x_1, vtU)] + list(zip(y, vM)))
        coms = ve.verify_compact(ch, rs)
        return OK, coms
    except ProofError:
        return VERIFICATION_FAILURE, None


def _enc_statement(cs, sp, index, pk, C_y_2, C_y_3, C_y_2p, C_y_1_minus_E2, E1, minus_E1, point):
    v = {}
    for label, P in ((b"pk", pk), (b"G_a", sp.G_a), (b"G_a_0", sp.G_a0), (b"G_a_1", sp.G_a1), (b"G_y_1", sp.G_y[0]), (b"G_y_2", sp.G_y[1]),
                     (b"G_y_3", sp.G_y[2]), (b"G_m_3", sp.G_m[index]), (b"C_y_2", C_y_2), (b"C_y_3", C_y_3), (b"C_y_2'", C_y_2p),
                     (b"C_y_1-E2", C_y_1_minus_E2), (b"E1", E1), (b"-E1", minus_E1)):
        v[label] = point(label, P)
    return v


def _enc_constraints(cs, v, a, a0, a1, m3, z, z1):
    cs.constrain(v[b"pk"], [(a, v[b"G_a"]), (a0, v[b"G_a_0"]), (a1, v[b"G_a_1"])])
    cs.constrain(v[b"C_y_1-E2"], [(z, v[b"G_y_1"]), (a, v[b"-E1"])])
    cs.constrain(v[b"C_y_2'"], [(a1, v[b"C_y_2"])])
    cs.constrain(v[b"E1"], [(a0, v[b"C_y_2"]), (m3, v[b"C_y_2'"]), (z1, v[b"G_y_2"])])
    cs.constrain(v[b"C_y_3"], [(z, v[b"G_y_3"]), (m3, v[b"G_m_3"])])


def prove_encryption(sp, plaintext, index, keypair, z, seed):
    """plaintext = (M1, M2, m3) as (point, point, int); keypair = (a, a0, a1, pk-point)"""
    M1, M2, m3 = plaintext
    a, a0, a1, pk = keypair
    E1 = R.mul(a0 + a1 * m3, M2)                                     # Keypair::encrypt, symmetric.rs:252-261
    E2 = R.add(R.mul(a, E1), M1)
    C_y_1 = R.add(R.mul(z, sp.G_y[0]), M1)
    C_y_2 = R.add(R.mul(z, sp.G_y[1]), M2)
    C_y_3 = R.add(R.mul(z, sp.G_y[2]), R.mul(m3, sp.G_m[index]))
    C_y_2p = R.mul(a1, C_y_2)
    z1 = (-z * (a0 + a1 * m3)) % R.L
    tr = Transcript(b"2019/1416 anonymous credentials")
    pr = Prover(b"2019/1416 proof of encryption", tr)
    sa, sa0, sa1, sm3, sz, sz1 = (pr.allocate_scalar(lbl, val) for lbl, val in ((b"a", a), (b"a0", a0), (b"a1", a1), (b"m3", m3), (b"z", z), (b"z1", z1)))
    v = _enc_statement(pr, sp, index, pk, C_y_2, C_y_3, C_y_2p, R.sub(C_y_1, E2), E1, R.neg(E1), pr.allocate_point)
    _enc_constraints(pr, v, sa, sa0, sa1, sm3, sz, sz1)
    ch, resp, coms = pr.prove_compact(seed)
    return dict(index=index, challenge=R.sc_bytes(ch), responses=[R.sc_bytes(r) for r in resp], pk=R.encode(pk), E1=R.encode(E1), E2=R.encode(E2),
                C_y_1=R.encode(C_y_1), C_y_2=R.encode(C_y_2), C_y_3=R.encode(C_y_3), C_y_2p=R.encode(C_y_2p), commitments=coms)


def verify_encryption(sp, e):
    """raises ProofError or returns the recomputed commitments"""
    if e["index"] >= sp.n:
        raise ProofError("G_m[index] out of range (the reference panics)")
    ch = R.sc_canonical(e["challenge"])
    rs = [R.sc_canonical(r) for r in e["responses"]]
    pts = {f: R.decode(e[f]) for f in ("pk", "E1", "E2", "C_y_1", "C_y_2", "C_y_3", "C_y_2p")}
    if ch is None or None in rs or None in pts.values():
        raise ProofError("malformed")
    tr = Transcript(b"2019/1416 anonymous credentials")
    ve = Verifier(b"2019/1416 proof of encryption", tr)
    sa, sa0, sa1, sm3, sz, sz1 = (ve.allocate_scalar(lbl) for lbl in (b"a", b"a0", b"a1", b"m3", b"z", b"z1"))
    v = _enc_statement(ve, sp, e["index"], pts["pk"], pts["C_y_2"], pts["C_y_3"], pts["C_y_2p"], R.sub(pts["C_y_1"], pts["E2"]), pts["E1"],
                       R.neg(pts["E1"]), lambda label, P: ve.allocate_point(label, R.encode(P)))
    _enc_constraints(ve, v, sa, sa0, sa1, sm3, sz, sz1)
    return ve.verify_compact(ch, rs)


def _lookup(pairs, i):
    """the reference's Index impls over (attribute index, var) lists: panic when absent (presentation.rs:74-103)"""
    for idx, var in pairs:
        if idx == i:
            return var
    raise ProofError("hidden-scalar lookup by a position that holds none (the reference panics)")


def _own_position_constraints(cs, kinds_at, secret_point, secret_scalar, C_y, keep, G_y, H_s, G_m, z):
    """strict mode: the j-th kept commitment against the generators and the kind of ITS OWN attribute position"""
    for j, i in enumerate(keep):
        if kinds_at[i] == secret_scalar:
            cs.constrain(C_y[j], [(z, G_y[i]), (_lookup(H_s, i), _lookup(G_m, i))])
        else:
            cs.constrain(C_y[j], [(z, G_y[i])])


def show(params, ip, kinds, values, t, U, V, keypair, z_wide, seed, enc_seeds, strict=False):
    """values[i]: 96 bytes = scalar or M1 | M2 | m3.  keypair: 128 bytes a|a0|a1|pk or None.  Returns (status, presentation dict).
    strict: the engine's opt-in strict mode (own-position constraint #3 and the DLEQ with each proof of encryption's C_y_1)."""
    sp = Params(params)
    C_W, I = _issuer_params(ip)
    n = sp.n
    if keypair is None and SECRET_POINT in kinds:
        return NO_SYMMETRIC_KEY, None
    tt, pU, pV = R.sc_canonical(t), R.decode(U), R.decode(V)
    z_ = R.sc_from_wide(z_wide)
    z_0_ = (-tt * z_) % R.L
    C_y_, H_s_ = [], []
    plain = {}
    for i, k in enumerate(kinds):
        zg = R.mul(z_, sp.G_y[i])
        if k == SECRET_POINT:
            M1, M2, m3 = R.decode(values[i][:32]), R.decode(values[i][32:64]), R.sc_canonical(values[i][64:96])
            plain[i] = (M1, M2, m3)
            C_y_.append(R.add(zg, M1))
        elif k == SECRET_SCALAR:
            m = R.sc_canonical(values[i][:32])
            C_y_.append(R.add(zg, R.mul(m, sp.G_m[i])))
            H_s_.append((i, sp.G_m[i], m))
        else:
            C_y_.append(zg)
    C_x_0_ = R.add(R.mul(z_, sp.G_x_0), pU)
    C_x_1_ = R.add(R.mul(z_, sp.G_x_1), R.mul(tt, pU))
    C_V_ = R.add(R.mul(z_, sp.G_V), pV)
    Z_ = R.mul(z_, I)
    tr = Transcript(b"2019/1416 anonymous credential")
    pr = Prover(b"2019/1416 presentation proof", tr)
    z = pr.allocate_scalar(b"z", z_)
    z_0 = pr.allocate_scalar(b"z_0", z_0_)
    tv = pr.allocate_scalar(b"t", tt)
    H_s = [(i, pr.allocate_scalar(b"m", m)) for i, _, m in H_s_]
    vI = pr.allocate_point(b"I", I)
    C_x_1 = pr.allocate_point(b"C_x_1", C_x_1_)
    C_x_0 = pr.allocate_point(b"C_x_0", C_x_0_)
    G_x_0 = pr.allocate_point(b"G_x_0", sp.G_x_0)
    G_x_1 = pr.allocate_point(b"G_x_1", sp.G_x_1)
    C_y = [pr.allocate_point(b"C_y", c) for i, c in enumerate(C_y_) if kinds[i] != SECRET_POINT]
    G_y = [pr.allocate_point(b"G_y", g) for g in sp.G_y]
    G_m = [(i, pr.allocate_point(b"G_m", g)) for i, g, _ in H_s_]
    dleq, neg_G_y_1 = [], None
    if strict:
        # C_y[i] - C_y_1 = z*G_y[i] + z*(-G_y[0]) for every hidden group element i != 0 (C_y_1 = z*G_y[0] + M1, encryption.rs:70)
        for i, k in enumerate(kinds):
            if k == SECRET_POINT and i != 0:
                if neg_G_y_1 is None:
                    neg_G_y_1 = pr.allocate_point(b"-G_y_1", R.neg(sp.G_y[0]))
                dleq.append((pr.allocate_point(b"C_y-C_y_1", R.sub(R.mul(z_, sp.G_y[i]), R.mul(z_, sp.G_y[0]))), i))
    Z = pr.allocate_point(b"Z", Z_)
    pr.constrain(Z, [(z, vI)])
    pr.constrain(C_x_1, [(tv, C_x_0), (z_0, G_x_0), (z, G_x_1)])
    try:
        if strict:
            _own_position_constraints(pr, kinds, SECRET_POINT, SECRET_SCALAR, C_y, [i for i in range(len(kinds)) if kinds[i] != SECRET_POINT], G_y, H_s, G_m, z)
            for D, i in dleq:
                pr.constrain(D, [(z, G_y[i]), (z, neg_G_y_1)])
        for i, C_y_i in ([] if strict else list(enumerate(C_y))):            # COMPACT index used as an attribute position (presentation.rs:267-273)
            if kinds[i] == SECRET_POINT:
                continue
            if kinds[i] == SECRET_SCALAR:
                pr.constrain(C_y_i, [(z, G_y[i]), (_lookup(H_s, i), _lookup(G_m, i))])
            else:
                pr.constrain(C_y_i, [(z, G_y[i])])
    except (ProofError, IndexError):
        return VERIFICATION_FAILURE, None           # the reference panics here; the engine reports a failure
    ch, resp, coms = pr.prove_compact(seed)
    enc, e = [], 0
    ekinds, avals = [], []
    for i, k in enumerate(kinds):
        if k == PUBLIC_SCALAR:
            ekinds.append(E_PUBLIC_SCALAR); avals.append(values[i][:32])
        elif k == SECRET_SCALAR:
            ekinds.append(E_SECRET_SCALAR); avals.append(bytes(32))
        elif k in (PUBLIC_POINT, EITHER_POINT):
            ekinds.append(E_PUBLIC_POINT); avals.append(values[i][:32])
        else:
            kp = (R.sc_canonical(keypair[:32]), R.sc_canonical(keypair[32:64]), R.sc_canonical(keypair[64:96]), R.decode(keypair[96:128]))
            enc.append(prove_encryption(sp, plain[i], i, kp, z_, enc_seeds[32 * e:32 * e + 32]))
            e += 1
            ekinds.append(E_SECRET_POINT); avals.append(bytes(32))
    return OK, dict(kinds=ekinds, attr_values=avals, hidden_scalar_indices=[i for i, _, _ in H_s_], challenge=R.sc_bytes(ch),
                    responses=[R.sc_bytes(r) for r in resp], C_x_0=R.encode(C_x_0_), C_x_1=R.encode(C_x_1_), C_V=R.encode(C_V_),
                    C_y=[R.encode(c) for c in C_y_], enc=enc, commitments=coms)


def verify_presentation(params, key, ip, p, strict=False):
    """p: dict with kinds (E_*), attr_values, hidden_scalar_indices, challenge, responses, C_x_0, C_x_1, C_V, C_y, enc[...]
    Returns (status, commitments of the last proof verified or None)"""
    sp, sk = Params(params), Key(key)
    C_W, I = _issuer_params(ip)
    try:
        kinds = p["kinds"]
        n = len(kinds)
        if strict:   # exactly one proof of encryption per hidden group element, in position order
            if [e["index"] for e in p["enc"]] != [i for i, k in enumerate(kinds) if k == E_SECRET_POINT]:
                raise ProofError("proofs of encryption do not match the hidden group elements")
        if n > sp.n:
            raise ProofError("more attributes than the key has")
        ch = R.sc_canonical(p["challenge"])
        rs = [R.sc_canonical(r) for r in p["responses"]]
        C_x_0, C_x_1, C_V = (R.decode(p[f]) for f in ("C_x_0", "C_x_1", "C_V"))
        C_y = [R.decode(c) for c in p["C_y"]]
        if ch is None or None in rs or None in (C_x_0, C_x_1, C_V) or None in C_y:
            raise ProofError("malformed")
        Z_ = R.sub(R.sub(R.sub(C_V, sk.W), R.mul(sk.x_0, C_x_0)), R.mul(sk.x_1, C_x_1))
        for i, k in enumerate(kinds):
            if k == E_PUBLIC_SCALAR:
                m = R.sc_canonical(p["attr_values"][i])
                if m is None:
                    raise ProofError("malformed")
                x = R.add(C_y[i], R.mul(m, sp.G_m[i]))
            elif k == E_PUBLIC_POINT:
                M = R.decode(p["attr_values"][i])
                if M is None:
                    raise ProofError("malformed")
                x = R.add(C_y[i], M)
            else:
                x = C_y[i]
            Z_ = R.sub(Z_, R.mul(sk.y[i], x))
        tr = Transcript(b"2019/1416 anonymous credential")
        ve = Verifier(b"2019/1416 presentation proof", tr)
        z = ve.allocate_scalar(b"z")
        z_0 = ve.allocate_scalar(b"z_0")
        t = ve.allocate_scalar(b"t")
        H_s = [(i, ve.allocate_scalar(b"m")) for i in p["hidden_scalar_indices"]]
        pt = lambda label, P: ve.allocate_point(label, R.encode(P))
        vI = pt(b"I", I)
        vC_x_1 = ve.allocate_point(b"C_x_1", p["C_x_1"])
        vC_x_0 = ve.allocate_point(b"C_x_0", p["C_x_0"])
        G_x_0 = pt(b"G_x_0", sp.G_x_0)
        G_x_1 = pt(b"G_x_1", sp.G_x_1)
        vC_y = [ve.allocate_point(b"C_y", p["C_y"][i]) for i in range(n) if kinds[i] != E_SECRET_POINT]
        G_y = [pt(b"G_y", g) for g in sp.G_y]
        G_m = []
        for i, _ in H_s:
            if i >= sp.n:
                raise ProofError("G_m index out of range (the reference panics)")
            G_m.append((i, pt(b"G_m", sp.G_m[i])))
        dleq, neg_G_y_1 = [], None
        if strict:
            for e, i in zip(p["enc"], [i for i, k in enumerate(kinds) if k == E_SECRET_POINT]):
                C_y_1 = R.decode(e["C_y_1"])
                if C_y_1 is None:
                    raise ProofError("malformed")
                D = R.sub(C_y[i], C_y_1)
                if i == 0:
                    if R.encode(D) != bytes(32):
                        raise ProofError("C_y[0] differs from the C_y_1 of its proof of encryption")
                    continue
                if neg_G_y_1 is None:
                    neg_G_y_1 = pt(b"-G_y_1", R.neg(sp.G_y[0]))
                dleq.append((pt(b"C_y-C_y_1", D), i))
        Z = pt(b"Z", Z_)
        ve.constrain(Z, [(z, vI)])
        ve.constrain(vC_x_1, [(t, vC_x_0), (z_0, G_x_0), (z, G_x_1)])
        if strict:
            _own_position_constraints(ve, kinds, E_SECRET_POINT, E_SECRET_SCALAR, vC_y, [i for i in range(n) if kinds[i] != E_SECRET_POINT], G_y, H_s, G_m, z)
            for D, i in dleq:
                ve.constrain(D, [(z, G_y[i]), (z, neg_G_y_1)])
        for i, C_y_i in ([] if strict else list(enumerate(vC_y))):           # COMPACT index used as an attribute position (presentation.rs:427-433)
            if kinds[i] == E_SECRET_POINT:
                continue
            if i >= len(G_y):
                raise ProofError("G_y index out of range (the reference panics)")
            if kinds[i] == E_SECRET_SCALAR:
                ve.constrain(C_y_i, [(z, G_y[i]), (_lookup(H_s, i), _lookup(G_m, i))])
            else:
                ve.constrain(C_y_i, [(z, G_y[i])])
        coms = ve.verify_compact(ch, rs)
        for e in p["enc"]:
            coms = verify_encryption(sp, e)
        return OK, coms
    except (ProofError, IndexError):
        return VERIFICATION_FAILURE, None
