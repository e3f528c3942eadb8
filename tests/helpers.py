"""Shared test helpers: build seeded batches of presentations with the ORACLE, corrupt them, run the GPU
engine through the C ABI.  (The oracle is the checker and input generator here, never the thing tested
against itself.)"""
import hashlib

import numpy as np

H = bytes.fromhex


def pres_from_json(r):
    import oracle
    p = oracle.Presentation()
    j = r["presentation"]
    p.n_attributes, p.n_responses = j["n_attributes"], j["n_responses"]

    def put(dst, hexs):
        b = H(hexs)
        for i in range(32):
            dst[i] = b[i]
    put(p.challenge, j["challenge"])
    for k, x in enumerate(j["responses"]):
        put(p.responses[k], x)
    put(p.C_x_0, j["C_x_0"])
    put(p.C_x_1, j["C_x_1"])
    put(p.C_V, j["C_V"])
    for k in range(p.n_attributes):
        put(p.C_y[k], j["C_y"][k])
        put(p.attr_values[k], j["attr_values"][k])
        p.kinds[k] = j["kinds"][k]
    p.n_hidden_scalars = len(j["hidden_scalar_indices"])
    for k, x in enumerate(j["hidden_scalar_indices"]):
        p.hidden_scalar_indices[k] = x
    p.n_enc_proofs = len(j["enc"])
    for e, q in enumerate(j["enc"]):
        put(p.enc[e].challenge, q["challenge"])
        for k in range(6):
            put(p.enc[e].responses[k], q["responses"][k])
        for f in ("pk", "E1", "E2", "C_y_1", "C_y_2", "C_y_3", "C_y_2p"):
            put(getattr(p.enc[e], f), q[f])
        p.enc[e].index = q["index"]
    return p


def gpu_verify(afx, ctx, pres, shape=None):
    from tests.soa import pack_presentations
    sh, soa, keep = pack_presentations(pres)
    if shape is not None:
        sh = shape
    gsh = afx.Shape.from_buffer_copy(bytes(sh))
    gsoa = afx.PresentationSoA.from_buffer_copy(bytes(soa))
    status = np.full(len(pres), 7, np.uint8)
    ctx.verify_presentations(gsh, gsoa, len(pres), status.ctypes.data)
    return status.tolist()


def make_credentials(n, layout, count, seed):
    """count issued credentials of one layout via the oracle.
    Returns dict(params,key,ip,issuer,user,take, creds=[(kinds, values, t,U,V, challenge, responses)])"""
    import oracle
    s = hashlib.shake_256(seed).digest(1 << 22)
    pos = [0]

    def take(k):
        b = s[pos[0]:pos[0] + k]
        assert len(b) == k
        pos[0] += k
        return b
    params, used = oracle.system_parameters_generate(n, s)
    pos[0] = used
    key, ip = oracle.issuer_new(params, take(64 * (4 + n)))
    issuer, user = oracle.Ctx(params, key, ip), oracle.Ctx(params, None, ip)
    pad = lambda b: b + bytes(96 - len(b))
    creds = []
    for _ in range(count):
        kinds, vals = [], []
        for c in layout:
            if c == "S":
                kinds.append(0)
                vals.append(pad(oracle.scalar_reduce_wide(take(64))))
            elif c == "P":
                kinds.append(2)
                vals.append(pad(oracle.point_from_uniform(take(64))))
            else:
                kinds.append(3)
                vals.append(oracle.plaintext_from_bytes(take(30))[0])
        rnd = (take(64), take(64), take(32))
        st, t, U, V, ch, resp = issuer.issue(kinds, vals, *rnd)
        assert st == 0
        creds.append(dict(kinds=kinds, values=vals, t=t, U=U, V=V, challenge=ch, responses=resp, rnd=rnd))
    return dict(params=params, key=key, ip=ip, issuer=issuer, user=user, take=take, creds=creds)


def make_batch(n, layout, hide, count, seed):
    """count honest presentations of one layout via the oracle; returns (params,key,ip,issuer,[Presentation])"""
    d = make_credentials(n, layout, count, seed)
    take, user = d["take"], d["user"]
    out = []
    for cr in d["creds"]:
        kinds = list(cr["kinds"])
        for i in hide:
            kinds[i] = 1 if kinds[i] == 0 else 4
        kp = user.keypair_derive(take(64))
        nsp = sum(1 for k in kinds if k == 4)
        st, p = user.show(kinds, cr["values"], cr["t"], cr["U"], cr["V"], kp, take(64), take(32), take(32 * nsp))
        assert st == 0
        out.append(p)
    return d["params"], d["key"], d["ip"], d["issuer"], out


def corrupt(pres, seed):
    """flip / replace one field in roughly 40 % of the items; every class of corruption appears"""
    import oracle
    rnd = hashlib.shake_256(seed).digest(64 * len(pres) + 64)
    other_pt = oracle.point_from_uniform(rnd[-64:])
    for i, p in enumerate(pres):
        mode = rnd[64 * i] % 24
        tgt, val = None, None
        if mode == 0:
            tgt = p.responses[0]
        elif mode == 1:
            tgt = p.C_V
        elif mode == 2 and p.n_enc_proofs:
            tgt = p.enc[0].E1
        elif mode == 3:
            tgt, val = p.C_y[0], bytes(32)                 # identity commitment
        elif mode == 4:
            tgt, val = p.C_x_0, b"\xff" * 32             # non-canonical encoding
        elif mode == 5:
            tgt = p.challenge
        elif mode == 6:
            tgt, val = p.C_x_1, other_pt
        elif mode == 7 and p.n_enc_proofs:
            tgt = p.enc[p.n_enc_proofs - 1].responses[5]
        elif mode == 8:
            tgt, val = p.responses[p.n_responses - 1], b"\xff" * 32   # non-canonical scalar
        elif mode == 9 and p.n_enc_proofs:
            tgt, val = p.enc[0].C_y_2p, other_pt
        if tgt is None:
            continue
        if val is None:
            tgt[rnd[64 * i + 1] % 31] ^= 1 << (rnd[64 * i + 2] % 8)
        else:
            for k in range(32):
                tgt[k] = val[k]


class DevMem:
    """Device buffers for the *_dev entry points without PyTorch: ctypes over the HIP runtime the engine itself is linked
    against (a second HIP runtime in the process - the one inside the torch wheel - does not initialise once this one has)."""
    _hip = None

    @classmethod
    def hip(cls):
        if cls._hip is None:
            import ctypes as C
            h = C.CDLL("/opt/rocm/lib/libamdhip64.so.7")
            h.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
            h.hipFree.argtypes = [C.c_void_p]
            h.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
            h.hipMemset.argtypes = [C.c_void_p, C.c_int, C.c_size_t]
            cls._hip = h
        return cls._hip

    def __init__(self, array=None, nbytes=None, fill=None):
        import ctypes as C
        a = None if array is None else np.ascontiguousarray(array, dtype=np.uint8)
        self.nbytes = a.nbytes if a is not None else nbytes
        self.shape = a.shape if a is not None else (nbytes,)
        p = C.c_void_p()
        assert self.hip().hipMalloc(C.byref(p), max(1, self.nbytes)) == 0
        self.ptr = p.value
        if a is not None:
            assert self.hip().hipMemcpy(self.ptr, a.ctypes.data, a.nbytes, 1) == 0      # hipMemcpyHostToDevice
        elif fill is not None:
            assert self.hip().hipMemset(self.ptr, fill, self.nbytes) == 0

    def numpy(self):
        out = np.zeros(self.shape, np.uint8)
        assert self.hip().hipMemcpy(out.ctypes.data, self.ptr, self.nbytes, 2) == 0      # hipMemcpyDeviceToHost (synchronises)
        return out

    def free(self):
        if self.ptr:
            self.hip().hipFree(self.ptr)
            self.ptr = None

    __del__ = free
