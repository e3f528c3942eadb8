"""ORACLE — test infrastructure only.

ctypes loader for the CPU restatement in oracle/*.c.  Only tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg may import this package; the product (aeonflux_amd) never does.
PARITY UNPINNED by the reference (no golden vectors in /root/reference); pinned by third-party KATs
and a libsodium cross-check (tests/golden/).
"""
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
MAX_ATTRS = 32

ST_OK, ST_VERIFICATION_FAILURE, ST_MAC_CREATION, ST_NO_SYMMETRIC_KEY = 0, 1, 2, 3
ATTR_PUBLIC_SCALAR, ATTR_SECRET_SCALAR, ATTR_PUBLIC_POINT, ATTR_EITHER_POINT, ATTR_SECRET_POINT = range(5)
ENC_PUBLIC_SCALAR, ENC_SECRET_SCALAR, ENC_PUBLIC_POINT, ENC_SECRET_POINT = range(4)

B32 = C.c_uint8 * 32


class EncProof(C.Structure):
    _fields_ = [("challenge", B32), ("responses", B32 * 6), ("pk", B32), ("E1", B32), ("E2", B32), ("C_y_1", B32),
                ("C_y_2", B32), ("C_y_3", B32), ("C_y_2p", B32), ("index", C.c_uint16)]


class Presentation(C.Structure):
    _fields_ = [("n_attributes", C.c_uint32), ("n_responses", C.c_uint32), ("challenge", B32),
                ("responses", B32 * (3 + MAX_ATTRS)), ("C_x_0", B32), ("C_x_1", B32), ("C_V", B32),
                ("C_y", B32 * MAX_ATTRS), ("kinds", C.c_uint8 * MAX_ATTRS), ("attr_values", B32 * MAX_ATTRS),
                ("n_hidden_scalars", C.c_uint32), ("hidden_scalar_indices", C.c_uint16 * MAX_ATTRS),
                ("n_enc_proofs", C.c_uint32), ("enc", EncProof * MAX_ATTRS)]


class Shape(C.Structure):
    """afx_shape of include/aeonflux_gpu.h"""
    _fields_ = [("n_attributes", C.c_uint32), ("kinds", C.c_uint8 * MAX_ATTRS), ("n_responses", C.c_uint32),
                ("n_hidden_scalars", C.c_uint32), ("hidden_scalar_indices", C.c_uint16 * MAX_ATTRS),
                ("n_enc_proofs", C.c_uint32), ("enc_indices", C.c_uint16 * MAX_ATTRS)]


class EncProofSoA(C.Structure):
    _fields_ = [(k, C.c_void_p) for k in ("challenge", "responses", "pk", "E1", "E2", "C_y_1", "C_y_2", "C_y_3", "C_y_2p")]


class PresentationSoA(C.Structure):
    _fields_ = [(k, C.c_void_p) for k in ("challenge", "responses", "C_x_0", "C_x_1", "C_V", "C_y", "attr_values")] + \
               [("enc", C.POINTER(EncProofSoA))]


def build(native=False):
    target = "native" if native else "build/libafx_oracle.so"
    subprocess.run(["make", "-C", _HERE, target], check=True, stdout=subprocess.DEVNULL)
    return os.path.join(_HERE, "build", "libafx_oracle_native.so" if native else "libafx_oracle.so")


def load(native=False):
    path = os.path.join(_HERE, "build", "libafx_oracle_native.so" if native else "libafx_oracle.so")
    if not os.path.exists(path):
        path = build(native)
    lib = C.CDLL(path)
    lib.afxo_ctx_new.restype = C.c_void_p
    lib.afxo_ctx_new.argtypes = [C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t, C.c_char_p]
    lib.afxo_ctx_free.argtypes = [C.c_void_p]
    lib.afxo_ctx_set_strict.argtypes = [C.c_void_p, C.c_int]
    lib.afxo_ctx_n.argtypes = [C.c_void_p]
    lib.afxo_ctx_n.restype = C.c_uint32
    lib.afxo_sizeof_system_parameters.restype = C.c_size_t
    lib.afxo_sizeof_secret_key.restype = C.c_size_t
    lib.afxo_system_parameters_generate.restype = C.c_long
    lib.afxo_system_parameters_generate.argtypes = [C.c_uint32, C.c_char_p, C.c_size_t, C.c_char_p]
    lib.afxo_issuer_new.argtypes = [C.c_char_p, C.c_size_t, C.c_char_p, C.c_char_p, C.c_char_p]
    lib.afxo_keypair_derive.argtypes = [C.c_void_p, C.c_char_p, C.c_char_p]
    lib.afxo_issue.argtypes = [C.c_void_p, C.c_uint32, C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p,
                               C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p]
    lib.afxo_issuance_verify.argtypes = [C.c_void_p, C.c_uint32, C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p,
                                         C.c_char_p, C.c_char_p, C.c_char_p, C.c_uint32]
    lib.afxo_show.argtypes = [C.c_void_p, C.c_uint32, C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p,
                              C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p, C.POINTER(Presentation)]
    lib.afxo_verify_presentation.argtypes = [C.c_void_p, C.POINTER(Presentation)]
    lib.afxo_verify_encryption_proof.argtypes = [C.c_void_p, C.POINTER(EncProof)]
    lib.afxo_verify_presentations_soa.argtypes = [C.c_void_p, C.POINTER(Shape), C.POINTER(PresentationSoA), C.c_size_t,
                                                  C.c_void_p, C.c_int]
    lib.afxo_verify_presentations_soa_traced.argtypes = [C.c_void_p, C.POINTER(Shape), C.POINTER(PresentationSoA), C.c_size_t,
                                                         C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
    lib.afxo_issue_soa.argtypes = [C.c_void_p, C.c_uint32, C.c_char_p] + [C.c_void_p] * 4 + [C.c_size_t] + [C.c_void_p] * 6 + [C.c_int]
    lib.afxo_verify_issuances_soa_traced.argtypes = [C.c_void_p, C.c_uint32, C.c_char_p] + [C.c_void_p] * 6 + [C.c_uint32, C.c_size_t] + \
        [C.c_void_p] * 3 + [C.c_int]
    lib.afxo_sha512.argtypes = [C.c_char_p, C.c_char_p, C.c_size_t]
    lib.afxo_merlin_simple.argtypes = [C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t,
                                       C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t]
    return lib


_LIB = None


def lib():
    global _LIB
    if _LIB is None:
        _LIB = load()
    return _LIB


# ---- thin pythonic helpers (bytes in / bytes out) ----

def _buf(n):
    return C.create_string_buffer(n)


def sha512(msg):
    o = _buf(64)
    lib().afxo_sha512(o, msg, len(msg))
    return o.raw


def point_from_uniform(b64):
    o = _buf(32)
    lib().afxo_point_from_uniform(b64, o)
    return o.raw


def point_decode_encode(b):
    o = _buf(32)
    ok = lib().afxo_point_decode_encode(b, o)
    return o.raw if ok else None


def point_add(a, b):
    o = _buf(32)
    return o.raw if lib().afxo_point_add(a, b, o) else None


def point_sub(a, b):
    o = _buf(32)
    return o.raw if lib().afxo_point_sub(a, b, o) else None


def point_scalarmult(s, p):
    o = _buf(32)
    return o.raw if lib().afxo_point_scalarmult(s, p, o) else None


def basepoint():
    o = _buf(32)
    lib().afxo_basepoint(o)
    return o.raw


def multiscalar(scalars, points, vartime=False):
    o = _buf(32)
    ok = lib().afxo_multiscalar(len(scalars), b"".join(scalars), b"".join(points), int(vartime), o)
    return o.raw if ok else None


def scalar_reduce_wide(b64):
    o = _buf(32)
    lib().afxo_scalar_reduce_wide(b64, o)
    return o.raw


def scalar_muladd(a, b, c):
    o = _buf(32)
    lib().afxo_scalar_muladd(a, b, c, o)
    return o.raw


def scalar_neg(a):
    o = _buf(32)
    lib().afxo_scalar_neg(a, o)
    return o.raw


def keccak_f1600(state200):
    o = C.create_string_buffer(bytes(state200), 200)
    lib().afxo_keccak_f1600(o)
    return o.raw


def merlin_simple(label, l1, m1, l2, outlen):
    o = _buf(outlen)
    lib().afxo_merlin_simple(label, len(label), l1, len(l1), m1, len(m1), l2, len(l2), o, outlen)
    return o.raw


def merlin_script(label, ops, fields=()):
    """a scripted merlin transcript through the oracle's strobe / merlin layer (afxo_merlin_script).  ops: ("append", label, bytes) |
    ("append_field", label, index) | ("challenge", label, n) | ("append_last_challenge", label); returns the list of challenges"""
    u32 = lambda v: int(v).to_bytes(4, "little")
    bs = lambda b: u32(len(b)) + bytes(b)
    script, sizes = bytes([1]) + bs(label), []
    for op in ops:
        if op[0] == "append":
            script += bytes([2]) + bs(op[1]) + bs(op[2])
        elif op[0] == "append_field":
            script += bytes([3]) + bs(op[1]) + u32(op[2])
        elif op[0] == "challenge":
            script += bytes([4]) + bs(op[1]) + u32(op[2])
            sizes.append(op[2])
        elif op[0] == "append_last_challenge":
            script += bytes([5]) + bs(op[1])
        else:
            raise ValueError(op[0])
    flat = b"".join(bytes(f) for f in fields)
    assert len(flat) == 32 * len(fields)
    o = _buf(max(1, sum(sizes)))
    lib().afxo_merlin_script.restype = C.c_long
    n = lib().afxo_merlin_script(script, len(script), flat, len(fields), o, sum(sizes))
    if n != sum(sizes):
        raise ValueError("afxo_merlin_script: %d" % n)
    out, at = [], 0
    for s in sizes:
        out.append(o.raw[at:at + s])
        at += s
    return out


def system_parameters_generate(n, stream):
    out = _buf(lib().afxo_sizeof_system_parameters(n))
    used = lib().afxo_system_parameters_generate(n, stream, len(stream), out)
    if used < 0:
        raise ValueError("hash_and_pray failed: %d" % used)
    return out.raw, used


def issuer_new(params, draws):
    n = int.from_bytes(params[:4], "little")
    assert len(draws) == 64 * (4 + n)
    key = _buf(lib().afxo_sizeof_secret_key(n))
    ip = _buf(64)
    rc = lib().afxo_issuer_new(params, len(params), draws, key, ip)
    if rc != 0:
        raise ValueError("issuer_new failed")
    return key.raw, ip.raw


class Ctx:
    def __init__(self, params, key=None, issuer_params=None):
        self.h = lib().afxo_ctx_new(params, len(params), key, len(key) if key else 0, issuer_params)
        if not self.h:
            raise ValueError("bad parameters / key")
        self.n = lib().afxo_ctx_n(self.h)

    def __del__(self):
        if getattr(self, "h", None):
            lib().afxo_ctx_free(self.h)
            self.h = None

    def set_strict(self, strict):
        lib().afxo_ctx_set_strict(self.h, 1 if strict else 0)

    def keypair_derive(self, master_secret):
        o = _buf(128)
        lib().afxo_keypair_derive(self.h, master_secret, o)
        return o.raw

    def issue(self, kinds, values, t_wide, U_wide, seed):
        """values: list of 96-byte records.  Returns (status, t, U, V, challenge, [responses])"""
        n = len(kinds)
        t, U, V, ch = _buf(32), _buf(32), _buf(32), _buf(32)
        resp = _buf(32 * (self.n + 5))
        st = lib().afxo_issue(self.h, n, bytes(kinds), b"".join(values), t_wide, U_wide, seed, t, U, V, ch, resp)
        return st, t.raw, U.raw, V.raw, ch.raw, [resp.raw[32 * i:32 * i + 32] for i in range(self.n + 5)]

    def issuance_verify(self, kinds, values, t, U, V, ch, responses):
        return lib().afxo_issuance_verify(self.h, len(kinds), bytes(kinds), b"".join(values), t, U, V, ch,
                                          b"".join(responses), len(responses))

    def show(self, kinds, values, t, U, V, keypair, z_wide, seed, enc_seeds=b""):
        p = Presentation()
        st = lib().afxo_show(self.h, len(kinds), bytes(kinds), b"".join(values), t, U, V, keypair, z_wide, seed,
                             enc_seeds, C.byref(p))
        return st, p

    def verify_presentation(self, p):
        return lib().afxo_verify_presentation(self.h, C.byref(p))

    def verify_encryption_proof(self, e):
        return lib().afxo_verify_encryption_proof(self.h, C.byref(e))


def plaintext_from_bytes(msg30):
    o = _buf(96)
    ctr = lib().afxo_plaintext_from_bytes(msg30, o)
    return o.raw, ctr


def encode_to_group(data):
    o = _buf(32)
    ctr = lib().afxo_encode_to_group(data, len(data), o)
    return o.raw, ctr


def decode_from_group(pt):
    o = _buf(30)
    ctr = lib().afxo_decode_from_group(pt, o)
    return o.raw, ctr


def encrypt(keypair, plaintext):
    o = _buf(64)
    rc = lib().afxo_encrypt(keypair, plaintext, o)
    return o.raw if rc == 0 else None


def decrypt(keypair, ciphertext):
    o = _buf(96)
    rc = lib().afxo_decrypt(keypair, ciphertext, o)
    return (rc, o.raw)


def debug_reset():
    lib().afxo_debug_reset()


def debug_last():
    commits = _buf(32 * 48)
    n = C.c_int(0)
    ch = _buf(32)
    lib().afxo_debug_last(commits, C.byref(n), ch)
    return [commits.raw[32 * i:32 * i + 32] for i in range(n.value)], ch.raw


# ---- batch forms over numpy struct-of-arrays (tests/test_gpu_full_size.py) ----

def host_threads():
    """cores this process may use (affinity mask): the oracle's batch forms split statically over that many threads"""
    return max(1, len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1))


def verify_presentations_traced(ctx, shape, soa, count, threads=None):
    """(status[count], trace[1 + n_enc_proofs, count, 32], reached[1 + n_enc_proofs, count]) from afxo_verify_presentations_soa_traced;
    shape / soa: this module's Shape / PresentationSoA (or byte-compatible ctypes structures of the engine's mirror)"""
    import numpy as np
    rows = 1 + shape.n_enc_proofs
    status, trace, reached = np.full(count, 255, np.uint8), np.zeros((rows, count, 32), np.uint8), np.zeros((rows, count), np.uint8)
    rc = lib().afxo_verify_presentations_soa_traced(ctx.h, C.byref(Shape.from_buffer_copy(bytes(shape))), C.byref(PresentationSoA.from_buffer_copy(bytes(soa))),
                                                    count, status.ctypes.data, trace.ctypes.data, reached.ctypes.data, threads or host_threads())
    assert rc == 0
    return status, trace, reached


def issue_soa(ctx, kinds, values, t_wide, U_wide, seed, threads=None):
    """Issuer::issue over a batch: values [n, count, 32], t_wide / U_wide [count, 64], seed [count, 32] -> (dict t U V challenge responses, status)"""
    import numpy as np
    n, count = values.shape[0], values.shape[1]
    o = {k: np.zeros((count, 32), np.uint8) for k in ("t", "U", "V", "challenge")}
    o["responses"] = np.zeros((ctx.n + 5, count, 32), np.uint8)
    status = np.full(count, 255, np.uint8)
    a = [np.ascontiguousarray(x, dtype=np.uint8) for x in (values, t_wide, U_wide, seed)]
    rc = lib().afxo_issue_soa(ctx.h, n, bytes(kinds), *(x.ctypes.data for x in a), count, *(o[k].ctypes.data for k in ("t", "U", "V", "challenge", "responses")),
                              status.ctypes.data, threads or host_threads())
    assert rc == 0
    return o, status


def verify_issuances_traced(ctx, kinds, values, iss, threads=None):
    """CredentialIssuance::verify over a batch -> (status[count], trace[count, 32], reached[count])"""
    import numpy as np
    n, count = values.shape[0], values.shape[1]
    a = [np.ascontiguousarray(x, dtype=np.uint8) for x in (values, iss["t"], iss["U"], iss["V"], iss["challenge"], iss["responses"])]
    status, trace, reached = np.full(count, 255, np.uint8), np.zeros((count, 32), np.uint8), np.zeros(count, np.uint8)
    rc = lib().afxo_verify_issuances_soa_traced(ctx.h, n, bytes(kinds), *(x.ctypes.data for x in a), a[5].shape[0], count, status.ctypes.data,
                                                trace.ctypes.data, reached.ctypes.data, threads or host_threads())
    assert rc == 0
    return status, trace, reached
