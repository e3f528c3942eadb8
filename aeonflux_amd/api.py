"""The reference crate's user-facing API, mirrored over the GPU engine (names and semantics follow
/root/reference/src/{parameters,issuer,user,credential,symmetric}.rs so the README flow reads the same).

Every object method takes/returns single items like the crate; underneath each call is a batch of one through the
C ABI (the batch forms in aeonflux_amd.batch are what a server uses).  `rng` is any object with
`fill_bytes(n) -> bytes` standing in for the crate's `CryptoRng + RngCore` arguments; zkp's hidden thread_rng()
draws are taken from the same rng (SURVEY.md §8b "Randomness").
"""
import ctypes as C

import numpy as np

from . import (ATTR_EITHER_POINT, ATTR_PUBLIC_POINT, ATTR_PUBLIC_SCALAR, ATTR_SECRET_POINT, ATTR_SECRET_SCALAR, ST_MAC_CREATION,
               ST_NO_SYMMETRIC_KEY, ST_OK, ST_UNDECRYPTABLE, Context, KeypairsSoA, batch, check, lib)


class CredentialError(Exception):
    """src/errors.rs:73-89"""


class VerificationFailure(CredentialError):
    pass


class MacCreation(CredentialError):
    pass


class NoSymmetricKey(CredentialError):
    pass


class UndecryptableAttribute(CredentialError):
    pass


def _col(b):
    return np.frombuffer(b, dtype=np.uint8).reshape(1, 32).copy()


class SystemParameters:
    """src/parameters.rs:62-76"""

    def __init__(self, data):
        self.bytes = bytes(data)
        self.NUMBER_OF_ATTRIBUTES = int.from_bytes(self.bytes[:4], "little")

    @staticmethod
    def generate(rng, number_of_attributes, device=0):
        """SystemParameters::generate == hash_and_pray (src/parameters.rs:196-336)"""
        g = max(3, number_of_attributes)
        total = 8 + g + number_of_attributes
        size = 4 + 32 * (1 + total)
        stream = b""
        while True:
            stream += rng.fill_bytes(32 * 24 * total)   # ~16 attempts expected per generator
            out = C.create_string_buffer(size)
            used = C.c_size_t(0)
            rc = lib().afx_system_parameters_generate(device, number_of_attributes, stream, len(stream), out, size, C.byref(used))
            if rc == 0:
                if hasattr(rng, "unread"):
                    rng.unread(len(stream) - used.value)
                return SystemParameters(out.raw)
            if rc != -1 or b"exhausted" not in (lib().afx_last_error() or b""):
                check(rc)

    def to_bytes(self):
        return self.bytes

    @staticmethod
    def from_bytes(b):
        return SystemParameters(b)


class Plaintext:
    """src/symmetric.rs:89-96"""

    def __init__(self, M1, M2, m3):
        self.M1, self.M2, self.m3 = bytes(M1), bytes(M2), bytes(m3)


class Attribute:
    """src/amacs.rs:168-179"""

    def __init__(self, kind, value):
        self.kind, self.value = kind, value   # value: 32-byte scalar, 32-byte point, or Plaintext

    def cell(self):
        return self.value.M1 if isinstance(self.value, Plaintext) else self.value


class CredentialRequest:
    def __init__(self, attributes):
        self.attributes = attributes


class CredentialRequestConstructor:
    """src/user.rs:26-134"""

    def __init__(self, system_parameters, ctx):
        self.parameters, self.attributes, self._ctx = system_parameters, [], ctx

    def append_revealed_scalar(self, scalar):
        self.attributes.append(Attribute(ATTR_PUBLIC_SCALAR, bytes(scalar)))

    def append_revealed_point(self, point):
        self.attributes.append(Attribute(ATTR_PUBLIC_POINT, bytes(point)))

    def append_plaintext(self, message):
        """Plaintext::from_slice: 30-byte chunks, zero padded (src/symmetric.rs:118-132)"""
        out = []
        for o in range(0, len(message), 30):
            chunk = bytes(message[o:o + 30]).ljust(30, b"\0")
            out.append(plaintext_from_bytes(self._ctx, chunk))
            self.attributes.append(Attribute(ATTR_EITHER_POINT, out[-1]))
        return out

    def finish(self):
        return CredentialRequest(self.attributes)


def plaintext_from_bytes(ctx, msg30):
    msgs = np.frombuffer(bytes(msg30), dtype=np.uint8).copy()
    M1, M2, m3 = (np.zeros(32, np.uint8) for _ in range(3))
    check(lib().afx_plaintexts_from_bytes(ctx.h, msgs.ctypes.data, 1, M1.ctypes.data, M2.ctypes.data, m3.ctypes.data, None))
    return Plaintext(M1.tobytes(), M2.tobytes(), m3.tobytes())


class Keypair:
    """symmetric::Keypair (src/symmetric.rs:72-81)"""

    def __init__(self, a, a0, a1, pk):
        self.a, self.a0, self.a1, self.pk = a, a0, a1, pk

    @staticmethod
    def derive(master_secret, ctx):
        ms = np.frombuffer(bytes(master_secret), dtype=np.uint8).copy()
        o = [np.zeros(32, np.uint8) for _ in range(4)]
        check(lib().afx_keypairs_derive(ctx.h, ms.ctypes.data, 1, *(x.ctypes.data for x in o)))
        return Keypair(*(x.tobytes() for x in o))

    @staticmethod
    def generate(ctx, rng):
        master_secret = rng.fill_bytes(64)
        return Keypair.derive(master_secret, ctx), master_secret

    def _soa(self):
        cols = [_col(x) for x in (self.a, self.a0, self.a1, self.pk)]
        return KeypairsSoA(*(c.ctypes.data for c in cols)), cols

    def encrypt(self, ctx, plaintext):
        soa, keep = self._soa()
        i = [_col(x) for x in (plaintext.M1, plaintext.M2, plaintext.m3)]
        E1, E2, st = np.zeros(32, np.uint8), np.zeros(32, np.uint8), np.zeros(1, np.uint8)
        check(lib().afx_encrypt(ctx.h, C.byref(soa), i[0].ctypes.data, i[1].ctypes.data, i[2].ctypes.data, 1, E1.ctypes.data, E2.ctypes.data, st.ctypes.data))
        return E1.tobytes(), E2.tobytes()

    def decrypt(self, ctx, ciphertext):
        soa, keep = self._soa()
        E1, E2 = _col(ciphertext[0]), _col(ciphertext[1])
        o = [np.zeros(32, np.uint8) for _ in range(3)]
        msg, st = np.zeros(30, np.uint8), np.zeros(1, np.uint8)
        check(lib().afx_decrypt(ctx.h, C.byref(soa), E1.ctypes.data, E2.ctypes.data, 1, o[0].ctypes.data, o[1].ctypes.data, o[2].ctypes.data,
                                msg.ctypes.data, st.ctypes.data))
        if st[0] != ST_OK:
            raise UndecryptableAttribute()
        return Plaintext(*(x.tobytes() for x in o)), msg.tobytes()


class AnonymousCredential:
    """src/credential.rs:30-33"""

    def __init__(self, t, U, V, attributes):
        self.t, self.U, self.V, self.attributes = t, U, V, attributes

    def hide_attribute(self, index):
        """src/credential.rs:77-97"""
        a = self.attributes[index]
        if a.kind == ATTR_PUBLIC_SCALAR:
            a.kind = ATTR_SECRET_SCALAR
        elif a.kind == ATTR_EITHER_POINT:
            a.kind = ATTR_SECRET_POINT
        elif a.kind == ATTR_PUBLIC_POINT:
            raise ValueError("Public point attributes cannot be converted to secret point attributes")

    def reveal_attribute(self, index):
        """src/credential.rs:53-70"""
        a = self.attributes[index]
        if a.kind == ATTR_SECRET_SCALAR:
            a.kind = ATTR_PUBLIC_SCALAR
        elif a.kind == ATTR_SECRET_POINT:
            a.kind = ATTR_EITHER_POINT

    def show(self, ctx, keypair, rng):
        """AnonymousCredential::show (src/credential.rs:37-46).  Returns (afx Shape, presentation dict of [1,32] columns)."""
        n = len(self.attributes)
        kinds = [a.kind for a in self.attributes]
        values = np.stack([_col(a.cell()) for a in self.attributes])
        M2 = np.stack([_col(a.value.M2) if isinstance(a.value, Plaintext) else np.zeros((1, 32), np.uint8) for a in self.attributes])
        m3 = np.stack([_col(a.value.m3) if isinstance(a.value, Plaintext) else np.zeros((1, 32), np.uint8) for a in self.attributes])
        nsp = sum(1 for k in kinds if k == ATTR_SECRET_POINT)
        z_wide = np.frombuffer(rng.fill_bytes(64), np.uint8).reshape(1, 64)
        seed = np.frombuffer(rng.fill_bytes(32), np.uint8).reshape(1, 32)
        es = np.frombuffer(rng.fill_bytes(32 * nsp), np.uint8).reshape(nsp, 1, 32) if nsp else None
        kp = None if keypair is None else {f: _col(getattr(keypair, f)) for f in ("a", "a0", "a1", "pk")}
        pres, shape, st = batch.show(ctx, kinds, values, _col(self.t), _col(self.U), _col(self.V), kp, z_wide, seed, es, M2, m3)
        if st[0] == ST_NO_SYMMETRIC_KEY:
            raise NoSymmetricKey()
        if st[0] != ST_OK:
            raise CredentialError("show failed: %d" % st[0])
        return shape, pres


class CredentialIssuance:
    """src/issuer.rs:42-58"""

    def __init__(self, proof, credential):
        self.proof, self.credential = proof, credential   # proof = (challenge, [responses])

    def verify(self, ctx):
        """CredentialIssuance::verify: returns the credential or raises VerificationFailure"""
        c = self.credential
        values = np.stack([_col(a.cell()) for a in c.attributes])
        iss = dict(t=_col(c.t), U=_col(c.U), V=_col(c.V), challenge=_col(self.proof[0]), responses=np.stack([_col(r) for r in self.proof[1]]))
        st = batch.verify_issuances(ctx, [a.kind for a in c.attributes], values, iss)
        if st[0] != ST_OK:
            raise VerificationFailure()
        return c


class Issuer:
    """src/issuer.rs:61-147"""

    def __init__(self, system_parameters, key_bytes, issuer_parameters, device=0):
        self.system_parameters, self.issuer_parameters, self._key = system_parameters, bytes(issuer_parameters), bytes(key_bytes)
        self.ctx = Context(system_parameters.to_bytes(), self._key, self.issuer_parameters, device)

    @staticmethod
    def new(system_parameters, rng, device=0):
        """Issuer::new: SecretKey::generate (4+n draws of 64 B, src/amacs.rs:89-107) + IssuerParameters::generate"""
        n = system_parameters.NUMBER_OF_ATTRIBUTES
        draws = np.frombuffer(rng.fill_bytes(64 * (4 + n)), np.uint8).reshape(4 + n, 64)
        bare = Context(system_parameters.to_bytes(), None, bytes(64), device)   # identity placeholders; only primitives are used
        scalars = batch.scalars_from_wide(bare, draws)
        bare.close()
        ks = system_parameters.to_bytes()[:4] + scalars.tobytes()
        W, ip = C.create_string_buffer(32), C.create_string_buffer(64)
        check(lib().afx_issuer_keygen(device, system_parameters.to_bytes(), len(system_parameters.to_bytes()), ks, len(ks), W, ip))
        return Issuer(system_parameters, ks + W.raw, ip.raw, device)

    def issue(self, request, rng):
        """Issuer::issue (src/issuer.rs:111-124)"""
        attrs = request.attributes
        values = np.stack([_col(a.cell()) for a in attrs]) if attrs else np.zeros((0, 1, 32), np.uint8)
        rb = lambda k: np.frombuffer(rng.fill_bytes(k), np.uint8).reshape(1, k)
        o, st = batch.issue(self.ctx, [a.kind for a in attrs], values, rb(64), rb(64), rb(32))
        if st[0] == ST_MAC_CREATION:
            raise MacCreation()
        cred = AnonymousCredential(o["t"][0].tobytes(), o["U"][0].tobytes(), o["V"][0].tobytes(), attrs)
        return CredentialIssuance((o["challenge"][0].tobytes(), [o["responses"][k, 0].tobytes() for k in range(o["responses"].shape[0])]), cred)

    def verify(self, presentation):
        """Issuer::verify (src/issuer.rs:141-147); presentation = (shape, dict) from AnonymousCredential.show"""
        shape, pres = presentation
        st = batch.verify_presentations(self.ctx, shape, pres)
        if st[0] != ST_OK:
            raise VerificationFailure()
