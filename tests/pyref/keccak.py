"""Keccak-f[1600], STROBE-128 as merlin uses it, and merlin's Transcript / TranscriptRng (merlin 2.x/3.x [3P])."""

_RC = [0x0000000000000001, 0x0000000000008082, 0x800000000000808A, 0x8000000080008000, 0x000000000000808B, 0x0000000080000001,
       0x8000000080008081, 0x8000000000008009, 0x000000000000008A, 0x0000000000000088, 0x0000000080008009, 0x000000008000000A,
       0x000000008000808B, 0x800000000000008B, 0x8000000000008089, 0x8000000000008003, 0x8000000000008002, 0x8000000000000080,
       0x000000000000800A, 0x800000008000000A, 0x8000000080008081, 0x8000000000008080, 0x0000000080000001, 0x8000000080008008]
_ROT = [[0, 36, 3, 41, 18], [1, 44, 10, 45, 2], [62, 6, 43, 15, 61], [28, 55, 25, 21, 56], [27, 20, 39, 8, 14]]   # [x][y]
_M = (1 << 64) - 1


def _rol(v, n):
    n %= 64
    return ((v << n) | (v >> (64 - n))) & _M if n else v


def keccak_f1600(lanes):
    """lanes: 25 ints, index x + 5*y"""
    a = list(lanes)
    for rc in _RC:
        c = [a[x] ^ a[x + 5] ^ a[x + 10] ^ a[x + 15] ^ a[x + 20] for x in range(5)]
        d = [c[(x - 1) % 5] ^ _rol(c[(x + 1) % 5], 1) for x in range(5)]
        a = [a[i] ^ d[i % 5] for i in range(25)]
        b = [0] * 25
        for x in range(5):
            for y in range(5):
                b[y + 5 * ((2 * x + 3 * y) % 5)] = _rol(a[x + 5 * y], _ROT[x][y])
        a = [b[i] ^ ((~b[(i % 5 + 1) % 5 + 5 * (i // 5)]) & b[(i % 5 + 2) % 5 + 5 * (i // 5)]) for i in range(25)]
        a[0] ^= rc
    return a


def _permute(state):
    lanes = [int.from_bytes(state[8 * i:8 * i + 8], "little") for i in range(25)]
    out = keccak_f1600(lanes)
    for i in range(25):
        state[8 * i:8 * i + 8] = out[i].to_bytes(8, "little")


R = 166
FLAG_I, FLAG_A, FLAG_C, FLAG_T, FLAG_M, FLAG_K = 1, 2, 4, 8, 16, 32


class Strobe128:
    def __init__(self, protocol_label=None, _copy=None):
        if _copy is not None:
            self.st, self.pos, self.pos_begin, self.cur_flags = bytearray(_copy.st), _copy.pos, _copy.pos_begin, _copy.cur_flags
            return
        self.st = bytearray(200)
        self.st[0:6] = bytes([1, R + 2, 1, 0, 1, 96])
        self.st[6:18] = b"STROBEv1.0.2"
        _permute(self.st)
        self.pos, self.pos_begin, self.cur_flags = 0, 0, 0
        self.meta_ad(protocol_label, False)

    def clone(self):
        return Strobe128(_copy=self)

    def _run_f(self):
        self.st[self.pos] ^= self.pos_begin
        self.st[self.pos + 1] ^= 0x04
        self.st[R + 1] ^= 0x80
        _permute(self.st)
        self.pos, self.pos_begin = 0, 0

    def _absorb(self, data):
        for b in data:
            self.st[self.pos] ^= b
            self.pos += 1
            if self.pos == R:
                self._run_f()

    def _overwrite(self, data):
        for b in data:
            self.st[self.pos] = b
            self.pos += 1
            if self.pos == R:
                self._run_f()

    def _squeeze(self, n):
        out = bytearray()
        for _ in range(n):
            out.append(self.st[self.pos])
            self.st[self.pos] = 0
            self.pos += 1
            if self.pos == R:
                self._run_f()
        return bytes(out)

    def _begin_op(self, flags, more):
        if more:
            assert self.cur_flags == flags
            return
        assert not flags & FLAG_T
        old_begin = self.pos_begin
        self.pos_begin = self.pos + 1
        self.cur_flags = flags
        self._absorb(bytes([old_begin, flags]))
        if flags & (FLAG_C | FLAG_K) and self.pos != 0:
            self._run_f()

    def meta_ad(self, data, more):
        self._begin_op(FLAG_M | FLAG_A, more)
        self._absorb(data)

    def ad(self, data, more):
        self._begin_op(FLAG_A, more)
        self._absorb(data)

    def prf(self, n, more):
        self._begin_op(FLAG_I | FLAG_A | FLAG_C, more)
        return self._squeeze(n)

    def key(self, data, more):
        self._begin_op(FLAG_A | FLAG_C, more)
        self._overwrite(data)


def _u32(n):
    return n.to_bytes(4, "little")


class Transcript:
    def __init__(self, label=None, _strobe=None):
        if _strobe is not None:
            self.strobe = _strobe
            return
        self.strobe = Strobe128(b"Merlin v1.0")
        self.append_message(b"dom-sep", label)

    def append_message(self, label, message):
        self.strobe.meta_ad(label, False)
        self.strobe.meta_ad(_u32(len(message)), True)
        self.strobe.ad(message, False)

    def challenge_bytes(self, label, n):
        self.strobe.meta_ad(label, False)
        self.strobe.meta_ad(_u32(n), True)
        return self.strobe.prf(n, False)

    def build_rng(self):
        return TranscriptRngBuilder(self.strobe.clone())


class TranscriptRngBuilder:
    def __init__(self, strobe):
        self.strobe = strobe

    def rekey_with_witness_bytes(self, label, witness):
        self.strobe.meta_ad(label, False)
        self.strobe.meta_ad(_u32(len(witness)), True)
        self.strobe.key(witness, False)
        return self

    def finalize(self, random_bytes32):
        """merlin draws 32 bytes from the external rng; here they are an explicit input"""
        assert len(random_bytes32) == 32
        self.strobe.meta_ad(b"rng", False)
        self.strobe.key(random_bytes32, False)
        return TranscriptRng(self.strobe)


class TranscriptRng:
    def __init__(self, strobe):
        self.strobe = strobe

    def fill_bytes(self, n):
        self.strobe.meta_ad(_u32(n), False)
        return self.strobe.prf(n, False)
