"""Replay of recorded ristretto255 group operations through libsodium (an implementation independent of oracle/ and of
tests/pyref).  Only available where libsodium is (the build container: /opt/conda/lib/libsodium.so.23, 1.0.18); the GPU box
has none, and nothing shipped depends on it.  Used by tests/gen_golden.py (flows.json is not written unless every group
operation of every flow replays) and by tests/test_pyref_cross_check.py."""
import ctypes as C
import os

CANDIDATES = ("/opt/conda/lib/libsodium.so.23", "libsodium.so.23", "libsodium.so")


def load():
    for name in CANDIDATES:
        try:
            lib = C.CDLL(name)
        except OSError:
            continue
        if hasattr(lib, "crypto_core_ristretto255_add") and lib.sodium_init() >= 0:
            return lib
    return None


def _mul(sod, k, enc):
    """k * P; libsodium masks bit 255 of the scalar (k < l here) and returns -1 when the result is the identity"""
    if k == 0 or enc == bytes(32):
        return bytes(32)
    o = C.create_string_buffer(32)
    rc = sod.crypto_scalarmult_ristretto255(o, k.to_bytes(32, "little"), enc)
    return o.raw if rc == 0 else bytes(32)


def _add(sod, p, q, sub=False):
    o = C.create_string_buffer(32)
    fn = sod.crypto_core_ristretto255_sub if sub else sod.crypto_core_ristretto255_add
    assert fn(o, p, q) == 0, "libsodium rejects an input encoding"
    return o.raw


def replay(sod, records):
    """records: tests/pyref/ristretto.TRACE entries.  Returns the number of operations replayed; raises AssertionError on the
    first result libsodium computes differently."""
    n = {"mul": 0, "msm": 0, "add": 0, "sub": 0, "neg": 0}
    for op, scalars, points, out in records:
        if op == "mul":
            got = _mul(sod, scalars[0], points[0])
        elif op == "msm":
            got = bytes(32)
            for k, p in zip(scalars, points):
                got = _add(sod, got, _mul(sod, k, p))
        elif op == "add":
            got = _add(sod, points[0], points[1])
        elif op == "sub":
            got = _add(sod, points[0], points[1], sub=True)
        elif op == "neg":
            got = _add(sod, bytes(32), points[0], sub=True)
        else:
            raise ValueError(op)
        assert got == out, (op, [hex(k) for k in scalars], [p.hex() for p in points], out.hex(), got.hex())
        n[op] += 1
    return n


def replay_flows(sod, flows):
    """every group operation tests/pyref performs while it replays `flows` (issue, issuance verification, show, presentation
    verification: tags, messages, commitments of both provers, ciphertexts, Z, every recomputed commitment)"""
    from tests.pyref import ristretto as R
    from tests.test_pyref_cross_check import test_every_flow_replays_byte_for_byte
    R.TRACE = []
    try:
        test_every_flow_replays_byte_for_byte(flows)
        records = R.TRACE
    finally:
        R.TRACE = None
    return replay(sod, records)
