"""aeonflux_amd — MI355X (gfx950) batch engine for aeonflux's credential NIZKs.

Python mirror of the C ABI in include/aeonflux_gpu.h (ctypes over aeonflux_amd/lib/libaeonflux_gpu.so).
All arithmetic runs in HIP kernels; there is no CPU fallback: importing the engine without the built
library, or creating a context without a GPU, raises.  PyTorch is used by callers only for device memory
and process groups (bench.py); this module takes raw pointers.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libaeonflux_gpu.so")
MAX_ATTRIBUTES = 32

OK, E_BAD_ARGS, E_BAD_PARAMS, E_NO_DEVICE, E_HIP, E_NO_KEY = 0, -1, -2, -3, -4, -5
E_NO_MEMORY = -6
ST_OK, ST_VERIFICATION_FAILURE, ST_MAC_CREATION, ST_NO_SYMMETRIC_KEY, ST_UNDECRYPTABLE = 0, 1, 2, 3, 4
ATTR_PUBLIC_SCALAR, ATTR_SECRET_SCALAR, ATTR_PUBLIC_POINT, ATTR_EITHER_POINT, ATTR_SECRET_POINT = range(5)
# afx_ctx_set_plan_variants (tests: the alternatives among a small pass's equivalent plans and kernels)
VARIANT_SEGMENTS_1, VARIANT_SEGMENTS_2, VARIANT_SEGMENTS_4, VARIANT_ONE_WAVE_CHAINS, VARIANT_HASH_HALF_WAVE, VARIANT_NO_POINTSUM_TREE, VARIANT_SELFCHECK = \
    0x01, 0x02, 0x04, 0x08, 0x10, 0x20, 0x40
# Tests that drive whole scripts under one variant (tests/test_hostsim.py) set this before creating contexts: every Context made
# afterwards starts with these flags.  A hook of this python mirror, not of the library - which reads no variant from the environment.
DEFAULT_PLAN_VARIANTS = int(os.environ.get("AFX_TEST_PLAN_VARIANTS", "0"), 0)
ENC_PUBLIC_SCALAR, ENC_SECRET_SCALAR, ENC_PUBLIC_POINT, ENC_SECRET_POINT = range(4)


class AfxError(RuntimeError):
    def __init__(self, rc, msg):
        super().__init__("aeonflux_gpu error %d: %s" % (rc, msg))
        self.rc = rc


class Shape(C.Structure):
    _fields_ = [("n_attributes", C.c_uint32), ("kinds", C.c_uint8 * MAX_ATTRIBUTES), ("n_responses", C.c_uint32),
                ("n_hidden_scalars", C.c_uint32), ("hidden_scalar_indices", C.c_uint16 * MAX_ATTRIBUTES),
                ("n_enc_proofs", C.c_uint32), ("enc_indices", C.c_uint16 * MAX_ATTRIBUTES)]


class EncProofSoA(C.Structure):
    _fields_ = [(k, C.c_void_p) for k in ("challenge", "responses", "pk", "E1", "E2", "C_y_1", "C_y_2", "C_y_3", "C_y_2p")]


class PresentationSoA(C.Structure):
    _fields_ = [(k, C.c_void_p) for k in ("challenge", "responses", "C_x_0", "C_x_1", "C_V", "C_y", "attr_values")] + \
               [("enc", C.POINTER(EncProofSoA))]


class PresentationGroup(C.Structure):
    """afx_presentation_group: one same-shape group of a mixed request (afx_verify_presentations_mixed)"""
    _fields_ = [("shape", Shape), ("batch", PresentationSoA), ("count", C.c_size_t), ("positions", C.POINTER(C.c_uint64))]


class AttributesSoA(C.Structure):
    _fields_ = [("n_attributes", C.c_uint32), ("kinds", C.c_uint8 * MAX_ATTRIBUTES), ("values", C.c_void_p)]


class IssueRandomness(C.Structure):
    _fields_ = [(k, C.c_void_p) for k in ("t_wide", "U_wide", "rng_seed")]


class IssuanceSoA(C.Structure):
    _fields_ = [(k, C.c_void_p) for k in ("t", "U", "V", "challenge", "responses")]


class CredentialsSoA(C.Structure):
    _fields_ = [("n_attributes", C.c_uint32), ("kinds", C.c_uint8 * MAX_ATTRIBUTES)] + \
               [(k, C.c_void_p) for k in ("values", "M2", "m3", "t", "U", "V")]


class KeypairsSoA(C.Structure):
    _fields_ = [(k, C.c_void_p) for k in ("a", "a0", "a1", "pk")]


class ShowRandomness(C.Structure):
    _fields_ = [(k, C.c_void_p) for k in ("z_wide", "rng_seed", "enc_seeds")]


class EncProofOut(C.Structure):
    _fields_ = [(k, C.c_void_p) for k in ("challenge", "responses", "pk", "E1", "E2", "C_y_1", "C_y_2", "C_y_3", "C_y_2p")]


class PresentationOut(C.Structure):
    _fields_ = [(k, C.c_void_p) for k in ("challenge", "responses", "C_x_0", "C_x_1", "C_V", "C_y", "attr_values")] + \
               [("enc", C.POINTER(EncProofOut))]


class IssueGroup(C.Structure):
    """afx_issue_group: the requests of one attribute layout in a mixed call (afx_issue_mixed)"""
    _fields_ = [("requests", AttributesSoA), ("rnd", IssueRandomness), ("out", IssuanceSoA), ("count", C.c_size_t), ("positions", C.POINTER(C.c_uint64))]


class IssuanceGroup(C.Structure):
    """afx_issuance_group: the issuances of one layout in a mixed call (afx_verify_issuances_mixed)"""
    _fields_ = [("attrs", AttributesSoA), ("issuances", IssuanceSoA), ("n_responses", C.c_uint32), ("count", C.c_size_t), ("positions", C.POINTER(C.c_uint64))]


class ShowGroup(C.Structure):
    """afx_show_group: the credentials of one layout in a mixed call (afx_show_mixed)"""
    _fields_ = [("creds", CredentialsSoA), ("keypairs", C.POINTER(KeypairsSoA)), ("rnd", ShowRandomness), ("out", PresentationOut), ("shape_out", Shape),
                ("count", C.c_size_t), ("positions", C.POINTER(C.c_uint64))]


class CoalescingStats(C.Structure):
    """afx_coalescing_stats: how concurrent small calls on one context were collected (afx_ctx_set_coalescing)"""
    _fields_ = [(k, C.c_uint64) for k in ("sessions", "calls", "items", "appended_calls", "max_calls", "leader_waits", "staging_ns", "launch_ns")]


class PlanCacheStats(C.Structure):
    """afx_plan_cache_stats"""
    _fields_ = [(k, C.c_uint64) for k in ("hits", "misses", "evictions", "entries", "bytes")]


_LIB = None


def lib():
    """Load the HIP engine.  Raises if it has not been built (python __graft_entry__.py build)."""
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError("aeonflux_amd: %s is missing — build it with `make -C aeonflux_amd/csrc`; "
                              "there is no CPU fallback" % LIB_PATH)
        _LIB = C.CDLL(LIB_PATH)
        _LIB.afx_last_error.restype = C.c_char_p
        _LIB.afx_ctx_stream.restype = C.c_void_p
        _LIB.afx_ctx_stream.argtypes = [C.c_void_p]
        _LIB.afx_ctx_n_attributes.restype = C.c_uint32
        _LIB.afx_ctx_n_attributes.argtypes = [C.c_void_p]
        _LIB.afx_ctx_create.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t, C.c_char_p]
        _LIB.afx_ctx_destroy.argtypes = [C.c_void_p]
        _LIB.afx_ctx_destroy.restype = None
        for name in ("afx_verify_presentations", "afx_verify_presentations_dev"):
            getattr(_LIB, name).argtypes = [C.c_void_p, C.POINTER(Shape), C.POINTER(PresentationSoA), C.c_size_t, C.c_void_p]
        for name in ("afx_verify_encryption_proofs", "afx_verify_encryption_proofs_dev"):
            getattr(_LIB, name).argtypes = [C.c_void_p, C.c_uint16, C.POINTER(EncProofSoA), C.c_size_t, C.c_void_p]
        _LIB.afx_points_from_uniform_bytes.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
        _LIB.afx_scalars_from_wide_bytes.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
        _LIB.afx_points_validate.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]
        _LIB.afx_multiscalar_mul.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]
        for name in ("afx_issue", "afx_issue_dev"):
            if hasattr(_LIB, name):
                getattr(_LIB, name).argtypes = [C.c_void_p, C.POINTER(AttributesSoA), C.POINTER(IssueRandomness), C.c_size_t,
                                                C.POINTER(IssuanceSoA), C.c_void_p]
        for name in ("afx_verify_issuances", "afx_verify_issuances_dev"):
            if hasattr(_LIB, name):
                getattr(_LIB, name).argtypes = [C.c_void_p, C.POINTER(AttributesSoA), C.POINTER(IssuanceSoA), C.c_uint32, C.c_size_t, C.c_void_p]
        for name in ("afx_show", "afx_show_dev"):
            if hasattr(_LIB, name):
                getattr(_LIB, name).argtypes = [C.c_void_p, C.POINTER(CredentialsSoA), C.POINTER(KeypairsSoA), C.POINTER(ShowRandomness),
                                                C.c_size_t, C.POINTER(PresentationOut), C.POINTER(Shape), C.c_void_p]
        _LIB.afx_system_parameters_generate.argtypes = [C.c_int, C.c_uint32, C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t, C.POINTER(C.c_size_t)]
        _LIB.afx_plaintexts_from_bytes.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        _LIB.afx_keypairs_derive.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        _LIB.afx_encrypt.argtypes = [C.c_void_p, C.POINTER(KeypairsSoA), C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p]
        _LIB.afx_decrypt.argtypes = [C.c_void_p, C.POINTER(KeypairsSoA), C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        _LIB.afx_wire_header_bytes.restype = C.c_size_t
        _LIB.afx_wire_header_bytes.argtypes = [C.POINTER(Shape)]
        _LIB.afx_wire_cells_per_record.restype = C.c_uint32
        _LIB.afx_wire_cells_per_record.argtypes = [C.POINTER(Shape)]
        _LIB.afx_wire_parse.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(Shape), C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]
        _LIB.afx_verify_presentations_wire.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]
        _LIB.afx_verify_presentations_wire_range.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t, C.c_size_t, C.c_size_t, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]
        for name in ("afx_group_verify_presentations_wire", "afx_group_verify_presentations_mixed_wire"):
            getattr(_LIB, name).argtypes = [C.c_void_p, C.c_char_p, C.c_size_t, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]
        _LIB.afx_wire_pack_presentations.argtypes = [C.POINTER(Shape), C.POINTER(PresentationSoA), C.c_size_t, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]
        _LIB.afx_issuance_wire_pack.argtypes = [C.POINTER(AttributesSoA), C.POINTER(IssuanceSoA), C.c_uint32, C.c_size_t, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]
        _LIB.afx_wire_section_bytes.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(C.c_size_t)]
        _LIB.afx_verify_presentations_mixed_wire.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]
        for name in ("afx_verify_presentations_mixed", "afx_group_verify_presentations_mixed"):
            getattr(_LIB, name).argtypes = [C.c_void_p, C.POINTER(PresentationGroup), C.c_size_t, C.c_void_p, C.c_size_t]
        for name, grp in (("afx_issue_mixed", IssueGroup), ("afx_group_issue_mixed", IssueGroup), ("afx_verify_issuances_mixed", IssuanceGroup),
                          ("afx_group_verify_issuances_mixed", IssuanceGroup), ("afx_show_mixed", ShowGroup), ("afx_group_show_mixed", ShowGroup)):
            getattr(_LIB, name).argtypes = [C.c_void_p, C.POINTER(grp), C.c_size_t, C.c_void_p, C.c_size_t]
        _LIB.afx_ctx_set_pipelining.argtypes = [C.c_void_p, C.c_int]
        _LIB.afx_ctx_get_plan_stats.argtypes = [C.c_void_p, C.c_void_p]
        _LIB.afx_ctx_set_strict.argtypes = [C.c_void_p, C.c_int]
        _LIB.afx_ctx_set_fixed_key_schedule.argtypes = [C.c_void_p, C.c_int]
        _LIB.afx_ctx_set_secret_independent_addressing.argtypes = [C.c_void_p, C.c_int]
        _LIB.afx_ctx_set_chunk_items.argtypes = [C.c_void_p, C.c_uint32]
        _LIB.afx_ctx_set_small_batch_items.argtypes = [C.c_void_p, C.c_uint32]
        _LIB.afx_ctx_issuer_parameters.argtypes = [C.c_void_p, C.c_void_p]
        _LIB.afx_issuance_wire_header_bytes.restype = C.c_size_t
        _LIB.afx_issuance_wire_header_bytes.argtypes = [C.c_uint32]
        _LIB.afx_issuance_wire_parse.argtypes = [C.c_char_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        _LIB.afx_verify_issuances_wire.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p]
        _LIB.afx_ctx_set_challenge_trace.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t]
        _LIB.afx_ctx_get_challenge_trace.argtypes = [C.c_void_p, C.c_void_p]
        _LIB.afx_ctx_synchronize.argtypes = [C.c_void_p]
        _LIB.afx_ctx_set_timing.argtypes = [C.c_void_p, C.c_int]
        _LIB.afx_ctx_get_timing.argtypes = [C.c_void_p, C.c_char_p, C.POINTER(C.c_double), C.POINTER(C.c_uint64)]
        _LIB.afx_ctx_get_core_clock_mhz.argtypes = [C.c_void_p, C.POINTER(C.c_double)]
        _LIB.afx_ctx_get_core_clock_samples.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.c_uint32, C.POINTER(C.c_uint32)]
        _LIB.afx_ctx_set_coalescing.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32]
        _LIB.afx_ctx_set_host_copy_threads.argtypes = [C.c_void_p, C.c_uint32]
        _LIB.afx_ctx_set_plan_variants.argtypes = [C.c_void_p, C.c_uint32]
        _LIB.afx_merlin_challenges.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t, C.c_void_p, C.c_uint32, C.c_size_t, C.c_void_p]
        _LIB.afx_ctx_get_coalescing_stats.argtypes = [C.c_void_p, C.POINTER(CoalescingStats)]
        _LIB.afx_ctx_get_plan_cache_stats.argtypes = [C.c_void_p, C.POINTER(PlanCacheStats)]
        _LIB.afx_verify_presentations_range.argtypes = [C.c_void_p, C.POINTER(Shape), C.POINTER(PresentationSoA), C.c_size_t, C.c_size_t, C.c_size_t, C.c_void_p]
        _LIB.afx_issue_range.argtypes = [C.c_void_p, C.POINTER(AttributesSoA), C.POINTER(IssueRandomness), C.c_size_t, C.c_size_t, C.c_size_t,
                                         C.POINTER(IssuanceSoA), C.c_void_p]
        _LIB.afx_group_create.argtypes = [C.POINTER(C.c_void_p), C.POINTER(C.c_int), C.c_uint32, C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t, C.c_char_p]
        _LIB.afx_group_destroy.argtypes = [C.c_void_p]
        _LIB.afx_group_destroy.restype = None
        _LIB.afx_group_size.argtypes = [C.c_void_p]
        _LIB.afx_group_size.restype = C.c_uint32
        _LIB.afx_group_member.argtypes = [C.c_void_p, C.c_uint32]
        _LIB.afx_group_member.restype = C.c_void_p
        _LIB.afx_shard_bounds.argtypes = [C.c_size_t, C.c_uint32, C.c_uint32, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]
        _LIB.afx_shard_bounds.restype = None
        _LIB.afx_group_verify_presentations.argtypes = [C.c_void_p, C.POINTER(Shape), C.POINTER(PresentationSoA), C.c_size_t, C.c_void_p]
        _LIB.afx_verify_issuances_range.argtypes = [C.c_void_p, C.POINTER(AttributesSoA), C.POINTER(IssuanceSoA), C.c_uint32, C.c_size_t, C.c_size_t, C.c_size_t, C.c_void_p]
        _LIB.afx_group_verify_issuances.argtypes = [C.c_void_p, C.POINTER(AttributesSoA), C.POINTER(IssuanceSoA), C.c_uint32, C.c_size_t, C.c_void_p]
        _LIB.afx_show_range.argtypes = [C.c_void_p, C.POINTER(CredentialsSoA), C.POINTER(KeypairsSoA), C.POINTER(ShowRandomness), C.c_size_t, C.c_size_t, C.c_size_t,
                                        C.POINTER(PresentationOut), C.POINTER(Shape), C.c_void_p]
        _LIB.afx_group_show.argtypes = [C.c_void_p, C.POINTER(CredentialsSoA), C.POINTER(KeypairsSoA), C.POINTER(ShowRandomness), C.c_size_t,
                                        C.POINTER(PresentationOut), C.POINTER(Shape), C.c_void_p]
        _LIB.afx_group_issue.argtypes = [C.c_void_p, C.POINTER(AttributesSoA), C.POINTER(IssueRandomness), C.c_size_t, C.POINTER(IssuanceSoA), C.c_void_p]
        if hasattr(_LIB, "afx_issuer_keygen"):
            _LIB.afx_issuer_keygen.argtypes = [C.c_int, C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t, C.c_char_p, C.c_char_p]
    return _LIB


def check(rc):
    if rc != OK:
        raise AfxError(rc, (lib().afx_last_error() or b"").decode())


class Context:
    """An issuer-side (key given) or user-side (key=None) engine context on one GPU."""

    def __init__(self, sysparams, amacs_key, issuer_params, device=0, _borrowed=None, _group=None):
        if _borrowed is not None:   # a member of a Group: owned by the group, which clears self.h when it closes
            self.h, self._owned, self._group = _borrowed, False, _group
        else:
            h = C.c_void_p()
            check(lib().afx_ctx_create(C.byref(h), device, sysparams, len(sysparams), amacs_key, len(amacs_key) if amacs_key else 0,
                                       issuer_params))
            self.h, self._owned = h, True
        if DEFAULT_PLAN_VARIANTS:
            check(lib().afx_ctx_set_plan_variants(self.h, DEFAULT_PLAN_VARIANTS))
        self.n = lib().afx_ctx_n_attributes(self.h)
        self.device = device

    @property
    def stream(self):
        return lib().afx_ctx_stream(self.h)

    def close(self):
        if getattr(self, "h", None):
            if self._owned:
                lib().afx_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:   # interpreter shutdown: module globals may be gone already
            pass

    def set_chunk_items(self, items):
        """items per internal pass (0 = default 2^19); bounds the device workspace"""
        check(lib().afx_ctx_set_chunk_items(self.h, items))

    def set_small_batch_items(self, items):
        """passes of at most this many items take the latency plan (one chain per term); 0 = off (afx_ctx_set_small_batch_items)"""
        check(lib().afx_ctx_set_small_batch_items(self.h, items))

    def set_strict(self, enable):
        """opt-in strict mode (not the reference's behaviour): see afx_ctx_set_strict in include/aeonflux_gpu.h"""
        check(lib().afx_ctx_set_strict(self.h, 1 if enable else 0))

    def set_fixed_key_schedule(self, enable):
        """key scalars without NAF: running time independent of the issuer key (afx_ctx_set_fixed_key_schedule)"""
        check(lib().afx_ctx_set_fixed_key_schedule(self.h, 1 if enable else 0))

    def set_secret_independent_addressing(self, mode):
        """where no memory address may depend on a secret scalar's digits (afx_ctx_set_secret_independent_addressing; same bytes in every
        mode).  True / 1: everywhere (the issuer key's terms of Issuer::verify included); False / 0: nowhere (fastest); 2 or "prover": the
        prover-side calls only - issue, show, the symmetric-key helpers - which is what a new context does."""
        m = {True: 1, False: 0, "prover": 2, "all": 1, "off": 0}.get(mode, mode)
        check(lib().afx_ctx_set_secret_independent_addressing(self.h, int(m)))

    def issuer_parameters(self):
        """IssuerParameters as C_W || I (64 bytes)"""
        out = C.create_string_buffer(64)
        check(lib().afx_ctx_issuer_parameters(self.h, out))
        return out.raw

    def verify_issuances_wire(self, blob):
        """CredentialIssuance::verify over an AFXI batch; returns the status array"""
        import numpy as np
        n, kinds, nr, count, off = C.c_uint32(0), (C.c_uint8 * 32)(), C.c_uint32(0), C.c_size_t(0), C.c_size_t(0)
        check(lib().afx_issuance_wire_parse(blob, len(blob), C.byref(n), kinds, C.byref(nr), C.byref(count), C.byref(off)))
        status = np.full(max(1, count.value), 255, np.uint8)
        check(lib().afx_verify_issuances_wire(self.h, blob, len(blob), status.ctypes.data, status.size, C.byref(count)))
        return status[:count.value]

    def plan_stats(self):
        """per-item operation counts of the most recent call (afx_plan_stats) as a dict"""
        names = ("msm_jobs", "doublings", "var_additions", "fixed_additions", "table_additions", "encodings", "decodings", "keccak_permutations", "field_mul", "field_sq", "secret_terms", "chain_mul", "chain_sq")
        v = (C.c_uint64 * len(names))()
        check(lib().afx_ctx_get_plan_stats(self.h, v))
        return dict(zip(names, (int(x) for x in v)))

    def set_challenge_trace(self, rows, count):
        """parity aid: verification calls also record each recomputed challenge in a [rows][count][32] array; (0, 0) = off"""
        check(lib().afx_ctx_set_challenge_trace(self.h, rows, count))
        self._trace_shape = (rows, count)

    def get_challenge_trace(self):
        import numpy as np
        out = np.zeros(self._trace_shape + (32,), np.uint8)
        check(lib().afx_ctx_get_challenge_trace(self.h, out.ctypes.data))
        return out

    def set_pipelining(self, enable):
        """alternate successive *_dev calls between two streams (independent calls only); see afx_ctx_set_pipelining"""
        check(lib().afx_ctx_set_pipelining(self.h, 1 if enable else 0))

    def synchronize(self):
        check(lib().afx_ctx_synchronize(self.h))

    def set_timing(self, enable):
        check(lib().afx_ctx_set_timing(self.h, 1 if enable else 0))

    def get_timing(self, kernel):
        """(total_ms, launches) of one kernel since set_timing(True); synchronises the context's stream"""
        ms, n = C.c_double(0), C.c_uint64(0)
        check(lib().afx_ctx_get_timing(self.h, kernel.encode(), C.byref(ms), C.byref(n)))
        return ms.value, n.value

    def set_coalescing(self, max_wait_us=2000, max_items=4096):
        """concurrent small host-pointer calls share launch sets (on by default); max_items=0 switches it off (afx_ctx_set_coalescing)"""
        check(lib().afx_ctx_set_coalescing(self.h, max_wait_us, max_items))

    def merlin_challenges(self, label, ops, fields, count):
        """a merlin transcript over a batch (afx_merlin_challenges).  ops: ("append", label, bytes) | ("append_field", label, index) |
        ("challenge", label, n) last; fields: list of uint8 arrays [count][32].  Returns [count][64] (the first n bytes: the challenge)"""
        import numpy as np
        u32 = lambda v: int(v).to_bytes(4, "little")
        bs = lambda b: u32(len(b)) + bytes(b)
        script = bytes([1]) + bs(label)
        for op in ops:
            if op[0] == "append":
                script += bytes([2]) + bs(op[1]) + bs(op[2])
            elif op[0] == "append_field":
                script += bytes([3]) + bs(op[1]) + u32(op[2])
            elif op[0] == "challenge":
                script += bytes([4]) + bs(op[1]) + u32(op[2])
            else:
                raise ValueError(op[0])
        keep = [np.ascontiguousarray(f, dtype=np.uint8) for f in fields]
        ptrs = (C.c_void_p * max(1, len(keep)))(*[f.ctypes.data for f in keep])
        out = np.zeros((count, 64), np.uint8)
        check(lib().afx_merlin_challenges(self.h, script, len(script), ptrs, len(keep), count, out.ctypes.data))
        return out

    def set_plan_variants(self, flags):
        """force the alternatives among a small pass's equivalent plans / kernels (VARIANT_*; tests) - afx_ctx_set_plan_variants"""
        check(lib().afx_ctx_set_plan_variants(self.h, flags))

    def set_host_copy_threads(self, threads):
        """host threads that gather a large host-pointer call's rows into the pinned image; 0 = the runtime's pageable copies
        (afx_ctx_set_host_copy_threads)"""
        check(lib().afx_ctx_set_host_copy_threads(self.h, threads))

    def coalescing_stats(self):
        s = CoalescingStats()
        check(lib().afx_ctx_get_coalescing_stats(self.h, C.byref(s)))
        return {k: int(getattr(s, k)) for k, _ in CoalescingStats._fields_}

    def plan_cache_stats(self):
        s = PlanCacheStats()
        check(lib().afx_ctx_get_plan_cache_stats(self.h, C.byref(s)))
        return {k: int(getattr(s, k)) for k, _ in PlanCacheStats._fields_}

    def core_clock_samples(self):
        """the clocks (MHz, increasing) of the 64 probing blocks of the timed k_msm_window launches (afx_ctx_get_core_clock_samples)"""
        v, n = (C.c_double * 64)(), C.c_uint32(0)
        check(lib().afx_ctx_get_core_clock_samples(self.h, v, 64, C.byref(n)))
        return [float(x) for x in v[:n.value]]

    def core_clock_mhz(self):
        """core clock the timed k_msm_window launches ran at (afx_ctx_get_core_clock_mhz)"""
        mhz = C.c_double(0)
        check(lib().afx_ctx_get_core_clock_mhz(self.h, C.byref(mhz)))
        return mhz.value

    # ---- Issuer::verify ----
    def verify_presentations(self, shape, soa, count, status_ptr, device_pointers=False):
        fn = lib().afx_verify_presentations_dev if device_pointers else lib().afx_verify_presentations
        check(fn(self.h, C.byref(shape), C.byref(soa), count, status_ptr))

    def verify_encryption_proofs(self, index, soa, count, status_ptr, device_pointers=False):
        fn = lib().afx_verify_encryption_proofs_dev if device_pointers else lib().afx_verify_encryption_proofs
        check(fn(self.h, index, C.byref(soa), count, status_ptr))


class Group:
    """The issuer on several GPUs of one node (afx_group_*): one context per listed device, batches split contiguously,
    one host thread per member, no collective.  The same device may be listed more than once."""

    def __init__(self, sysparams, amacs_key, issuer_params, devices):
        h = C.c_void_p()
        devs = (C.c_int * len(devices))(*devices)
        check(lib().afx_group_create(C.byref(h), devs, len(devices), sysparams, len(sysparams), amacs_key, len(amacs_key) if amacs_key else 0,
                                     issuer_params))
        self.h = h
        self.devices = list(devices)
        self._members = []   # weak references to the borrowed contexts handed out: invalidated by close()
        self.n = self.member(0).n

    def __len__(self):
        return lib().afx_group_size(self.h)

    def member(self, i):
        m = lib().afx_group_member(self.h, i)
        if not m:
            raise IndexError(i)
        import weakref
        c = Context(None, None, None, device=self.devices[i], _borrowed=C.c_void_p(m), _group=self)   # keeps the group alive
        self._members.append(weakref.ref(c))
        return c

    def close(self):
        if getattr(self, "h", None):
            for r in self._members:   # a member handle must not outlive the contexts the group owns
                c = r()
                if c is not None:
                    c.h = None
            self._members = []
            lib().afx_group_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def shard_bounds(count, members, index):
    """(first, n) of member `index`: the library's contiguous split (afx_shard_bounds)"""
    first, n = C.c_size_t(0), C.c_size_t(0)
    lib().afx_shard_bounds(count, members, index, C.byref(first), C.byref(n))
    return first.value, n.value
