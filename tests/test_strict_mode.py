"""Strict mode (opt-in, NOT the reference's behaviour): constraint #3 by own position, exactly one
proof of encryption per hidden group element, and the DLEQ tying each hidden group element's commitment C_y[i] to the C_y_1
of its proof of encryption (the TODO of the reference's README.md:121-122).  The oracle carries the same switch (oracle/aeonflux.c,
afxo_ctx_set_strict), so the GPU path is compared with it byte for byte; the default mode must be unaffected."""
import numpy as np
import pytest

from tests.helpers import gpu_verify, make_credentials


def _show_all(user, d, kinds, count, take):
    nsp = sum(1 for k in kinds if k == 4)
    kps = [user.keypair_derive(take(64)) for _ in range(count)]
    zw, sd, es = [take(64) for _ in range(count)], [take(32) for _ in range(count)], [take(32 * nsp) for _ in range(count)]
    pres = []
    for c, kp, z, s, e in zip(d["creds"], kps, zw, sd, es):
        st, p = user.show(kinds, c["values"], c["t"], c["U"], c["V"], kp, z, s, e)
        assert st == 0
        pres.append(p)
    return pres, (kps, zw, sd, es)


def test_oracle_strict_mode_semantics():
    """CPU: what strict mode means, on the oracle alone"""
    import oracle
    # leading hidden group element: the reference's statement rejects its own honest proofs (App. B); strict accepts
    d = make_credentials(3, "ESS", 3, b"strict-oracle")
    user, issuer = d["user"], d["issuer"]
    kinds = [4, 0, 0]
    default_pres, rnd = _show_all(user, d, kinds, 3, d["take"])
    assert [issuer.verify_presentation(p) for p in default_pres] == [1, 1, 1]
    user.set_strict(True)
    issuer.set_strict(True)
    strict_pres = []
    for c, kp, z, s, e in zip(d["creds"], *rnd):
        st, p = user.show(kinds, c["values"], c["t"], c["U"], c["V"], kp, z, s, e)
        assert st == 0
        strict_pres.append(p)
    assert [issuer.verify_presentation(p) for p in strict_pres] == [0, 0, 0]
    assert [issuer.verify_presentation(p) for p in default_pres] == [1, 1, 1]
    # a presentation stripped of its proof of encryption: accepted by the reference (presentation.rs:438-440), not by strict
    q = oracle.Presentation.from_buffer_copy(bytes(strict_pres[0]))
    q.n_enc_proofs = 0
    assert issuer.verify_presentation(q) == 1
    issuer.set_strict(False)
    user.set_strict(False)
    assert [issuer.verify_presentation(p) for p in strict_pres] == [1, 1, 1]
    # trailing hidden group element: constraint #3 coincides in both modes, so the commitments are the same bytes; the proof
    # differs because the strict statement also carries the DLEQ (one more point variable and constraint in the transcript)
    d2 = make_credentials(4, "SSPE", 2, b"strict-oracle-2")
    k2 = [1, 0, 2, 4]
    a, rnd2 = _show_all(d2["user"], d2, k2, 2, d2["take"])
    d2["user"].set_strict(True)
    strict2 = []
    for p, (c, kp, z, s, e) in zip(a, zip(d2["creds"], *rnd2)):
        st, ps = d2["user"].show(k2, c["values"], c["t"], c["U"], c["V"], kp, z, s, e)
        assert st == 0 and bytes(ps.C_x_0) == bytes(p.C_x_0) and bytes(ps.C_y[3]) == bytes(p.C_y[3]) and bytes(ps.enc[0].C_y_1) == bytes(p.enc[0].C_y_1)
        assert bytes(ps.challenge) != bytes(p.challenge)
        strict2.append(ps)
    q = oracle.Presentation.from_buffer_copy(bytes(a[0]))
    q.n_enc_proofs = 0
    assert d2["issuer"].verify_presentation(q) == 0    # the reference's behaviour: nothing ties the count to the hidden points
    # what the DLEQ is for.  Take presentation 0 and attach presentation 1's (perfectly valid) proof of encryption: the
    # reference accepts the pair although the ciphertext is about another plaintext; strict mode rejects it.
    swapped = oracle.Presentation.from_buffer_copy(bytes(a[0]))
    swapped.enc[0] = oracle.EncProof.from_buffer_copy(bytes(a[1].enc[0]))
    assert d2["issuer"].verify_presentation(swapped) == 0
    d2["issuer"].set_strict(True)
    assert [d2["issuer"].verify_presentation(p) for p in strict2] == [0, 0]
    swapped = oracle.Presentation.from_buffer_copy(bytes(strict2[0]))
    swapped.enc[0] = oracle.EncProof.from_buffer_copy(bytes(strict2[1].enc[0]))
    assert d2["issuer"].verify_presentation(swapped) == 1
    assert [d2["issuer"].verify_presentation(p) for p in a] == [1, 1]     # default-mode proofs carry no DLEQ
    d2["issuer"].set_strict(False)
    assert [d2["issuer"].verify_presentation(p) for p in strict2] == [1, 1]


@pytest.mark.gpu
@pytest.mark.parametrize("n,layout,hide", [(3, "ESS", [0]), (6, "SESPSE", [0, 1, 4, 5]), (4, "SSPE", [0, 3])])
def test_gpu_strict_mode_matches_strict_oracle(n, layout, hide):
    import oracle
    import aeonflux_amd as afx
    from tests.test_gpu_prove import gpu_show
    count = 12
    d = make_credentials(n, layout, count, b"strict-gpu-%d" % n)
    user, issuer = d["user"], d["issuer"]
    kinds = list(d["creds"][0]["kinds"])
    for i in hide:
        kinds[i] = 1 if kinds[i] == 0 else 4
    nsp = sum(1 for k in kinds if k == 4)
    user.set_strict(True)
    issuer.set_strict(True)
    want, (kps, zw, sd, es) = _show_all(user, d, kinds, count, d["take"])
    uctx = afx.Context(d["params"], None, d["ip"])
    uctx.set_strict(True)
    o, shape, status = gpu_show(afx, uctx, kinds, d["creds"], kps, zw, sd, es)
    uctx.close()
    assert status.tolist() == [0] * count
    for i, p in enumerate(want):
        assert bytes(o["challenge"][32 * i:32 * i + 32]) == bytes(p.challenge)
        for k in range(p.n_responses):
            assert bytes(o["responses"][32 * (k * count + i):32 * (k * count + i) + 32]) == bytes(p.responses[k])
        for k in range(n):
            assert bytes(o["C_y"][32 * (k * count + i):32 * (k * count + i) + 32]) == bytes(p.C_y[k])
    # verification: tamper a few, strip the proofs of encryption from one
    want[1].responses[0][3] ^= 1
    want[2].C_V[0] ^= 4
    ictx = afx.Context(d["params"], d["key"], d["ip"])
    ictx.set_strict(True)
    assert gpu_verify(afx, ictx, want) == [issuer.verify_presentation(p) for p in want] == [0, 1, 1] + [0] * (count - 3)
    if nsp:
        stripped = [oracle.Presentation.from_buffer_copy(bytes(p)) for p in want[3:6]]
        for q in stripped:
            q.n_enc_proofs = nsp - 1
        assert gpu_verify(afx, ictx, stripped) == [issuer.verify_presentation(q) for q in stripped] == [1, 1, 1]
        # the DLEQ: another presentation's valid proof of encryption in place of the own one is rejected
        swapped = [oracle.Presentation.from_buffer_copy(bytes(p)) for p in want[6:9]]
        for q, donor in zip(swapped, want[7:10]):
            q.enc[0] = oracle.EncProof.from_buffer_copy(bytes(donor.enc[0]))
        assert gpu_verify(afx, ictx, swapped) == [issuer.verify_presentation(q) for q in swapped] == [1, 1, 1]
    # the same context back in the reference's mode gives the reference's answers again
    ictx.set_strict(False)
    issuer.set_strict(False)
    ref = [issuer.verify_presentation(p) for p in want]
    assert gpu_verify(afx, ictx, want) == ref
    assert ref == [1] * count                            # strict-mode proofs do not verify under the reference's statement
    ictx.close()
