"""GPU: seeded random statement shapes against the oracle - attribute counts 1..12, random kinds and hidden sets (hidden group
elements kept trailing so that honest proofs verify under the reference's constraint #3, App. B; a few leading ones on purpose),
every kind of corruption.  For each shape: show on the GPU must give the oracle's bytes, verification the oracle's statuses,
and the recomputed challenges (a hash over every recomputed commitment) the oracle's - in the reference's mode and in strict mode.
This is the net under the engine's structural optimisations (shared window tables, commitments encoded as doubles by
k_compress2x, per-class kernels): they depend on the shape of the launch list, which fixed-shape tests cannot vary."""
import os
import random

import numpy as np
import pytest

from tests.helpers import corrupt, gpu_verify, make_credentials

pytestmark = pytest.mark.gpu


def _random_shape(rng):
    n = rng.randint(1, 12)
    n_e = rng.randint(0, min(3, n))                     # plaintext attributes, trailing
    layout = "".join(rng.choice("SP") for _ in range(n - n_e)) + "E" * n_e
    if n >= 3 and rng.random() < 0.15:                   # a leading hidden group element now and then (honest proofs then fail, App. B)
        layout = "E" + layout[1:]
    hide = [i for i, c in enumerate(layout) if (c == "S" and rng.random() < 0.4) or (c == "E" and rng.random() < 0.8)]
    return n, layout, hide


# AFX_FUZZ_SEEDS=n widens the sweep for soak runs (the suite runs the first 12)
@pytest.mark.parametrize("seed", range(int(os.environ.get("AFX_FUZZ_SEEDS", "12"))))
def test_random_shapes_show_and_verify_like_the_oracle(seed):
    import oracle
    import aeonflux_amd as afx
    from tests.test_gpu_prove import gpu_show
    rng = random.Random(1000 + seed)
    for case in range(4):
        n, layout, hide = _random_shape(rng)
        strict = rng.random() < 0.3
        count = rng.choice((1, 7, 33, 65))
        secret = random.Random(7700 + 4 * seed + case).random() < 0.4   # secret-independent addressing on both sides (its own stream:
                                                                         # the shapes of earlier rounds' sweeps stay what they were)
        tag = b"fuzz-%d-%d" % (seed, case)
        d = make_credentials(n, layout, count, tag)
        user, issuer, take = d["user"], d["issuer"], d["take"]
        user.set_strict(strict)
        issuer.set_strict(strict)
        kinds = list(d["creds"][0]["kinds"])
        for i in hide:
            kinds[i] = 1 if kinds[i] == 0 else 4
        nsp = sum(1 for k in kinds if k == 4)
        kps = [user.keypair_derive(take(64)) for _ in range(count)]
        zw, sd, es = [take(64) for _ in range(count)], [take(32) for _ in range(count)], [take(32 * nsp) for _ in range(count)]
        pres = []
        for c, kp, z, s, e in zip(d["creds"], kps, zw, sd, es):
            st, p = user.show(kinds, c["values"], c["t"], c["U"], c["V"], kp, z, s, e)
            assert st == 0, (layout, hide)
            pres.append(p)
        # AnonymousCredential::show on the GPU: the oracle's bytes
        uctx = afx.Context(d["params"], None, d["ip"])
        latency_plan = random.Random(8800 + 4 * seed + case).random() < 0.5   # one chain per term (the default for small passes) or one per job
        uctx.set_strict(strict)
        uctx.set_small_batch_items(2048 if latency_plan else 0)
        uctx.set_secret_independent_addressing(secret)
        o, shape, status = gpu_show(afx, uctx, kinds, d["creds"], kps, zw, sd, es)
        uctx.close()
        assert status.tolist() == [0] * count, (layout, hide)
        for i, p in enumerate(pres):
            assert bytes(o["challenge"][32 * i:32 * i + 32]) == bytes(p.challenge), (layout, hide, strict, i)
            for k in range(p.n_responses):
                assert bytes(o["responses"][32 * (k * count + i):32 * (k * count + i) + 32]) == bytes(p.responses[k])
            for k in range(n):
                assert bytes(o["C_y"][32 * (k * count + i):32 * (k * count + i) + 32]) == bytes(p.C_y[k])
            for f in ("C_x_0", "C_x_1", "C_V"):
                assert bytes(o[f][32 * i:32 * i + 32]) == bytes(getattr(p, f))
            for e in range(nsp):
                q = p.enc[e]
                assert bytes(o["enc"][e]["challenge"][32 * i:32 * i + 32]) == bytes(q.challenge)
                for f in ("E1", "E2", "C_y_1", "C_y_2", "C_y_3", "C_y_2p"):
                    assert bytes(o["enc"][e][f][32 * i:32 * i + 32]) == bytes(getattr(q, f)), f
        # Issuer::verify on corrupted copies: the oracle's statuses and recomputed challenges
        corrupt(pres, tag + b"-c")
        want = [issuer.verify_presentation(p) for p in pres]
        want_ch = [[None] * count for _ in range(1 + nsp)]
        for i, p in enumerate(pres):
            q = oracle.Presentation.from_buffer_copy(bytes(p))
            q.n_enc_proofs = 0
            if strict:
                continue        # the strict statement needs the proofs of encryption attached: statuses only
            oracle.debug_reset()
            issuer.verify_presentation(q)
            commits, ch = oracle.debug_last()
            if commits:
                want_ch[0][i] = ch
            for e in range(nsp):
                oracle.debug_reset()
                issuer.verify_encryption_proof(p.enc[e])
                commits, ch = oracle.debug_last()
                if commits:
                    want_ch[1 + e][i] = ch
        ictx = afx.Context(d["params"], d["key"], d["ip"])
        ictx.set_strict(strict)
        ictx.set_small_batch_items(2048 if latency_plan else 0)
        ictx.set_secret_independent_addressing(secret)
        ictx.set_challenge_trace(1 + nsp, count)
        got = gpu_verify(afx, ictx, pres)
        tr = ictx.get_challenge_trace()
        ictx.close()
        assert got == want, (layout, hide, strict, got, want)
        for r in range(1 + nsp):
            for i in range(count):
                if want_ch[r][i] is not None:
                    assert bytes(tr[r, i]) == want_ch[r][i], (layout, hide, r, i)
