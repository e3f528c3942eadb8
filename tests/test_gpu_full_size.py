"""GPU, BASELINE.json's full sizes: size-independent properties of the whole pipeline.
Every credential issued on the GPU, shown on the GPU, must verify on the GPU (issue -> show -> verify round trip), and
exactly the items corrupted afterwards must be rejected; a sample of the same batch goes through the oracle."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name,n,layout,hide,count,fixture", [
    ("C2", 4, "SSPE", [0, 3], 1 << 16, "readme_4attrs_sSPe"),
    ("C3", 8, "SSPPEEEE", [4, 5, 6, 7], 1 << 20, "c3_8attrs_SSPPeeee"),
])
def test_issue_show_verify_round_trip_at_full_size(name, n, layout, hide, count, fixture):
    import oracle
    import aeonflux_amd as afx
    import bench
    from aeonflux_amd import batch
    params, key, ip = bench.load_fixture(fixture)
    issuer = afx.Context(params, key, ip)
    user = afx.Context(params, None, ip)
    chunk = 1 << 16
    parts = [bench.generate(afx, batch, issuer, user, params, n, layout, hide, min(chunk, count - o), 555 + o) for o in range(0, count, chunk)]
    shape = parts[0][1]
    pres = {f: np.concatenate([p[0][f] for p in parts], axis=-2) for f in batch.PRES_FIELDS}
    pres["enc"] = [{f: np.concatenate([p[0]["enc"][e][f] for p in parts], axis=-2) for f in batch.ENC_FIELDS} for e in range(shape.n_enc_proofs)]
    del parts
    user.close()
    # round trip: everything honest verifies
    assert not batch.verify_presentations(issuer, shape, pres).any()
    # exactly the corrupted items are rejected
    want = bench.corrupt(pres, count, 31)
    got = batch.verify_presentations(issuer, shape, pres)
    assert np.array_equal(got, want) and want.sum() == count // 100
    # verification is a function of the item alone: a permuted batch gives the permuted statuses
    perm = np.random.default_rng(5).permutation(count)
    shuffled = {f: np.ascontiguousarray(pres[f][..., perm, :]) for f in batch.PRES_FIELDS}
    shuffled["enc"] = [{f: np.ascontiguousarray(d[f][..., perm, :]) for f in batch.ENC_FIELDS} for d in pres["enc"]]
    assert np.array_equal(batch.verify_presentations(issuer, shape, shuffled), want[perm])
    issuer.close()
    # a sample through the oracle (first items + every corrupted item among the first 4096)
    sample = sorted(set(range(64)) | set(int(i) for i in np.nonzero(want[:4096])[0]))
    octx = oracle.Ctx(params, key, ip)
    for i in sample:
        p = oracle.Presentation()
        p.n_attributes, p.n_responses, p.n_hidden_scalars, p.n_enc_proofs = shape.n_attributes, shape.n_responses, shape.n_hidden_scalars, shape.n_enc_proofs
        for k in range(n):
            p.kinds[k] = shape.kinds[k]
            C.memmove(p.C_y[k], pres["C_y"][k, i].tobytes(), 32)
            C.memmove(p.attr_values[k], pres["attr_values"][k, i].tobytes(), 32)
        for k in range(shape.n_hidden_scalars):
            p.hidden_scalar_indices[k] = shape.hidden_scalar_indices[k]
        C.memmove(p.challenge, pres["challenge"][i].tobytes(), 32)
        for k in range(shape.n_responses):
            C.memmove(p.responses[k], pres["responses"][k, i].tobytes(), 32)
        for f in ("C_x_0", "C_x_1", "C_V"):
            C.memmove(getattr(p, f), pres[f][i].tobytes(), 32)
        for e in range(shape.n_enc_proofs):
            q, d = p.enc[e], pres["enc"][e]
            q.index = shape.enc_indices[e]
            C.memmove(q.challenge, d["challenge"][i].tobytes(), 32)
            for k in range(6):
                C.memmove(q.responses[k], d["responses"][k, i].tobytes(), 32)
            for f in ("pk", "E1", "E2", "C_y_1", "C_y_2", "C_y_3", "C_y_2p"):
                C.memmove(getattr(q, f), d[f][i].tobytes(), 32)
        assert octx.verify_presentation(p) == int(want[i]), (name, i)
