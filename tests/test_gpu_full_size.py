"""GPU, BASELINE.json's full sizes: size-independent properties of the whole pipeline.
Every credential issued on the GPU, shown on the GPU, must verify on the GPU (issue -> show -> verify round trip), and
exactly the items corrupted afterwards must be rejected; a sample of the same batch goes through the oracle."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

SAMPLE = 1 << 16   # "every recomputed commitment + challenge compared ... on a 2^16 sample for C3-C5"


def _strided(count, offset):
    """SAMPLE item indices spread over the whole batch (every count/SAMPLE-th item from `offset`), or all of a smaller batch"""
    if count <= SAMPLE:
        return np.arange(count)
    return (np.arange(SAMPLE, dtype=np.int64) * (count // SAMPLE) + offset) % count


def _oracle_agrees_on_sample(oracle, octx, shape, pres, idx, gpu_status, gpu_trace):
    """The items `idx` of the batch through the oracle (all host cores): its status and, for every proof whose verifier reaches
    its commitments, the challenge it recomputes - a hash over every recomputed commitment of that proof - must equal what the
    GPU answered / recorded for the same items.  gpu_status [len(idx)], gpu_trace [1 + n_enc_proofs, len(idx), 32]."""
    from aeonflux_amd import batch
    sub = {f: np.ascontiguousarray(pres[f][..., idx, :]) for f in batch.PRES_FIELDS}
    sub["enc"] = [{f: np.ascontiguousarray(d[f][..., idx, :]) for f in batch.ENC_FIELDS} for d in pres["enc"]]
    soa, keep = batch.presentation_soa(sub)
    st, trace, reached = oracle.verify_presentations_traced(octx, shape, soa, len(idx))
    assert np.array_equal(st, gpu_status), np.nonzero(st != gpu_status)[0][:8]
    r = reached.astype(bool)
    assert np.array_equal(trace[r], gpu_trace[r]), "a recomputed challenge differs from the oracle's"
    return int(r.sum()), int(st.sum())


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
    # the oracle on a strided 2^16-item sample (C2: the whole batch): statuses and every recomputed challenge
    issuer.set_challenge_trace(1 + shape.n_enc_proofs, count)
    assert np.array_equal(batch.verify_presentations(issuer, shape, pres), want)
    trace = issuer.get_challenge_trace()
    issuer.set_challenge_trace(0, 0)
    idx = _strided(count, 5)
    octx = oracle.Ctx(params, key, ip)
    n_reached, n_rejected = _oracle_agrees_on_sample(oracle, octx, shape, pres, idx, got[idx], trace[:, idx])
    assert n_reached > (1 + shape.n_enc_proofs) * len(idx) * 0.99 and n_rejected >= len(idx) // 200
    del trace
    # verification is a function of the item alone: a permuted batch gives the permuted statuses
    perm = np.random.default_rng(5).permutation(count)
    shuffled = {f: np.ascontiguousarray(pres[f][..., perm, :]) for f in batch.PRES_FIELDS}
    shuffled["enc"] = [{f: np.ascontiguousarray(d[f][..., perm, :]) for f in batch.ENC_FIELDS} for d in pres["enc"]]
    assert np.array_equal(batch.verify_presentations(issuer, shape, shuffled), want[perm])
    issuer.close()
    # ... and item by item through the oracle's single-presentation entry point (first items + every corrupted item among the first 4096)
    sample = sorted(set(range(64)) | set(int(i) for i in np.nonzero(want[:4096])[0]))
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


@pytest.mark.parametrize("count", [1 << 12, 1 << 14, 1 << 17])   # the latency plan, the key-job split, the plan of large passes
def test_secret_independent_addressing_gives_the_same_bytes_at_size(count):
    """afx_ctx_set_secret_independent_addressing changes HOW tables are read (2-bit windows over two-entry tables read whole, the
    generators' 4-bit tables read whole through scalar registers), never a byte of a result: the C3 pipeline - issue, show with
    four proofs of encryption, verify with 1 % corrupted - with the mode on, on every context, against the same run with it off."""
    import aeonflux_amd as afx
    import bench
    from aeonflux_amd import batch
    n, layout, hide = 8, "SSPPEEEE", [4, 5, 6, 7]
    params, key, ip = bench.load_fixture("c3_8attrs_SSPPeeee")
    runs = []
    for secret in (False, True):
        issuer, user = afx.Context(params, key, ip), afx.Context(params, None, ip)
        issuer.set_secret_independent_addressing(secret)
        user.set_secret_independent_addressing(secret)
        pres, shape = bench.generate(afx, batch, issuer, user, params, n, layout, hide, count, 4242)
        assert (user.plan_stats()["secret_terms"] > 0) == secret   # the last call on `user`: show
        want = bench.corrupt(pres, count, 17)
        issuer.set_challenge_trace(1 + shape.n_enc_proofs, count)
        got = batch.verify_presentations(issuer, shape, pres)
        assert (issuer.plan_stats()["secret_terms"] > 0) == secret   # the key's terms of Z
        trace = issuer.get_challenge_trace().copy()
        issuer.set_challenge_trace(0, 0)
        assert np.array_equal(got, want)
        runs.append((pres, got, trace))
        issuer.close(); user.close()
    (p0, g0, t0), (p1, g1, t1) = runs
    for f in batch.PRES_FIELDS:
        assert np.array_equal(p0[f], p1[f]), f
    for e0, e1 in zip(p0["enc"], p1["enc"]):
        for f in batch.ENC_FIELDS:
            assert np.array_equal(e0[f], e1[f]), f
    assert np.array_equal(g0, g1) and np.array_equal(t0, t1)


def _tile(a, reps):
    return np.ascontiguousarray(np.concatenate([a] * reps, axis=-2))


def test_c4_shards_of_a_2_22_batch_equal_the_unsharded_answer():
    """BASELINE config 4: 2^22 presentations (C3 statement) host-sharded over 8 GPUs.  One GPU here: the shards of ranks 0
    and 7 (afx_shard_bounds(2^22, 8, r): 2^19 items each) go through a second context as ranges of the whole batch, and must
    equal both the expected statuses and the first context's unsharded answer on the same items.  The 2^22 presentations
    are four copies of 2^20 distinct ones, corrupted independently after the copy (1 % of all items)."""
    import aeonflux_amd as afx
    import bench
    from aeonflux_amd import batch
    n, layout, hide, fixture = 8, "SSPPEEEE", [4, 5, 6, 7], "c3_8attrs_SSPPeeee"
    params, key, ip = bench.load_fixture(fixture)
    issuer = afx.Context(params, key, ip)
    user = afx.Context(params, None, ip)
    base, total, chunk = 1 << 20, 1 << 22, 1 << 16
    parts = [bench.generate(afx, batch, issuer, user, params, n, layout, hide, chunk, 9000 + o) for o in range(0, base, chunk)]
    shape = parts[0][1]
    pres = {f: _tile(np.concatenate([p[0][f] for p in parts], axis=-2), total // base) for f in batch.PRES_FIELDS}
    pres["enc"] = [{f: _tile(np.concatenate([p[0]["enc"][e][f] for p in parts], axis=-2), total // base) for f in batch.ENC_FIELDS}
                   for e in range(shape.n_enc_proofs)]
    del parts
    user.close()
    want = bench.corrupt(pres, total, 4)
    assert want.sum() == total // 100
    whole = batch.verify_presentations(issuer, shape, pres)          # unsharded, 2^22 items through one context
    assert np.array_equal(whole, want)
    import oracle
    second = afx.Context(params, key, ip)
    for r in (0, 7):
        first, cnt = afx.shard_bounds(total, 8, r)
        assert cnt == 1 << 19 and first == r << 19
        if r == 7:
            second.set_challenge_trace(1 + shape.n_enc_proofs, cnt)   # indexed by the item's position inside the range call
        part = batch.verify_presentations(second, shape, pres, first=first, n=cnt)
        assert np.array_equal(part[first:first + cnt], whole[first:first + cnt])
        assert (part[:first] == 255).all() and (part[first + cnt:] == 255).all()
    # rank 7's shard: a strided 2^16-item sample through the oracle, statuses and recomputed challenges
    trace = second.get_challenge_trace()
    second.set_challenge_trace(0, 0)
    local = _strided(cnt, 3)
    octx = oracle.Ctx(params, key, ip)
    n_reached, n_rejected = _oracle_agrees_on_sample(oracle, octx, shape, pres, first + local, part[first + local], trace[:, local])
    assert n_reached > 5 * len(local) * 0.99 and n_rejected >= len(local) // 200
    del trace
    second.close()
    issuer.close()


def test_c5_issue_2_20_credentials_16_attributes():
    """BASELINE config 5 at its full size on one GPU: 2^20 issuances, 16 attributes (S x8, P x4, E x4).  Every issuance verifies
    on the user side (CredentialIssuance::verify), exactly the corrupted 1 % are rejected, and the first 64 are the oracle's bytes."""
    import oracle
    import aeonflux_amd as afx
    import bench
    from aeonflux_amd import batch
    n, layout, count = 16, "SSSSSSSSPPPPEEEE", 1 << 20
    params, key, ip = bench.load_fixture("c5_16attrs")
    issuer = afx.Context(params, key, ip)
    user = afx.Context(params, None, ip)
    rng = np.random.default_rng(2020)
    rb = lambda *s: rng.integers(0, 256, size=s, dtype=np.uint8)
    kinds = [{"S": afx.ATTR_PUBLIC_SCALAR, "P": afx.ATTR_PUBLIC_POINT, "E": afx.ATTR_EITHER_POINT}[c] for c in layout]
    values = np.zeros((n, count, 32), np.uint8)
    for i, c in enumerate(layout):
        for o in range(0, count, 1 << 18):
            w = rb(1 << 18, 64)
            values[i, o:o + (1 << 18)] = batch.scalars_from_wide(issuer, w) if c == "S" else batch.points_from_uniform(issuer, w)
    tw, uw, seed = rb(count, 64), rb(count, 64), rb(count, 32)
    iss, st = batch.issue(issuer, kinds, values, tw, uw, seed)
    assert not st.any()
    assert not batch.verify_issuances(user, kinds, values, iss).any()
    octx = oracle.Ctx(params, key, ip)
    for i in range(64):
        vals = [bytes(values[k, i]) + bytes(64) for k in range(n)]
        s, t, U, V, ch, resp = octx.issue(kinds, vals, bytes(tw[i]), bytes(uw[i]), bytes(seed[i]))
        assert s == 0 and t == bytes(iss["t"][i]) and U == bytes(iss["U"][i]) and V == bytes(iss["V"][i]) and ch == bytes(iss["challenge"][i])
        assert all(resp[k] == bytes(iss["responses"][k, i]) for k in range(n + 5))
    # a strided 2^16-item sample re-issued by the oracle (all host cores): every output byte
    sidx = _strided(count, 11)
    oi, ost = oracle.issue_soa(octx, kinds, np.ascontiguousarray(values[:, sidx]), tw[sidx], uw[sidx], seed[sidx])
    assert not ost.any()
    for f in ("t", "U", "V", "challenge", "responses"):
        assert np.array_equal(oi[f], iss[f][..., sidx, :]), f
    # 1 % corrupted: a response bit, the tag's V, an attribute value, the identity as U
    want = np.zeros(count, np.uint8)
    idx = np.random.default_rng(3).choice(count, size=count // 100, replace=False)
    for j, i in enumerate(idx):
        mode = j % 4
        if mode == 0:
            iss["responses"][j % (n + 5), i, 2] ^= 0x20
        elif mode == 1:
            iss["V"][i, 9] ^= 0x01
        elif mode == 2:
            values[j % 8, i, 0] ^= 0x01       # a scalar attribute changed after issuance
        else:
            iss["U"][i, :] = 0
        want[i] = 1
    user.set_challenge_trace(1, count)
    got = batch.verify_issuances(user, kinds, values, iss)
    assert np.array_equal(got, want)
    trace = user.get_challenge_trace()[0]
    user.set_challenge_trace(0, 0)
    # the same strided sample through the oracle's CredentialIssuance::verify: statuses and recomputed challenges
    uctx = oracle.Ctx(params, None, ip)
    sub = {f: np.ascontiguousarray(iss[f][..., sidx, :]) for f in ("t", "U", "V", "challenge", "responses")}
    vst, vtrace, vreached = oracle.verify_issuances_traced(uctx, kinds, np.ascontiguousarray(values[:, sidx]), sub)
    assert len(sidx) == SAMPLE and np.array_equal(vst, got[sidx]) and vst.sum() >= len(sidx) // 200
    r = vreached.astype(bool)
    assert r.sum() > 0.99 * len(sidx) and np.array_equal(vtrace[r], trace[sidx][r])
    # a sample of the rejected ones through the oracle's single-issuance entry point
    for i in sorted(int(x) for x in idx[:24]):
        vals = [bytes(values[k, i]) + bytes(64) for k in range(n)]
        assert uctx.issuance_verify(kinds, vals, bytes(iss["t"][i]), bytes(iss["U"][i]), bytes(iss["V"][i]), bytes(iss["challenge"][i]),
                                    [bytes(iss["responses"][k, i]) for k in range(n + 5)]) == 1
    user.close()
    issuer.close()


def test_large_host_calls_with_and_without_the_copy_pool():
    """Host-pointer calls of more than 16 MB of rows: the runtime's pageable copies (the default) and the context's copy pool
    (afx_ctx_set_host_copy_threads: rows gathered into the lane's pinned image by host threads on the device's NUMA node, one transfer per
    contiguous run) stage the same bytes - on column arrays, on a sub-range and on a serialized batch (2^17 + 1 C2 presentations = 119 MB:
    a first slice of 2^16 items and a whole pass, statements.hpp host_slice_items), statuses of a 1 % corrupted batch equal either way
    and equal to the expected ones; issue through the pool returns the bytes it returns without (100 MB of results scattered by it)."""
    import aeonflux_amd as afx
    import bench
    from aeonflux_amd import batch, wire
    n, layout, hide, count = 4, "SSPE", [0, 3], (1 << 17) + 1
    params, key, ip = bench.load_fixture("readme_4attrs_sSPe")
    issuer = afx.Context(params, key, ip)
    user = afx.Context(params, None, ip)
    parts = [bench.generate(afx, batch, issuer, user, params, n, layout, hide, min(1 << 16, count - o), 77 + o, fast_tables=True) for o in range(0, count, 1 << 16)]
    shape = parts[0][1]
    pres = {f: np.concatenate([p[0][f] for p in parts], axis=-2) for f in batch.PRES_FIELDS}
    pres["enc"] = [{f: np.concatenate([p[0]["enc"][e][f] for p in parts], axis=-2) for f in batch.ENC_FIELDS} for e in range(shape.n_enc_proofs)]
    user.close()
    want = bench.corrupt(pres, count, 5)
    blob = wire.pack_presentations(shape, pres)
    soa, keep = batch.presentation_soa(pres)
    for threads in (0, 4, 1):
        issuer.set_host_copy_threads(threads)
        assert np.array_equal(batch.verify_presentations(issuer, shape, pres), want), threads
        st = np.full(count, 0xEE, np.uint8)
        afx.check(afx.lib().afx_verify_presentations_range(issuer.h, C.byref(shape), C.byref(soa), count, 1000, 100000, st.ctypes.data))
        assert np.array_equal(st[1000:101000], want[1000:101000]) and (st[:1000] == 0xEE).all() and (st[101000:] == 0xEE).all(), threads
        assert np.array_equal(wire.verify_wire(issuer, blob), want), threads
    rng = np.random.default_rng(3)
    rb = lambda *s: rng.integers(0, 256, size=s, dtype=np.uint8)
    kinds = [afx.ATTR_PUBLIC_SCALAR, afx.ATTR_PUBLIC_SCALAR, afx.ATTR_PUBLIC_POINT, afx.ATTR_PUBLIC_POINT]
    values = np.stack([batch.scalars_from_wide(issuer, rb(count, 64)), batch.scalars_from_wide(issuer, rb(count, 64)),
                       batch.points_from_uniform(issuer, rb(count, 64)), batch.points_from_uniform(issuer, rb(count, 64))])
    tw, uw, sd = rb(count, 64), rb(count, 64), rb(count, 32)
    outs = []
    for threads in (0, 4):
        issuer.set_host_copy_threads(threads)
        o, st = batch.issue(issuer, kinds, values, tw, uw, sd)
        assert not st.any()
        outs.append(o)
    for f in ("t", "U", "V", "challenge", "responses"):
        assert np.array_equal(outs[0][f], outs[1][f]), f
    issuer.close()
