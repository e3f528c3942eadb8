"""GPU parity for Issuer::verify: the HIP path through the C ABI must give
the ORACLE's accept/reject for every item — committed flows, seeded batches with every kind of corruption,
ragged batch sizes, and the shapes on which the reference panics."""
import hashlib

import numpy as np
import pytest

from tests.helpers import corrupt, gpu_verify, make_batch, pres_from_json

pytestmark = pytest.mark.gpu
H = bytes.fromhex


def test_committed_flows(flows):
    import aeonflux_amd as afx
    n_checked = 0
    for r in flows:
        if "presentation" not in r:
            continue
        ctx = afx.Context(H(r["params"]), H(r["key"]), H(r["issuer_params"]))
        p = pres_from_json(r)
        assert gpu_verify(afx, ctx, [p]) == [r["verify"]], r["name"]
        ctx.close()
        n_checked += 1
    assert n_checked >= 12


@pytest.mark.parametrize("n,layout,hide,count", [
    (4, "SSPE", [0, 3], 200),           # BASELINE config 1/2 shape  s S P e
    (8, "SSPPEEEE", [4, 5, 6, 7], 70),  # BASELINE config 3/4 shape  S S P P e e e e
    (1, "S", [], 65),
    (3, "ESS", [0], 33),                # leading hidden point: honest proofs are rejected (App. B)
    (6, "SSSPSE", [0, 2, 4, 5], 64),
    (2, "SP", [], 1),
])
def test_batches_with_corruptions(n, layout, hide, count):
    import aeonflux_amd as afx
    params, key, ip, issuer, pres = make_batch(n, layout, hide, count, b"gpu-verify-%d-%s" % (n, layout.encode()))
    corrupt(pres, b"corrupt-" + layout.encode())
    want = [issuer.verify_presentation(p) for p in pres]
    ctx = afx.Context(params, key, ip)
    got = gpu_verify(afx, ctx, pres)
    ctx.close()
    assert got == want
    if count > 30 and layout != "ESS":
        assert 0 in want and 1 in want


def test_shapes_the_reference_panics_on():
    import oracle
    import aeonflux_amd as afx
    from tests.soa import shape_of
    params, key, ip, issuer, pres = make_batch(4, "SSPE", [0, 3], 3, b"panic-shapes")
    ctx = afx.Context(params, key, ip)
    base = shape_of(pres[0])
    assert gpu_verify(afx, ctx, pres) == [0, 0, 0]

    def variant(**kw):
        s = oracle.Shape.from_buffer_copy(bytes(base))
        for k, v in kw.items():
            if isinstance(v, tuple):
                getattr(s, k)[v[0]] = v[1]
            else:
                setattr(s, k, v)
        return s
    for s in (variant(n_hidden_scalars=0), variant(hidden_scalar_indices=(0, 1)), variant(hidden_scalar_indices=(0, 9)),
              variant(enc_indices=(0, 7)), variant(kinds=(1, 1)), variant(kinds=(0, 0)), variant(n_responses=3)):
        # same answer as the oracle given the same (mis-shaped) presentation
        want = []
        for p in pres:
            q = oracle.Presentation.from_buffer_copy(bytes(p))
            q.n_hidden_scalars = s.n_hidden_scalars
            q.n_responses = s.n_responses
            for k in range(32):
                q.hidden_scalar_indices[k] = s.hidden_scalar_indices[k]
                q.kinds[k] = s.kinds[k]
            for e in range(q.n_enc_proofs):
                q.enc[e].index = s.enc_indices[e]
            want.append(issuer.verify_presentation(q))
        assert gpu_verify(afx, ctx, pres, shape=s) == want
        assert want == [1, 1, 1]
    ctx.close()


def test_encryption_proofs_alone():
    import aeonflux_amd as afx
    from tests.soa import presentation_arrays
    params, key, ip, issuer, pres = make_batch(5, "SESSS", [1], 40, b"enc-alone")
    pres[3].enc[0].C_y_3[2] ^= 4
    pres[17].enc[0].pk[0] ^= 1
    a = presentation_arrays(pres)["enc"][0]
    soa = afx.EncProofSoA()
    for f in ("challenge", "responses", "pk", "E1", "E2", "C_y_1", "C_y_2", "C_y_3", "C_y_2p"):
        setattr(soa, f, a[f].ctypes.data)
    status = np.full(len(pres), 9, np.uint8)
    ctx = afx.Context(params, key, ip)
    ctx.verify_encryption_proofs(1, soa, len(pres), status.ctypes.data)
    ctx.close()
    assert status.tolist() == [issuer.verify_encryption_proof(p.enc[0]) for p in pres]
    assert status[3] == 1 and status[17] == 1 and status.sum() == 2


def test_empty_and_chunk_boundary_batches():
    """count == 0 is a no-op; a batch larger than the engine's internal pass size crosses pass boundaries with a ragged
    last pass (pass size set to 2^15 here; the default is 2^19).  The big batch is a tiling of 60 distinct
    presentations (some corrupted)."""
    import ctypes as C
    import aeonflux_amd as afx
    from aeonflux_amd import batch
    from tests.soa import presentation_arrays, shape_of
    params, key, ip, issuer, pres = make_batch(4, "SSPE", [0, 3], 60, b"gpu-chunks")
    corrupt(pres, b"chunk-corrupt")
    want = np.array([issuer.verify_presentation(p) for p in pres], np.uint8)
    a = presentation_arrays(pres)
    sh = afx.Shape.from_buffer_copy(bytes(shape_of(pres[0])))
    ctx = afx.Context(params, key, ip)
    ctx.set_chunk_items(1 << 15)
    soa, keep = batch.presentation_soa(a)
    st = np.full(4, 77, np.uint8)
    afx.check(afx.lib().afx_verify_presentations(ctx.h, C.byref(sh), C.byref(soa), 0, st.ctypes.data))
    assert st.tolist() == [77] * 4
    total = (1 << 17) + 77
    reps = -(-total // 60)
    tile = lambda x: np.ascontiguousarray(np.concatenate([x] * reps, axis=-2)[..., :total, :])
    big = {k: tile(v) for k, v in a.items() if k != "enc"}
    big["enc"] = [{k: tile(v) for k, v in d.items()} for d in a["enc"]]
    got = batch.verify_presentations(ctx, sh, big)
    ctx.close()
    assert np.array_equal(got, np.concatenate([want] * reps)[:total])
    assert 0 < got.sum() < total


def test_mixed_shapes_grouped_on_the_host():
    """presentations of one issuer with different hide/reveal choices (different shapes) in one request stream"""
    import aeonflux_amd as afx
    from aeonflux_amd import batch
    from tests.helpers import make_credentials
    from tests.soa import presentation_arrays, shape_of
    d = make_credentials(4, "SSPE", 12, b"gpu-mixed")
    take, user, issuer = d["take"], d["user"], d["issuer"]
    items, want = [], []
    for i, cr in enumerate(d["creds"]):
        hide = [[0, 3], [3], [0, 1], [], [1, 3]][i % 5]
        kinds = list(cr["kinds"])
        for j in hide:
            kinds[j] = 1 if kinds[j] == 0 else 4
        nsp = sum(1 for k in kinds if k == 4)
        st, p = user.show(kinds, cr["values"], cr["t"], cr["U"], cr["V"], user.keypair_derive(take(64)), take(64), take(32), take(32 * nsp))
        assert st == 0
        if i in (4, 7):
            p.C_V[3] ^= 1
        want.append(issuer.verify_presentation(p))
        items.append((afx.Shape.from_buffer_copy(bytes(shape_of(p))), presentation_arrays([p])))
    ctx = afx.Context(d["params"], d["key"], d["ip"])
    got = batch.verify_mixed(ctx, items)
    ctx.close()
    assert [int(g[0]) for g in got] == want and sum(want) == 2


def test_recomputed_challenges_equal_the_oracles():
    """Beyond accept/reject: for every item whose verification reaches the transcript stage in the oracle — valid or
    corrupted — the challenge the GPU recomputes (a hash over all of its recomputed commitments) is the oracle's, for
    the main proof and for every attached proof of encryption (afx_ctx_set_challenge_trace)."""
    import oracle
    import aeonflux_amd as afx
    n_items = 96
    params, key, ip, issuer, pres = make_batch(8, "SSPPEEEE", [4, 5, 6, 7], n_items, b"gpu-trace")
    corrupt(pres, b"trace-corrupt")
    ne = pres[0].n_enc_proofs
    want = [[None] * n_items for _ in range(1 + ne)]
    for i, p in enumerate(pres):
        q = oracle.Presentation.from_buffer_copy(bytes(p))
        q.n_enc_proofs = 0                                   # the main proof alone
        oracle.debug_reset()
        issuer.verify_presentation(q)
        commits, ch = oracle.debug_last()
        if commits:
            want[0][i] = ch
        for e in range(ne):
            oracle.debug_reset()
            issuer.verify_encryption_proof(p.enc[e])
            commits, ch = oracle.debug_last()
            if commits:
                want[1 + e][i] = ch
    ctx = afx.Context(params, key, ip)
    ctx.set_challenge_trace(1 + ne, n_items)
    got_status = gpu_verify(afx, ctx, pres)
    got = ctx.get_challenge_trace()
    ctx.set_challenge_trace(0, 0)
    ctx.close()
    assert got_status == [issuer.verify_presentation(p) for p in pres]
    reached = failing_reached = 0
    for r in range(1 + ne):
        for i in range(n_items):
            if want[r][i] is None:
                continue
            assert bytes(got[r, i]) == want[r][i], (r, i)
            reached += 1
            proof_challenge = bytes(pres[i].challenge) if r == 0 else bytes(pres[i].enc[r - 1].challenge)
            failing_reached += want[r][i] != proof_challenge
    assert reached > (1 + ne) * n_items * 0.8 and failing_reached >= 10


def test_concurrent_calls_from_threads():
    """two contexts used from two host threads at once, and two threads sharing one context (calls serialise on the
    context's mutex): every call returns the single-threaded answer"""
    import threading
    import aeonflux_amd as afx
    params, key, ip, issuer, pres = make_batch(4, "SSPE", [0, 3], 48, b"gpu-threads")
    corrupt(pres, b"threads-corrupt")
    want = [issuer.verify_presentation(p) for p in pres]
    a, b = afx.Context(params, key, ip), afx.Context(params, key, ip)
    results, errors = {}, []

    def work(name, ctx, rounds):
        try:
            for r in range(rounds):
                results[(name, r)] = gpu_verify(afx, ctx, pres)
        except Exception as e:   # noqa: BLE001 - reported below
            errors.append((name, repr(e)))
    threads = [threading.Thread(target=work, args=("a1", a, 4)), threading.Thread(target=work, args=("b1", b, 4)),
               threading.Thread(target=work, args=("a2", a, 4))]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    a.close()
    b.close()
    assert not errors, errors
    assert len(results) == 12 and all(v == want for v in results.values())


@pytest.mark.parametrize("layout,hide", [("E" * 32, list(range(32))), ("S" * 32, list(range(32))), ("P" * 32, [])])
def test_extreme_shapes_at_the_attribute_limit(layout, hide):
    """32 attributes, all of one kind: 32 proofs of encryption attached (the plan's largest: 1 + 2 + 32 * 5 multiscalar
    jobs), 32 hidden scalars (35 responses), 32 revealed points.  Status parity with the oracle, corrupted items included."""
    import aeonflux_amd as afx
    params, key, ip, issuer, pres = make_batch(32, layout, hide, 6, b"gpu-extreme-" + layout[:1].encode())
    pres[1].responses[0][9] ^= 8
    if pres[2].n_enc_proofs:
        pres[2].enc[31].C_y_2[4] ^= 1
    pres[4].C_y[17][0] ^= 2
    want = [issuer.verify_presentation(p) for p in pres]
    ctx = afx.Context(params, key, ip)
    got = gpu_verify(afx, ctx, pres)
    ctx.close()
    assert got == want and want[0] == 0 and want[1] == 1 and want[4] == 1


def test_identity_commitments_beside_ordinary_ones():
    """Commitments are encoded by k_compress2x with one inversion per item (Montgomery's trick over all of an item's commitments):
    a commitment that is the identity (zero factor in the product) must be rejected like zkp rejects it, and must not disturb
    the encodings of the item's other commitments.  All-zero challenge and responses make every commitment of that proof the
    identity; the proofs of encryption of the same item stay ordinary."""
    import oracle
    import aeonflux_amd as afx
    params, key, ip, issuer, pres = make_batch(8, "SSPPEEEE", [4, 5, 6, 7], 6, b"identity-commitments")
    assert [issuer.verify_presentation(p) for p in pres] == [0] * 6
    z = bytes(32)
    for k in range(32):
        pres[1].challenge[k] = 0                    # main proof of item 1: all commitments are the identity
        pres[3].enc[2].challenge[k] = 0             # third proof of encryption of item 3
    for r in range(pres[1].n_responses):
        for k in range(32):
            pres[1].responses[r][k] = 0
    for r in range(6):
        for k in range(32):
            pres[3].enc[2].responses[r][k] = 0
    want = [issuer.verify_presentation(p) for p in pres]
    assert want == [0, 1, 0, 1, 0, 0]
    ctx = afx.Context(params, key, ip)
    ctx.set_challenge_trace(5, 6)
    assert gpu_verify(afx, ctx, pres) == want
    # the untouched proofs of the two rejected items still recompute the challenges they carry (their encodings came out right)
    tr = ctx.get_challenge_trace()
    for e in range(4):
        assert bytes(tr[1 + e, 1]) == bytes(pres[1].enc[e].challenge)
    assert bytes(tr[0, 3]) == bytes(pres[3].challenge) and bytes(tr[1, 3]) == bytes(pres[3].enc[0].challenge)
    ctx.close()


def test_workspace_growth_while_an_earlier_dev_call_is_in_flight():
    """The *_dev calls are asynchronous.  A small call followed at once by one that needs a larger workspace must not have
    its live workspace (bad[] flags, tables, digits) wiped or freed under it (round-2 advisor finding: a cleared bad[] flag
    is a false accept).  Fresh contexts so that the second call really grows the buffer; repeated to give a race room."""
    import ctypes as C
    import aeonflux_amd as afx
    from aeonflux_amd import batch
    from tests.soa import presentation_arrays, shape_of
    params, key, ip, issuer, pres = make_batch(4, "SSPE", [0, 3], 48, b"gpu-grow-race")
    corrupt(pres, b"grow-race-corrupt")
    want = np.array([issuer.verify_presentation(p) for p in pres], np.uint8)
    assert want.any() and not want.all()
    a = presentation_arrays(pres)
    sh = afx.Shape.from_buffer_copy(bytes(shape_of(pres[0])))
    reps = 1400   # 48 * 1400 = 67 200 items: a workspace some hundred times the small call's
    big = {f: np.ascontiguousarray(np.concatenate([a[f]] * reps, axis=-2)) for f in batch.PRES_FIELDS}
    big["enc"] = [{f: np.ascontiguousarray(np.concatenate([d[f]] * reps, axis=-2)) for f in batch.ENC_FIELDS} for d in a["enc"]]
    from tests.helpers import DevMem

    def to_dev(p):
        d = {f: DevMem(p[f]) for f in batch.PRES_FIELDS}
        d["enc"] = [{f: DevMem(e[f]) for f in batch.ENC_FIELDS} for e in p["enc"]]
        return d
    d_small, d_big = to_dev(a), to_dev(big)
    soa_s, keep_s = batch.presentation_soa(d_small, ptr=lambda t: t.ptr)
    soa_b, keep_b = batch.presentation_soa(d_big, ptr=lambda t: t.ptr)
    fn = afx.lib().afx_verify_presentations_dev
    for _ in range(6):
        ctx = afx.Context(params, key, ip)
        st_s, st_b = DevMem(nbytes=48, fill=77), DevMem(nbytes=48 * reps, fill=77)
        afx.check(fn(ctx.h, C.byref(sh), C.byref(soa_s), 48, st_s.ptr))
        afx.check(fn(ctx.h, C.byref(sh), C.byref(soa_b), 48 * reps, st_b.ptr))   # no synchronisation in between
        ctx.synchronize()
        assert np.array_equal(st_s.numpy(), want)
        assert np.array_equal(st_b.numpy(), np.tile(want, reps))
        ctx.close()
        st_s.free()
        st_b.free()


def _mixed_stream(seed, n_creds=20):
    """presentations of one issuer with different hide/reveal choices, in arrival order, two of them corrupted;
    returns (ctx args, [(afx.Shape, arrays)], oracle statuses)"""
    import aeonflux_amd as afx
    from tests.helpers import make_credentials
    from tests.soa import presentation_arrays, shape_of
    d = make_credentials(4, "SSPE", n_creds, seed)
    take, user, issuer = d["take"], d["user"], d["issuer"]
    items, want = [], []
    for i, cr in enumerate(d["creds"]):
        hide = [[0, 3], [3], [0, 1], [], [1, 3]][(i * 3) % 5]   # shapes interleave: no two neighbours alike
        kinds = list(cr["kinds"])
        for j in hide:
            kinds[j] = 1 if kinds[j] == 0 else 4
        nsp = sum(1 for k in kinds if k == 4)
        st, p = user.show(kinds, cr["values"], cr["t"], cr["U"], cr["V"], user.keypair_derive(take(64)), take(64), take(32), take(32 * nsp))
        assert st == 0
        if i in (4, 7, 13):
            p.C_V[3] ^= 1
        want.append(issuer.verify_presentation(p))
        items.append((afx.Shape.from_buffer_copy(bytes(shape_of(p))), presentation_arrays([p])))
    return d, items, want


def test_mixed_shapes_through_the_c_entry_points():
    """Issuer::verify takes any presentation (src/issuer.rs:141-147; the shape is per presentation, presentation.rs:293-309):
    a stream of interleaved shapes goes through afx_verify_presentations_mixed (struct-of-arrays groups with positions),
    through afx_verify_presentations_mixed_wire (AFXP sections back to back, grouped inside the library), and through the
    group (multi-device) form; every status must be the oracle's, at the presentation's own place in the stream."""
    import ctypes as C
    import aeonflux_amd as afx
    from aeonflux_amd import batch, wire
    d, items, want = _mixed_stream(b"gpu-mixed-c")
    assert sum(want) == 3 and len({bytes(s) for s, _ in items}) == 5
    ctx = afx.Context(d["params"], d["key"], d["ip"])
    got = batch.verify_mixed(ctx, items)
    assert [int(g[0]) for g in got] == want
    blob = wire.pack_mixed(items)
    assert wire.verify_mixed_wire(ctx, blob).tolist() == want
    # sections with count > 1, shapes repeating non-adjacently, garbage in the unused shape tails (must not split a group)
    def cat(idx):
        ps = [items[i][1] for i in idx]
        out = {f: np.concatenate([p[f] for p in ps], axis=-2) for f in batch.PRES_FIELDS}
        out["enc"] = [{f: np.concatenate([p["enc"][e][f] for p in ps], axis=-2) for f in batch.ENC_FIELDS} for e in range(len(ps[0]["enc"]))]
        return out
    by_shape = {}
    for i, (s, _) in enumerate(items):
        by_shape.setdefault(bytes(s), []).append(i)
    sections, order = [], []
    for idx in by_shape.values():
        half = len(idx) // 2
        sections.append((items[idx[0]][0], cat(idx[:half]), idx[:half]))
        sections.append((items[idx[0]][0], cat(idx[half:]), idx[half:]))
    sections = sections[::2] + sections[1::2]   # first halves of every shape, then second halves
    for s, _, idx in sections:
        order += idx
    st = wire.verify_mixed_wire(ctx, b"".join(wire.pack_presentations(s, p) for s, p, _ in sections))
    assert st.tolist() == [want[i] for i in order]
    dirty = []
    for s, p in items:
        s2 = afx.Shape.from_buffer_copy(bytes(s))
        for k in range(s2.n_attributes, 32):
            s2.kinds[k] = 0xAB
        for k in range(s2.n_enc_proofs, 32):
            s2.enc_indices[k] = 0x7777
        dirty.append((s2, p))
    assert [int(g[0]) for g in batch.verify_mixed(ctx, dirty)] == want
    # bad requests are refused before anything runs
    n = C.c_size_t(0)
    stb = np.zeros(64, np.uint8)
    assert afx.lib().afx_verify_presentations_mixed_wire(ctx.h, blob[:-7], len(blob) - 7, stb.ctypes.data, 64, C.byref(n)) == afx.E_BAD_ARGS
    assert afx.lib().afx_verify_presentations_mixed_wire(ctx.h, blob, len(blob), stb.ctypes.data, 3, C.byref(n)) == afx.E_BAD_ARGS and n.value == len(items)
    assert afx.lib().afx_verify_presentations_mixed_wire(ctx.h, b"", 0, stb.ctypes.data, 64, C.byref(n)) == afx.OK and n.value == 0
    g = (afx.PresentationGroup * 2)()
    soa, keep = batch.presentation_soa(items[0][1])
    pos = (C.c_uint64 * 1)(0)
    for k in range(2):
        g[k].shape, g[k].batch, g[k].count, g[k].positions = items[0][0], soa, 1, pos
    assert afx.lib().afx_verify_presentations_mixed(ctx.h, g, 2, stb.ctypes.data, 2) == afx.E_BAD_ARGS   # position 0 twice
    pos[0] = 5
    assert afx.lib().afx_verify_presentations_mixed(ctx.h, g, 1, stb.ctypes.data, 2) == afx.E_BAD_ARGS   # outside the array
    ctx.close()
    grp = afx.Group(d["params"], d["key"], d["ip"], [0, 0])
    assert [int(x[0]) for x in batch.verify_mixed(grp, items)] == want
    grp.close()


def test_the_three_plans_agree():
    """One batch of 9000 presentations (60 distinct ones tiled, a third corrupted) under the plan of large passes (one chain per
    job), the latency plan (one chain per term - at this size up to four terms per chain in the wide stage -, -c*Z over Z's terms) and the plan in between (only the job that multiplies by
    the issuer key split into NAF chains): the same statuses - the oracle's - and the same recomputed challenges."""
    import aeonflux_amd as afx
    from aeonflux_amd import batch
    from tests.soa import presentation_arrays, shape_of
    params, key, ip, issuer, pres = make_batch(8, "SSPPEEEE", [4, 5, 6, 7], 60, b"gpu-three-plans")
    corrupt(pres, b"three-plans-corrupt")
    want = np.array([issuer.verify_presentation(p) for p in pres], np.uint8)
    a = presentation_arrays(pres)
    sh = afx.Shape.from_buffer_copy(bytes(shape_of(pres[0])))
    reps, total = 150, 9000
    big = {f: np.ascontiguousarray(np.concatenate([a[f]] * reps, axis=-2)) for f in batch.PRES_FIELDS}
    big["enc"] = [{f: np.ascontiguousarray(np.concatenate([d[f]] * reps, axis=-2)) for f in batch.ENC_FIELDS} for d in a["enc"]]
    ctx = afx.Context(params, key, ip)
    seen = {}
    for name, thr in (("one chain per job", 0), ("one chain per term", 16384), ("key job split", 4096)):
        ctx.set_small_batch_items(thr)
        ctx.set_challenge_trace(5, total)
        st = batch.verify_presentations(ctx, sh, big)
        tr = ctx.get_challenge_trace()
        ctx.set_challenge_trace(0, 0)
        assert np.array_equal(st, np.tile(want, reps)), name
        seen[name] = (tr, ctx.plan_stats()["msm_jobs"])
        if name == "one chain per job":
            # the per-presentation operation counts DESIGN.md section 3 publishes for C3 (afx_ctx_get_plan_stats of the plan of
            # large passes); chain_*: the share of the field operations inside the inversion / square-root chains (fe10.cuh)
            ps = ctx.plan_stats()
            published = {"msm_jobs": 27, "doublings": 6804, "fixed_additions": 740, "table_additions": 287, "encodings": 35, "decodings": 41,
                         "keccak_permutations": 41, "field_sq": 39796, "chain_mul": 528, "chain_sq": 12054, "secret_terms": 0}
            assert {k: ps[k] for k in published} == published, ps
            # the key's ten scalars run as width-5 NAF schedules: ~42 additions each, the exact number is the key's NAF weight
            # (3182 and 55 691 products with the bench fixture's key)
            assert 3140 <= ps["var_additions"] <= 3230 and ps["field_mul"] == 55691 + 8 * (ps["var_additions"] - 3182), ps
    ctx.close()
    assert np.array_equal(seen["one chain per job"][0], seen["one chain per term"][0]) and np.array_equal(seen["one chain per job"][0], seen["key job split"][0])
    assert seen["one chain per job"][1] < seen["key job split"][1] < seen["one chain per term"][1]   # 27 < 27 - 1 + 11 < 86
