"""GPU: one batch over several contexts (include/aeonflux_gpu.h afx_group_*, afx_*_range).
A two-member group (device 0 listed twice when the box has one GPU, devices 0 and 1 otherwise), explicit ranges and the
sliced host-pointer pipeline must give the single-context answer, which must be the oracle's."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _devices():
    import torch
    return [0, 1] if torch.cuda.device_count() > 1 else [0, 0]


def _oracle_statuses(params, key, ip, shape, pres, count):
    import oracle
    from aeonflux_amd import batch
    olib = oracle.load(native=True)
    olib.afxo_ctx_new.restype = C.c_void_p
    octx = olib.afxo_ctx_new(params, len(params), key, len(key), ip)
    osoa, keep = batch.presentation_soa(pres)
    ost = np.full(count, 255, np.uint8)
    olib.afxo_verify_presentations_soa(C.c_void_p(octx), C.byref(oracle.Shape.from_buffer_copy(bytes(shape))),
                                       C.byref(oracle.PresentationSoA.from_buffer_copy(bytes(osoa))), count, ost.ctypes.data, 8)
    return ost


def test_group_ranges_and_slices_equal_one_context_and_the_oracle():
    import aeonflux_amd as afx
    import bench
    from aeonflux_amd import batch
    params, key, ip = bench.load_fixture("c3_8attrs_SSPPeeee")
    issuer = afx.Context(params, key, ip)
    user = afx.Context(params, None, ip)
    count = 3001
    pres, shape = bench.generate(afx, batch, issuer, user, params, 8, "SSPPEEEE", [4, 5, 6, 7], count, 2024)
    user.close()
    want = bench.corrupt(pres, count, 77)
    pres["responses"][1, 2999, 4] ^= 2
    want[2999] = 1
    one = batch.verify_presentations(issuer, shape, pres)
    assert np.array_equal(one, want)
    assert np.array_equal(_oracle_statuses(params, key, ip, shape, pres, count), one)
    # the group: every member its contiguous part, status concatenated in place
    grp = afx.Group(params, key, ip, _devices())
    grp.member(0).set_small_batch_items(0)   # these batches are small: without this the group hands each call to ONE member (tested below)
    assert len(grp) == 2
    assert np.array_equal(batch.verify_presentations(grp, shape, pres), one)
    # explicit ranges through a second context: three parts, written into one status array
    other = afx.Context(params, key, ip)
    merged = np.full(count, 255, np.uint8)
    for r in range(3):
        first, n = afx.shard_bounds(count, 3, r)
        part = batch.verify_presentations(other if r % 2 else issuer, shape, pres, first=first, n=n)
        assert (part[:first] == 255).all() and (part[first + n:] == 255).all()
        merged[first:first + n] = part[first:first + n]
    assert np.array_equal(merged, one)
    # many small slices on alternating streams (the host-pointer pipeline), ragged tail included
    other.set_chunk_items(256)
    assert np.array_equal(batch.verify_presentations(other, shape, pres), one)
    # the same batch as one serialized blob (AFXP): slices are byte ranges of the record area
    from aeonflux_amd import wire
    blob = wire.pack_presentations(shape, pres)
    stw, cw = np.full(count, 9, np.uint8), C.c_size_t(0)
    afx.check(afx.lib().afx_verify_presentations_wire(other.h, blob, len(blob), stw.ctypes.data, count, C.byref(cw)))
    assert cw.value == count and np.array_equal(stw, one)
    grp.member(0).set_chunk_items(512)
    assert np.array_equal(batch.verify_presentations(grp, shape, pres), one)
    # empty and one-item batches
    assert len(batch.verify_presentations(grp, shape, {f: (v[..., :0, :] if f != "enc" else [{g: w[..., :0, :] for g, w in d.items()} for d in v])
                                                       for f, v in pres.items()})) == 0
    one_item = {f: (v[..., 5:6, :] if f != "enc" else [{g: w[..., 5:6, :] for g, w in d.items()} for d in v]) for f, v in pres.items()}
    assert batch.verify_presentations(grp, shape, one_item).tolist() == [int(one[5])]
    grp.close()
    other.close()
    issuer.close()


def test_group_issue_equals_one_context_and_the_oracle():
    import oracle
    import aeonflux_amd as afx
    import bench
    from aeonflux_amd import batch
    n, layout, count = 16, "SSSSSSSSPPPPEEEE", 1500
    params, key, ip = bench.load_fixture("c5_16attrs")
    issuer = afx.Context(params, key, ip)
    rng = np.random.default_rng(11)
    rb = lambda *s: rng.integers(0, 256, size=s, dtype=np.uint8)
    kinds = [{"S": afx.ATTR_PUBLIC_SCALAR, "P": afx.ATTR_PUBLIC_POINT, "E": afx.ATTR_EITHER_POINT}[c] for c in layout]
    values = np.stack([batch.scalars_from_wide(issuer, rb(count, 64)) if c == "S" else batch.points_from_uniform(issuer, rb(count, 64)) for c in layout])
    values[3, 7] = 0xFF            # a non-canonical scalar attribute: that request fails alone
    tw, uw, seed = rb(count, 64), rb(count, 64), rb(count, 32)
    o1, s1 = batch.issue(issuer, kinds, values, tw, uw, seed)
    assert s1[7] == afx.ST_MAC_CREATION and s1.sum() == afx.ST_MAC_CREATION
    grp = afx.Group(params, key, ip, _devices())
    grp.member(0).set_small_batch_items(0)   # these batches are small: without this the group hands each call to ONE member (tested below)
    grp.member(1).set_chunk_items(256)
    o2, s2 = batch.issue(grp, kinds, values, tw, uw, seed)
    assert np.array_equal(s1, s2)
    ok = s1 == 0
    for f in ("t", "U", "V", "challenge", "responses"):
        assert np.array_equal(o1[f][..., ok, :], o2[f][..., ok, :]), f
    # a range leaves everything outside it untouched
    o3, s3 = batch.issue(issuer, kinds, values, tw, uw, seed, first=100, n=50)
    assert (s3[:100] == 255).all() and (s3[150:] == 255).all() and not s3[100:150].any()
    assert np.array_equal(o3["responses"][:, 100:150], o1["responses"][:, 100:150]) and not o3["responses"][:, :100].any() and not o3["V"][150:].any()
    # the oracle's bytes for a few items
    octx = oracle.Ctx(params, key, ip)
    for i in (0, 1, 749, 750, 1499):
        vals = [bytes(values[k, i]) + bytes(64) for k in range(n)]
        st, t, U, V, ch, resp = octx.issue(kinds, vals, bytes(tw[i]), bytes(uw[i]), bytes(seed[i]))
        assert st == 0 and t == bytes(o2["t"][i]) and U == bytes(o2["U"][i]) and V == bytes(o2["V"][i]) and ch == bytes(o2["challenge"][i])
        assert all(resp[k] == bytes(o2["responses"][k, i]) for k in range(n + 5))
    # and the issued credentials verify on the user side
    user = afx.Context(params, None, ip)
    sv = batch.verify_issuances(user, kinds, values, o2)
    assert not sv[ok].any() and sv[7] == afx.ST_VERIFICATION_FAILURE
    # ... also as one serialized blob (AFXI) in slices of 256 records
    from aeonflux_amd import wire
    o2["responses"][5, 1000, 1] ^= 4
    sv[1000] = afx.ST_VERIFICATION_FAILURE
    user.set_chunk_items(256)
    assert np.array_equal(user.verify_issuances_wire(wire.pack_issuances(kinds, values, o2)), sv)
    user.close()
    grp.close()
    issuer.close()


def test_group_show_and_verify_issuances_equal_one_context():
    """afx_group_show / afx_group_verify_issuances and their _range forms (the user side on several GPUs): byte-equal to the
    single-context calls, ragged slices and a range included; the presentations then verify"""
    import aeonflux_amd as afx
    import bench
    from aeonflux_amd import batch
    n, layout, hide, count = 4, "SSPE", [0, 3], 1203
    params, key, ip = bench.load_fixture("readme_4attrs_sSPe")
    issuer = afx.Context(params, key, ip)
    user = afx.Context(params, None, ip)
    rng = np.random.default_rng(31)
    rb = lambda *s: rng.integers(0, 256, size=s, dtype=np.uint8)
    kinds = [afx.ATTR_PUBLIC_SCALAR, afx.ATTR_PUBLIC_SCALAR, afx.ATTR_PUBLIC_POINT, afx.ATTR_EITHER_POINT]
    values = np.stack([batch.scalars_from_wide(issuer, rb(count, 64)), batch.scalars_from_wide(issuer, rb(count, 64)),
                       batch.points_from_uniform(issuer, rb(count, 64)), batch.points_from_uniform(issuer, rb(count, 64))])
    M2, m3 = np.zeros((n, count, 32), np.uint8), np.zeros((n, count, 32), np.uint8)
    M2[3], m3[3] = batch.points_from_uniform(issuer, rb(count, 64)), batch.scalars_from_wide(issuer, rb(count, 64))
    iss, st = batch.issue(issuer, kinds, values, rb(count, 64), rb(count, 64), rb(count, 32))
    assert not st.any()
    bad = {k: v.copy() for k, v in iss.items()}
    bad["responses"][2, 1100, 0] ^= 1
    bad["V"][17] = bad["V"][18]
    # user-side group (no issuer key)
    grp = afx.Group(params, None, ip, _devices())
    grp.member(0).set_small_batch_items(0)   # see above
    grp.member(1).set_chunk_items(256)
    s1 = batch.verify_issuances(user, kinds, values, bad)
    assert s1.sum() == 2 * afx.ST_VERIFICATION_FAILURE and s1[17] and s1[1100]
    assert np.array_equal(batch.verify_issuances(grp, kinds, values, bad), s1)
    part = batch.verify_issuances(user, kinds, values, bad, first=1000, n=203)
    assert (part[:1000] == 255).all() and np.array_equal(part[1000:], s1[1000:])
    # show
    skinds = [afx.ATTR_SECRET_SCALAR, afx.ATTR_PUBLIC_SCALAR, afx.ATTR_PUBLIC_POINT, afx.ATTR_SECRET_POINT]
    g = max(3, n)
    gen = lambda idx: np.frombuffer(params[4 + 32 * idx:4 + 32 * idx + 32], np.uint8)
    a, a0, a1 = (batch.scalars_from_wide(issuer, rb(count, 64)) for _ in range(3))
    bases = np.stack([np.broadcast_to(gen(5 + g + n + 1 + k), (count, 32)) for k in range(3)])
    pk, ok = batch.multiscalar_mul(issuer, np.stack([a, a0, a1]), bases)
    kp = dict(a=a, a0=a0, a1=a1, pk=pk)
    zw, seed, es = rb(count, 64), rb(count, 32), rb(1, count, 32)
    p1, sh1, t1 = batch.show(user, skinds, values, iss["t"], iss["U"], iss["V"], kp, zw, seed, es, M2, m3)
    assert not t1.any()
    p2, sh2, t2 = batch.show(grp, skinds, values, iss["t"], iss["U"], iss["V"], kp, zw, seed, es, M2, m3)
    assert bytes(sh1) == bytes(sh2) and np.array_equal(t1, t2)

    def same(x, y, sel=slice(None)):
        for f, v in x.items():
            if f == "enc":
                for d, e in zip(v, y["enc"]):
                    for h, w in d.items():
                        assert np.array_equal(w[..., sel, :], e[h][..., sel, :]), ("enc", h)
            else:
                assert np.array_equal(v[..., sel, :], y[f][..., sel, :]), f
    same(p1, p2)
    p3, sh3, t3 = batch.show(user, skinds, values, iss["t"], iss["U"], iss["V"], kp, zw, seed, es, M2, m3, first=300, n_items=77)
    assert bytes(sh3) == bytes(sh1) and (t3[:300] == 255).all() and (t3[377:] == 255).all() and not t3[300:377].any()
    same(p1, p3, slice(300, 377))
    assert not p3["C_V"][:300].any() and not p3["enc"][0]["responses"][:, 377:].any()
    # no keypair with a hidden group element: every status NoSymmetricKey, through the group as well
    _, _, t4 = batch.show(grp, skinds, values, iss["t"], iss["U"], iss["V"], None, zw, seed, es, M2, m3)
    assert (t4 == afx.ST_NO_SYMMETRIC_KEY).all()
    # and what the group showed verifies
    assert not batch.verify_presentations(issuer, sh2, p2).any()
    grp.close()
    user.close()
    issuer.close()


def test_fixed_key_schedule_gives_identical_results():
    """afx_ctx_set_fixed_key_schedule: the issuer key's scalars without NAF (running time independent of the key)"""
    import aeonflux_amd as afx
    import bench
    from aeonflux_amd import batch
    params, key, ip = bench.load_fixture("readme_4attrs_sSPe")
    issuer = afx.Context(params, key, ip)
    user = afx.Context(params, None, ip)
    count = 700
    issuer.set_small_batch_items(0)   # the plan of large passes: that is where the key's scalars run as NAF schedules
    pres, shape = bench.generate(afx, batch, issuer, user, params, 4, "SSPE", [0, 3], count, 606)
    want = bench.corrupt(pres, count, 8)
    assert np.array_equal(batch.verify_presentations(issuer, shape, pres), want)
    naf = issuer.plan_stats()
    issuer.set_fixed_key_schedule(True)
    assert np.array_equal(batch.verify_presentations(issuer, shape, pres), want)
    fixed = issuer.plan_stats()
    assert fixed["var_additions"] > naf["var_additions"] and fixed["doublings"] >= naf["doublings"]
    rng = np.random.default_rng(12)
    rb = lambda *s: rng.integers(0, 256, size=s, dtype=np.uint8)
    kinds = [afx.ATTR_PUBLIC_SCALAR, afx.ATTR_PUBLIC_SCALAR, afx.ATTR_PUBLIC_POINT, afx.ATTR_EITHER_POINT]
    values = np.stack([batch.scalars_from_wide(issuer, rb(64, 64)), batch.scalars_from_wide(issuer, rb(64, 64)),
                       batch.points_from_uniform(issuer, rb(64, 64)), batch.points_from_uniform(issuer, rb(64, 64))])
    tw, uw, sd = rb(64, 64), rb(64, 64), rb(64, 32)
    a, sa = batch.issue(issuer, kinds, values, tw, uw, sd)
    issuer.set_fixed_key_schedule(False)
    b, sb = batch.issue(issuer, kinds, values, tw, uw, sd)
    assert not sa.any() and not sb.any()
    for f in ("t", "U", "V", "challenge", "responses"):
        assert np.array_equal(a[f], b[f]), f
    user.close()
    issuer.close()


def test_group_wire_and_mixed_wire_equal_one_context():
    """a serialized batch, and a stream of sections of mixed shapes, over a group's members (byte ranges of the caller's blob):
    the statuses of one context, which are the oracle's; ranges of a blob on one context too"""
    import aeonflux_amd as afx
    from aeonflux_amd import wire
    from tests.helpers import corrupt, make_batch
    from tests.soa import presentation_arrays, shape_of
    params, key, ip, issuer, pres = make_batch(4, "SSPE", [0, 3], 301, b"gpu-group-wire")
    corrupt(pres, b"group-wire-corrupt")
    want = [issuer.verify_presentation(p) for p in pres]
    sh = afx.Shape.from_buffer_copy(bytes(shape_of(pres[0])))
    blob = wire.pack_presentations(sh, presentation_arrays(pres))
    one = afx.Context(params, key, ip)
    grp = afx.Group(params, key, ip, _devices())
    grp.member(0).set_small_batch_items(0)   # these batches are small: without this the group hands each call to ONE member (tested below)
    for m in range(2):
        grp.member(m).set_chunk_items(256)   # several slices per member
    assert wire.verify_wire(one, blob).tolist() == want
    assert wire.verify_wire(grp, blob).tolist() == want
    part = wire.verify_wire(one, blob, first=100, n=57)
    assert part[100:157].tolist() == want[100:157] and (part[:100] == 255).all() and (part[157:] == 255).all()
    # a second shape (the hidden scalar revealed) in between two runs of the first
    params2, key2, ip2 = params, key, ip
    from tests.helpers import make_credentials
    d = make_credentials(4, "SSPE", 40, b"gpu-group-wire")   # same seed: same parameters and key
    assert d["params"] == params and d["key"] == key
    take, user = d["take"], d["user"]
    others = []
    for cr in d["creds"]:
        kinds = list(cr["kinds"])
        kinds[3] = 4
        st, p = user.show(kinds, cr["values"], cr["t"], cr["U"], cr["V"], user.keypair_derive(take(64)), take(64), take(32), take(32))
        assert st == 0
        others.append(p)
    others[7].C_V[1] ^= 2
    want2 = [issuer.verify_presentation(p) for p in others]
    sh2 = afx.Shape.from_buffer_copy(bytes(shape_of(others[0])))
    assert bytes(sh2) != bytes(sh)
    stream = wire.pack_presentations(sh, presentation_arrays(pres[:150])) + wire.pack_presentations(sh2, presentation_arrays(others)) + \
        wire.pack_presentations(sh, presentation_arrays(pres[150:]))
    expect = want[:150] + want2 + want[150:]
    assert wire.verify_mixed_wire(one, stream).tolist() == expect
    assert wire.verify_mixed_wire(grp, stream).tolist() == expect
    grp.close()
    one.close()


def test_small_group_calls_go_to_one_member_in_turn():
    """a call small enough for the latency plan is not cut up: it goes to one member, the next in turn (so that small calls
    from several threads spread over the devices); same statuses either way"""
    import aeonflux_amd as afx
    from aeonflux_amd import batch
    from tests.helpers import corrupt, make_batch
    from tests.soa import presentation_arrays, shape_of
    params, key, ip, issuer, pres = make_batch(4, "SSPE", [0, 3], 50, b"gpu-group-small")
    corrupt(pres, b"group-small-corrupt")
    want = [issuer.verify_presentation(p) for p in pres]
    a = presentation_arrays(pres)
    sh = afx.Shape.from_buffer_copy(bytes(shape_of(pres[0])))
    grp = afx.Group(params, key, ip, _devices())
    m0, m1 = grp.member(0), grp.member(1)
    for m in (m0, m1):
        m.set_timing(True)
    for _ in range(4):
        assert batch.verify_presentations(grp, sh, a).tolist() == want
    launches = [m.get_timing("k_finish")[1] for m in (m0, m1)]
    assert launches == [2, 2], launches        # four calls, two on each member
    m0.set_small_batch_items(0)                # the shortcut off: both members take a part of every call
    for _ in range(2):
        assert batch.verify_presentations(grp, sh, a).tolist() == want
    assert [m.get_timing("k_finish")[1] for m in (m0, m1)] == [4, 4]
    grp.close()
