"""GPU: every plan variant of a small pass against the ORACLE, byte for byte.

Which of several equivalent layouts a small pass takes is decided by its size and by the width of the launch set it shares with
other calls (include/aeonflux_gpu.h afx_ctx_set_plan_variants): secret scalars of a prover pass in 8 / 4 / 1 segment(s) over the
bases' powers (engine.cpp Assembler::segments), chains on four waves per item or one (kernels.hip k_msm_quad / k_msm), a transcript
on a wave or on 32 lanes (k_hash_coop64 / k_hash_coop), sums of many parts with a lane per part or per item (k_pointsum_tree /
k_pointsum_quad / k_pointsum).  The other GPU tests compare the AUTOMATIC choice with the oracle at their sizes; here every
alternative is forced through the API at sizes on both sides of the thresholds (1, 70, 300 items: 8 segments up to 256 items, 4
above; cached narrow tables up to 512) and issue (/root/reference/src/issuer.rs:111-124), show (src/credential.rs:37-46) and verify
(src/issuer.rs:141-147) are compared with the oracle's bytes - not with another GPU run (rounds 4-5 compared digests of two GPU
runs; the round-5 review asked for this)."""
import numpy as np
import pytest

from tests.helpers import corrupt, gpu_verify, make_credentials
from tests.test_gpu_prove import gpu_issue, gpu_show

pytestmark = pytest.mark.gpu

SIZES = (1, 70, 300)
N, LAYOUT, HIDE = 6, "SPPEES", [0, 3, 4]


@pytest.fixture(scope="module")
def world():
    """300 oracle-issued credentials and their oracle-made presentations, from explicit randomness"""
    d = make_credentials(N, LAYOUT, max(SIZES), b"gpu-plan-variants")
    take, user = d["take"], d["user"]
    kinds = list(d["creds"][0]["kinds"])
    shown = [1 if (i in HIDE and k == 0) else 4 if i in HIDE else k for i, k in enumerate(kinds)]
    nsp = sum(1 for k in shown if k == 4)
    cnt = max(SIZES)
    kps = [user.keypair_derive(take(64)) for _ in range(cnt)]
    zw = [take(64) for _ in range(cnt)]
    sd = [take(32) for _ in range(cnt)]
    es = [take(32 * nsp) for _ in range(cnt)]
    pres = []
    for c, kp, z, s, e in zip(d["creds"], kps, zw, sd, es):
        st, p = user.show(shown, c["values"], c["t"], c["U"], c["V"], kp, z, s, e)
        assert st == 0
        pres.append(p)
    return dict(d=d, kinds=kinds, shown=shown, nsp=nsp, kps=kps, zw=zw, sd=sd, es=es, pres=pres)


def variants(afx):
    return [
        ("automatic", 0),
        ("whole chains", afx.VARIANT_SEGMENTS_1),
        ("2 segments", afx.VARIANT_SEGMENTS_2),
        ("4 segments", afx.VARIANT_SEGMENTS_4),
        ("one-wave chains", afx.VARIANT_ONE_WAVE_CHAINS),
        ("one-wave chains, 4 segments", afx.VARIANT_ONE_WAVE_CHAINS | afx.VARIANT_SEGMENTS_4),
        ("transcripts on 32 lanes", afx.VARIANT_HASH_HALF_WAVE),
        ("sums without the tree", afx.VARIANT_NO_POINTSUM_TREE),
        ("everything off", afx.VARIANT_SEGMENTS_1 | afx.VARIANT_ONE_WAVE_CHAINS | afx.VARIANT_HASH_HALF_WAVE | afx.VARIANT_NO_POINTSUM_TREE),
        ("automatic + self-check", afx.VARIANT_SELFCHECK),
    ]


def check_issue(afx, ctx, w, count):
    cr = w["d"]["creds"][:count]
    vals = [[c["values"][i][:32] for c in cr] for i in range(N)]
    o, st = gpu_issue(afx, ctx, w["kinds"], vals, [c["rnd"][0] for c in cr], [c["rnd"][1] for c in cr], [c["rnd"][2] for c in cr])
    assert st.tolist() == [0] * count
    for i, c in enumerate(cr):
        for f in ("t", "U", "V", "challenge"):
            assert bytes(o[f][32 * i:32 * i + 32]) == c[f], (f, i)
        for k in range(N + 5):
            off = 32 * (k * count + i)
            assert bytes(o["responses"][off:off + 32]) == c["responses"][k], ("response", k, i)


def check_show(afx, ctx, w, count):
    o, shape, st = gpu_show(afx, ctx, w["shown"], w["d"]["creds"][:count], w["kps"][:count], w["zw"][:count], w["sd"][:count], w["es"][:count])
    assert st.tolist() == [0] * count
    cell = lambda arr, k, i: bytes(arr[32 * (k * count + i):32 * (k * count + i) + 32])
    for i, p in enumerate(w["pres"][:count]):
        assert cell(o["challenge"], 0, i) == bytes(p.challenge), i
        for k in range(p.n_responses):
            assert cell(o["responses"], k, i) == bytes(p.responses[k]), (k, i)
        assert cell(o["C_x_0"], 0, i) == bytes(p.C_x_0) and cell(o["C_x_1"], 0, i) == bytes(p.C_x_1) and cell(o["C_V"], 0, i) == bytes(p.C_V)
        for k in range(N):
            assert cell(o["C_y"], k, i) == bytes(p.C_y[k])
        for e in range(w["nsp"]):
            q, g = p.enc[e], o["enc"][e]
            assert cell(g["challenge"], 0, i) == bytes(q.challenge)
            for k in range(6):
                assert cell(g["responses"], k, i) == bytes(q.responses[k])
            for f in ("pk", "E1", "E2", "C_y_1", "C_y_2", "C_y_3", "C_y_2p"):
                assert cell(g[f], 0, i) == bytes(getattr(q, f)), f


def test_every_prover_variant_returns_the_oracles_bytes(world):
    import aeonflux_amd as afx
    d = world["d"]
    for mode in (2, 0):   # the library's default and the fast tables (where small passes still take the secret-independent plan)
        ictx = afx.Context(d["params"], d["key"], d["ip"])
        uctx = afx.Context(d["params"], None, d["ip"])
        ictx.set_secret_independent_addressing(mode)
        uctx.set_secret_independent_addressing(mode)
        for name, flags in variants(afx):
            ictx.set_plan_variants(flags)
            uctx.set_plan_variants(flags)
            for count in SIZES:
                try:
                    check_issue(afx, ictx, world, count)
                    check_show(afx, uctx, world, count)
                except AssertionError as e:
                    raise AssertionError("variant %r, secret mode %d, %d items: %s" % (name, mode, count, e))
        ictx.close()
        uctx.close()


def test_the_automatic_choice_takes_the_segmented_chains_and_the_forced_one_does_not(world):
    """the flags do select: k_powers runs for a small prover pass by default and not under AFX_VARIANT_SEGMENTS_1"""
    import aeonflux_amd as afx
    d = world["d"]
    ctx = afx.Context(d["params"], d["key"], d["ip"])
    for flags, expect in ((0, True), (afx.VARIANT_SEGMENTS_1, False), (afx.VARIANT_SEGMENTS_4, True)):
        ctx.set_plan_variants(flags)
        ctx.set_timing(True)
        check_issue(afx, ctx, world, 70)
        ms, launches = ctx.get_timing("k_powers")
        ctx.set_timing(False)
        assert (launches >= 1) == expect, (flags, launches)
    with pytest.raises(afx.AfxError):
        ctx.set_plan_variants(afx.VARIANT_SEGMENTS_1 | afx.VARIANT_SEGMENTS_2)   # at most one segment count
    with pytest.raises(afx.AfxError):
        ctx.set_plan_variants(0x1000)
    ctx.close()


def test_every_verifier_variant_returns_the_oracles_statuses_and_challenges(world):
    """Issuer::verify on a partly corrupted batch: statuses and every recomputed challenge (main proof and each proof of encryption)
    against the oracle's, under each variant"""
    import copy
    import oracle
    import aeonflux_amd as afx
    from tests.soa import pack_presentations
    d = world["d"]
    pres = [copy.deepcopy(p) for p in world["pres"]]
    corrupt(pres, b"gpu-plan-variants-corrupt")
    want = {}
    for count in SIZES:
        sh, soa, keep = pack_presentations(pres[:count])
        st, trace, reached = oracle.verify_presentations_traced(d["issuer"], sh, soa, count)
        want[count] = (st.tolist(), trace, reached.astype(bool))
    assert 0 < sum(want[max(SIZES)][0]) < max(SIZES)
    ctx = afx.Context(d["params"], d["key"], d["ip"])
    for mode in (2, 1):
        ctx.set_secret_independent_addressing(mode)
        for name, flags in variants(afx):
            ctx.set_plan_variants(flags)
            for count in SIZES:
                st, trace, reached = want[count]
                ctx.set_challenge_trace(1 + world["nsp"], count)
                assert gpu_verify(afx, ctx, pres[:count]) == st, (name, mode, count)
                got = ctx.get_challenge_trace()
                ctx.set_challenge_trace(0, 0)
                assert reached.sum() > 0.9 * reached.size
                assert np.array_equal(got[reached], trace[reached]), (name, mode, count)
    ctx.close()
