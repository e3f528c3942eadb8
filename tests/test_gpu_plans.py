"""GPU parity for REUSED plans.  A small call's launch list is assembled once per (statement, shape, padded size) and kept on the
context (afx::Plan, plans.cpp run_chunked: position-independent, relocated to wherever the call's staging and workspace stand); every
later call of that statement and shape up to the padded size runs the kept plan on its own data.  The reference has no such state
(Issuer::verify is a pure function of the presentation, /root/reference/src/issuer.rs:141-145), so nothing of an earlier call may show
in a later one: calls of different counts, different items, other shapes in between and a changed workspace must each give the
ORACLE's statuses / bytes, with the plan self-check on (every plan assembled twice against different provisional bases and compared
after relocation) as well as off."""
import numpy as np
import pytest

from tests.helpers import corrupt, gpu_verify, make_batch, make_credentials

pytestmark = pytest.mark.gpu


def oracle_statuses(issuer, pres):
    return [issuer.verify_presentation(p) for p in pres]


@pytest.fixture(scope="module")
def batches():
    params, key, ip, issuer, a = make_batch(4, "SSPE", [0, 3], 70, b"gpu-plans-a")
    _, _, _, _, b = make_batch(4, "SSPE", [1], 20, b"gpu-plans-a")        # same parameters and key, another shape
    corrupt(a, b"plans-a")
    corrupt(b, b"plans-b")
    return params, key, ip, issuer, a, b, oracle_statuses(issuer, a), oracle_statuses(issuer, b)


@pytest.mark.parametrize("selfcheck", [False, True])
def test_a_kept_plan_serves_other_counts_and_other_items(batches, selfcheck, monkeypatch):
    import aeonflux_amd as afx
    params, key, ip, issuer, a, b, want_a, want_b = batches
    if selfcheck:
        monkeypatch.setenv("AFX_PLAN_SELFCHECK", "1")
    else:
        monkeypatch.delenv("AFX_PLAN_SELFCHECK", raising=False)
    ctx = afx.Context(params, key, ip)
    assert any(want_a) and not all(want_a), "the batch must hold rejected items"
    # (offset, count): padded sizes 16, 16, 32, 64, 16 (kept plan, other items), 128 (70 items), 32 again, 1
    for off, cnt in [(0, 5), (5, 16), (3, 17), (10, 40), (50, 9), (0, 70), (40, 30), (69, 1)]:
        assert gpu_verify(afx, ctx, a[off:off + cnt]) == want_a[off:off + cnt], (off, cnt)
        if cnt % 2:   # another shape's plan in between, and back
            assert gpu_verify(afx, ctx, b[:cnt % 20 + 1]) == want_b[:cnt % 20 + 1]
    # the large-pass plan moves the workspace (ensure_ws grows it); the kept small plans must follow
    ctx.set_small_batch_items(0)
    assert gpu_verify(afx, ctx, a) == want_a
    ctx.set_small_batch_items(4096)
    assert gpu_verify(afx, ctx, a[7:30]) == want_a[7:30]
    assert gpu_verify(afx, ctx, b) == want_b
    ctx.close()


def test_kept_prover_plans_issue_the_oracles_bytes_call_after_call():
    """afx_issue on a kept plan: the second and third calls (other requests, another count under the same padded size) give the
    oracle's bytes; the secret mode is part of the plan's identity, so switching it between calls must not reuse the other mode's
    launch list"""
    import aeonflux_amd as afx
    from aeonflux_amd import batch
    n = 4
    d = make_credentials(n, "SSPE", 24, b"gpu-plans-issue")
    cr = d["creds"]
    col = lambda items, f: np.stack([np.frombuffer(f(c), np.uint8) for c in items])
    ctx = afx.Context(d["params"], d["key"], d["ip"])

    def issue(items):
        g = dict(kinds=items[0]["kinds"], values=np.stack([col(items, lambda c, i=i: c["values"][i][:32]) for i in range(n)]),
                 t_wide=col(items, lambda c: c["rnd"][0]), U_wide=col(items, lambda c: c["rnd"][1]), rng_seed=col(items, lambda c: c["rnd"][2]),
                 positions=np.arange(len(items), dtype=np.uint64))
        outs, status = batch.issue_mixed(ctx, [g])
        assert not status.any()
        o = outs[0]
        for i, c in enumerate(items):
            assert (o["t"][i].tobytes(), o["U"][i].tobytes(), o["V"][i].tobytes(), o["challenge"][i].tobytes()) == (c["t"], c["U"], c["V"], c["challenge"])
            assert [o["responses"][k, i].tobytes() for k in range(n + 5)] == c["responses"]
    for mode in (2, 0, 2, 1):
        ctx.set_secret_independent_addressing({0: False, 1: True, 2: "prover"}[mode])
        issue(cr[:7])
        issue(cr[7:20])
        issue(cr[20:])
        issue(cr[3:19])
    ctx.close()


def test_a_wide_request_takes_longer_chains_and_the_same_statuses(batches):
    """210 groups of one presentation each (one shape repeated: the library keeps the caller's groups apart) in one
    afx_verify_presentations_mixed request: the merged launches are 210 waves wide per grid row, so the constraint stage of every
    group's latency plan takes more than one term per chain (engine.cpp Assembler::msm); the statuses are the oracle's, at the
    positions given (reversed), and one call alone goes back to one chain per term."""
    import ctypes as C

    import aeonflux_amd as afx
    from aeonflux_amd import batch
    from tests.soa import presentation_arrays, shape_of
    params, key, ip, issuer, a, b, want_a, want_b = batches
    ctx = afx.Context(params, key, ip)
    sh = afx.Shape.from_buffer_copy(bytes(shape_of(a[0])))
    reps = 3
    total = reps * len(a)
    arr = (afx.PresentationGroup * total)()
    keep = []
    for g in range(total):
        cols = presentation_arrays([a[g % len(a)]])
        soa, encs = batch.presentation_soa(cols)
        pos = np.array([total - 1 - g], np.uint64)
        arr[g].shape, arr[g].batch, arr[g].count = sh, soa, 1
        arr[g].positions = pos.ctypes.data_as(C.POINTER(C.c_uint64))
        keep.append((cols, soa, encs, pos))
    # with the context's collector on (the default) the request's groups join the collecting session and groups of ONE shape share a pass's
    # item slots: 210 one-presentation groups are four passes of 64 - the statuses must be the same ...
    status = np.full(total, 255, np.uint8)
    afx.check(afx.lib().afx_verify_presentations_mixed(ctx.h, arr, total, status.ctypes.data, total))
    assert status.tolist() == (want_a * reps)[::-1]
    cs = ctx.coalescing_stats()
    assert cs["calls"] == total and cs["appended_calls"] >= total - 8 and cs["sessions"] <= 4, cs
    # ... as with the request's own session of 210 plans (collection off: the round-4 path), which is what this test is about
    ctx.set_coalescing(0, 0)
    status = np.full(total, 255, np.uint8)
    afx.check(afx.lib().afx_verify_presentations_mixed(ctx.h, arr, total, status.ctypes.data, total))
    assert status.tolist() == (want_a * reps)[::-1]
    wide_jobs = ctx.plan_stats()["msm_jobs"]
    assert gpu_verify(afx, ctx, a[:1]) == want_a[:1]
    assert ctx.plan_stats()["msm_jobs"] > wide_jobs, "one call alone: one chain per term"
    ctx.close()


def test_a_stream_of_junk_shapes_does_not_pin_the_plan_cache():
    """Shapes come from callers (a serialized batch names its own): 10 000 distinct valid-but-unusual shapes go through the cache of
    kept plans, which holds 512 / 64 MB and drops the least recently used.  Afterwards the shape a server really sees is kept again
    after one call, and a small call costs what it cost before (afx_ctx_get_plan_cache_stats)."""
    import time
    import aeonflux_amd as afx
    from aeonflux_amd import batch
    params, key, ip, issuer, real = make_batch(8, "SSPPEEEE", [4, 5, 6, 7], 4, b"gpu-plans-lru")
    want = oracle_statuses(issuer, real)
    ctx = afx.Context(params, key, ip)

    def small_call_ms(reps=30):
        ts = []
        for _ in range(reps):
            t0 = time.perf_counter()
            assert gpu_verify(afx, ctx, real) == want
            ts.append((time.perf_counter() - t0) * 1e3)
        return sorted(ts)[len(ts) // 2]

    assert gpu_verify(afx, ctx, real) == want and gpu_verify(afx, ctx, real) == want
    s0 = ctx.plan_cache_stats()
    assert s0["hits"] >= 1 and s0["misses"] >= 1 and s0["evictions"] == 0
    before_ms = small_call_ms()
    # junk: 8 attributes, every pattern of revealed scalars / revealed points (256), with 0, 1 or 2 proofs of encryption attached at arbitrary
    # positions (the reference verifies whatever is attached, presentation.rs:438-440): 256 * (1 + 8 + 31) distinct accepted shapes
    z = lambda *s: np.zeros(s, np.uint8)
    junk = 0
    enc_choices = [()] + [(i,) for i in range(8)] + [(i, j) for i in range(8) for j in range(8) if i != j][:31]
    for pattern in range(256):
        for encs in enc_choices:
            sh = afx.Shape()
            sh.n_attributes, sh.n_responses, sh.n_hidden_scalars, sh.n_enc_proofs = 8, 3, 0, len(encs)
            for i in range(8):
                sh.kinds[i] = 2 if (pattern >> i) & 1 else 0
            for e, idx in enumerate(encs):
                sh.enc_indices[e] = idx
            p = {"challenge": z(1, 32), "responses": z(3, 1, 32), "C_x_0": z(1, 32), "C_x_1": z(1, 32), "C_V": z(1, 32), "C_y": z(8, 1, 32), "attr_values": z(8, 1, 32),
                 "enc": [{f: (z(6, 1, 32) if f == "responses" else z(1, 32)) for f in batch.ENC_FIELDS} for _ in encs]}
            assert batch.verify_presentations(ctx, sh, p).tolist() == [1]      # (the identity commitments are rejected: zkp)
            junk += 1
    assert junk >= 10000
    s1 = ctx.plan_cache_stats()
    assert s1["misses"] - s0["misses"] >= junk and s1["evictions"] >= junk - 512 and s1["entries"] <= 512 and s1["bytes"] <= 64 << 20, s1
    # the real shape was dropped on the way: one call assembles and keeps it again, the next one reuses it
    assert gpu_verify(afx, ctx, real) == want
    s2 = ctx.plan_cache_stats()
    assert s2["misses"] == s1["misses"] + 1
    assert gpu_verify(afx, ctx, real) == want
    s3 = ctx.plan_cache_stats()
    assert s3["hits"] == s2["hits"] + 1 and s3["misses"] == s2["misses"]
    after_ms = small_call_ms()
    print("small call: %.3f ms before the junk, %.3f ms after; cache %s" % (before_ms, after_ms, s3))
    assert after_ms <= max(1.0, 1.25 * before_ms), (before_ms, after_ms)
    ctx.close()
