"""GPU parity for concurrent small calls on ONE context.  Issuer::verify, Issuer::issue and AnonymousCredential::show are `&self`
methods a server calls one item at a time from all its threads (/root/reference/src/issuer.rs:141-147, :111-124,
src/credential.rs:37-46); the engine collects such calls into shared launch sets and lets calls of one shape share a pass
(include/aeonflux_gpu.h afx_ctx_set_coalescing; plans.cpp coalesced_call).  32 threads x mixed shapes and operations against one
issuer context and one user context: every status and every issued / shown byte equals the ORACLE's, whoever shared a pass with
whom."""
import threading

import numpy as np
import pytest

from tests.helpers import gpu_verify, make_credentials
from tests.test_gpu_prove import gpu_issue, gpu_show

pytestmark = pytest.mark.gpu

N = 4
SEED = b"gpu-coalescing"
LAYOUTS = [("SSPE", [0, 3], 12), ("PPPP", [], 9), ("ESSE", [1, 3], 10), ("SSSS", [0, 1, 2, 3], 8), ("SPEE", [2, 3], 11)]


def co_stats(afx, ctx):
    return ctx.coalescing_stats()


@pytest.fixture(scope="module")
def world():
    """per layout: oracle-issued credentials, the oracle's presentations of them (some damaged) and the oracle's verdicts"""
    import oracle   # checker / input generator
    out = []
    for layout, hide, cnt in LAYOUTS:
        d = make_credentials(N, layout, cnt, SEED)
        take, user, issuer = d["take"], d["user"], d["issuer"]
        kinds = list(d["creds"][0]["kinds"])
        for i in hide:
            kinds[i] = 1 if kinds[i] == 0 else 4
        nsp = sum(1 for k in kinds if k == 4)
        kps = [user.keypair_derive(take(64)) for _ in range(cnt)]
        zw, sd, es = [take(64) for _ in range(cnt)], [take(32) for _ in range(cnt)], [take(32 * nsp) for _ in range(cnt)]
        shown = []
        for c, kp, z, s, e in zip(d["creds"], kps, zw, sd, es):
            st, p = user.show(kinds, c["values"], c["t"], c["U"], c["V"], kp, z, s, e)
            assert st == 0
            shown.append(p)
        damaged = []
        for i, p in enumerate(shown):   # every third presentation is damaged somewhere else; the oracle says what each is worth
            q = oracle.Presentation.from_buffer_copy(bytes(p))
            if i % 3 == 1:
                q.responses[i % q.n_responses][5] ^= 1 << (i % 8)
            damaged.append(q)
        verdicts = [issuer.verify_presentation(q) for q in damaged]
        assert 0 in verdicts and 1 in verdicts
        out.append(dict(d=d, show_kinds=kinds, nsp=nsp, kps=kps, zw=zw, sd=sd, es=es, shown=shown, damaged=damaged, verdicts=verdicts))
    assert all(w["d"]["params"] == out[0]["d"]["params"] and w["d"]["key"] == out[0]["d"]["key"] for w in out)
    return out


def check_issue(afx, ictx, w, lo, hi):
    cr = w["d"]["creds"][lo:hi]
    o, st = gpu_issue(afx, ictx, cr[0]["kinds"], [[c["values"][i][:32] for c in cr] for i in range(N)], [c["rnd"][0] for c in cr], [c["rnd"][1] for c in cr],
                      [c["rnd"][2] for c in cr])
    assert st.tolist() == [0] * len(cr)
    cnt = len(cr)
    for i, c in enumerate(cr):
        assert (bytes(o["t"][32 * i:32 * i + 32]), bytes(o["U"][32 * i:32 * i + 32]), bytes(o["V"][32 * i:32 * i + 32]), bytes(o["challenge"][32 * i:32 * i + 32])) == \
               (c["t"], c["U"], c["V"], c["challenge"])
        for k in range(N + 5):
            assert bytes(o["responses"][32 * (k * cnt + i):32 * (k * cnt + i) + 32]) == c["responses"][k], (i, k)


def check_show(afx, uctx, w, lo, hi):
    cr, cnt = w["d"]["creds"][lo:hi], hi - lo
    o, shape, st = gpu_show(afx, uctx, w["show_kinds"], cr, w["kps"][lo:hi], w["zw"][lo:hi], w["sd"][lo:hi], w["es"][lo:hi])
    assert st.tolist() == [0] * cnt
    for i, pr in enumerate(w["shown"][lo:hi]):
        cell = lambda a, k=0: bytes(a[32 * (k * cnt + i):32 * (k * cnt + i) + 32])
        assert cell(o["challenge"]) == bytes(pr.challenge)
        assert all(cell(o["responses"], k) == bytes(pr.responses[k]) for k in range(pr.n_responses))
        assert (cell(o["C_x_0"]), cell(o["C_x_1"]), cell(o["C_V"])) == (bytes(pr.C_x_0), bytes(pr.C_x_1), bytes(pr.C_V))
        assert all(cell(o["C_y"], k) == bytes(pr.C_y[k]) for k in range(N))
        for e in range(w["nsp"]):
            q, g = pr.enc[e], o["enc"][e]
            assert cell(g["challenge"]) == bytes(q.challenge) and all(cell(g["responses"], k) == bytes(q.responses[k]) for k in range(6))
            assert all(cell(g[f]) == bytes(getattr(q, f)) for f in ("pk", "E1", "E2", "C_y_1", "C_y_2", "C_y_3", "C_y_2p"))


def check_issuance_verify(afx, uctx, w, lo, hi):
    from aeonflux_amd import batch
    cr = w["d"]["creds"][lo:hi]
    col = lambda f: np.stack([np.frombuffer(f(c), np.uint8) for c in cr]).copy()
    iss = {k: col(lambda c, k=k: c[k]) for k in ("t", "U", "V", "challenge")}
    iss["responses"] = np.stack([col(lambda c, k=k: c["responses"][k]) for k in range(N + 5)])
    values = np.stack([col(lambda c, i=i: c["values"][i][:32]) for i in range(N)])
    want = []
    for i, c in enumerate(cr):
        if (lo + i) % 2:
            iss["responses"][2, i, 3] ^= 1
        want.append(w["d"]["user"].issuance_verify(c["kinds"], c["values"], c["t"], c["U"], c["V"], c["challenge"],
                                                   [iss["responses"][k, i].tobytes() for k in range(N + 5)]))
    assert batch.verify_issuances(uctx, cr[0]["kinds"], values, iss).tolist() == want and (len(cr) < 2 or 1 in want)


def drive(afx, world, ictx, uctx, threads, rounds):
    errs = []

    def work(t):
        try:
            for r in range(rounds):
                w = world[(t + r) % len(world)]
                total = len(w["shown"])
                lo = (3 * t + r) % total
                hi = min(total, lo + 1 + (t + r) % 3)    # 1 .. 3 items a call
                what = (t + 2 * r) % 6
                if what <= 1:     # Issuer::verify on column arrays
                    assert gpu_verify(afx, ictx, w["damaged"][lo:hi]) == w["verdicts"][lo:hi]
                elif what == 2:
                    check_issue(afx, ictx, w, lo, hi)
                elif what == 3:
                    check_show(afx, uctx, w, lo, hi)
                elif what == 4:   # Issuer::verify on a serialized batch: the records join the same sessions
                    from aeonflux_amd import wire
                    from tests.soa import presentation_arrays, shape_of
                    sub = w["damaged"][lo:hi]
                    blob = wire.pack_presentations(afx.Shape.from_buffer_copy(bytes(shape_of(sub[0]))), presentation_arrays(sub))
                    assert wire.verify_wire(ictx, blob).tolist() == w["verdicts"][lo:hi]
                else:             # CredentialIssuance::verify on the user's context, one response of every other item damaged
                    check_issuance_verify(afx, uctx, w, lo, hi)
        except BaseException as e:   # noqa: an assertion in a thread must fail the test
            errs.append((t, repr(e)[:400]))

    ths = [threading.Thread(target=work, args=(t,)) for t in range(threads)]
    for th in ths:
        th.start()
    for th in ths:
        th.join()
    return errs


def test_32_threads_of_mixed_shapes_and_operations_on_one_context(world):
    import aeonflux_amd as afx
    d0 = world[0]["d"]
    ictx, uctx = afx.Context(d0["params"], d0["key"], d0["ip"]), afx.Context(d0["params"], None, d0["ip"])
    errs = drive(afx, world, ictx, uctx, 32, 12)
    assert not errs, errs[:3]
    si, su = co_stats(afx, ictx), co_stats(afx, uctx)
    # the calls were collected: fewer launch sets than calls, calls in other calls' item slots, many calls in one set
    assert si["calls"] + su["calls"] == 32 * 12
    assert si["sessions"] < si["calls"] and si["appended_calls"] > 0 and si["max_calls"] >= 4, si
    assert su["sessions"] < su["calls"] and su["appended_calls"] > 0, su
    # the same drive with collection switched off: every call takes the context in turn, same bytes
    ictx.set_coalescing(0, 0)
    uctx.set_coalescing(0, 0)
    errs = drive(afx, world, ictx, uctx, 8, 6)
    assert not errs, errs[:3]
    assert co_stats(afx, ictx)["calls"] == si["calls"] and co_stats(afx, uctx)["calls"] == su["calls"]
    ictx.close()
    uctx.close()


def test_large_calls_and_setters_between_collected_calls(world):
    """a batch too large to be collected and the afx_ctx_set_* functions wait for the collections in flight, take the context for
    themselves, and the small calls go on around them"""
    import aeonflux_amd as afx
    w = world[0]
    d0 = w["d"]
    ictx, uctx = afx.Context(d0["params"], d0["key"], d0["ip"]), afx.Context(d0["params"], None, d0["ip"])
    stop, errs = threading.Event(), []

    def alone():
        try:
            big = w["damaged"] * 60          # 720 presentations: beyond the 512 a collected call may have
            want = w["verdicts"] * 60
            for r in range(4):
                assert gpu_verify(afx, ictx, big) == want
                ictx.set_small_batch_items(4096)
                ictx.plan_stats()
        except BaseException as e:   # noqa
            errs.append(("alone", repr(e)[:400]))
        finally:
            stop.set()

    th = threading.Thread(target=alone)
    th.start()
    rounds = 0
    while not stop.is_set() and rounds < 200:
        errs += drive(afx, world, ictx, uctx, 8, 2)
        rounds += 1
    th.join()
    assert not errs, errs[:3]
    assert rounds >= 1
    ictx.close()
    uctx.close()


def test_collected_calls_of_dozens_of_items_cross_the_pass_size_thresholds():
    """Calls of 40 ... 120 items from 10 threads: the passes they share hold a few hundred to over a thousand items, on both sides of
    where a small prover pass changes form (8 segments up to 256 items, 4 above; cached tables up to 512; the one-wave kernels beyond a
    launch's 1024 blocks - engine.cpp Assembler::segments, msm_list) - and of where a joined call no longer fits the item slots of the
    pass that is collecting.  Every issued and shown byte against the oracle, as above."""
    import aeonflux_amd as afx
    cnt = 130
    d = make_credentials(N, "SPES", cnt, SEED + b"-dozens")
    take, user = d["take"], d["user"]
    kinds = list(d["creds"][0]["kinds"])
    kinds[2] = 4
    kinds[3] = 1
    kps = [user.keypair_derive(take(64)) for _ in range(cnt)]
    zw, sd, es = [take(64) for _ in range(cnt)], [take(32) for _ in range(cnt)], [take(32) for _ in range(cnt)]
    shown = []
    for c, kp, z, s, e in zip(d["creds"], kps, zw, sd, es):
        st, p = user.show(kinds, c["values"], c["t"], c["U"], c["V"], kp, z, s, e)
        assert st == 0
        shown.append(p)
    w = dict(d=d, show_kinds=kinds, nsp=1, kps=kps, zw=zw, sd=sd, es=es, shown=shown)
    ictx, uctx = afx.Context(d["params"], d["key"], d["ip"]), afx.Context(d["params"], None, d["ip"])
    errs = []

    def work(t):
        try:
            for r in range(6):
                n = 40 + 16 * ((t + r) % 6)          # 40 ... 120 items a call
                lo = (7 * t + 11 * r) % (cnt - n + 1)
                (check_issue if (t + r) % 2 else check_show)(afx, ictx if (t + r) % 2 else uctx, w, lo, lo + n)
        except BaseException as e:   # noqa
            errs.append((t, repr(e)[:400]))

    ths = [threading.Thread(target=work, args=(t,)) for t in range(10)]
    for th in ths:
        th.start()
    for th in ths:
        th.join()
    assert not errs, errs[:3]
    si, su = co_stats(afx, ictx), co_stats(afx, uctx)
    assert si["calls"] + su["calls"] == 60 and si["sessions"] < si["calls"] and su["sessions"] < su["calls"], (si, su)
    assert max(si["items"], su["items"]) > 1000
    ictx.close()
    uctx.close()
