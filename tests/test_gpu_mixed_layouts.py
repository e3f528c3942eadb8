"""GPU parity for requests, credentials and issuances of DIFFERENT attribute layouts in one call (afx_issue_mixed,
afx_verify_issuances_mixed, afx_show_mixed and their afx_group_* forms).  The reference takes any request per call
(/root/reference/src/issuer.rs:111-124; kinds are per attribute, src/amacs.rs:168-179; `show` is per credential,
src/credential.rs:37-46), so a server's stream interleaves layouts.  Every output byte is compared with the ORACLE's, the
status bytes come back in the caller's order, and what the GPU showed verifies on the GPU (again as one mixed call)."""
import numpy as np
import pytest

from tests.helpers import make_credentials

pytestmark = pytest.mark.gpu

N = 4
SEED = b"gpu-mixed-layouts"
# (layout of the credential, positions hidden at show time, items)
LAYOUTS = [("SSPE", [0, 3], 5), ("PPPP", [], 3), ("ESSE", [1, 3], 4), ("SSSS", [0, 1, 2, 3], 2), ("SPEE", [2, 3], 6)]


def col(items, f):
    return np.stack([np.frombuffer(f(c), np.uint8) for c in items])


def interleaved_positions(counts):
    """group g's items at every len(counts)-th place of the caller's order while it has items left (round robin)"""
    pos = [[] for _ in counts]
    left = list(counts)
    at = 0
    while any(left):
        for g in range(len(counts)):
            if left[g]:
                pos[g].append(at)
                at += 1
                left[g] -= 1
    return [np.array(p, np.uint64) for p in pos], at


@pytest.fixture(scope="module")
def world():
    """credentials of every layout under ONE parameter set and issuer key (the oracle made them: inputs and expected outputs)"""
    ds = [make_credentials(N, layout, cnt, SEED) for layout, _, cnt in LAYOUTS]
    assert all(d["params"] == ds[0]["params"] and d["key"] == ds[0]["key"] for d in ds)
    return ds


def issue_items(ds, pos):
    items = []
    for d, p in zip(ds, pos):
        cr = d["creds"]
        items.append(dict(kinds=cr[0]["kinds"], values=np.stack([col(cr, lambda c, i=i: c["values"][i][:32]) for i in range(N)]),
                          t_wide=col(cr, lambda c: c["rnd"][0]), U_wide=col(cr, lambda c: c["rnd"][1]), rng_seed=col(cr, lambda c: c["rnd"][2]), positions=p))
    return items


@pytest.mark.parametrize("grouped", [False, True])
def test_issue_mixed_matches_oracle_bytes_in_caller_order(world, grouped):
    import aeonflux_amd as afx
    from aeonflux_amd import batch
    ds = world
    pos, total = interleaved_positions([len(d["creds"]) for d in ds] + [2])
    ctx = afx.Group(ds[0]["params"], ds[0]["key"], ds[0]["ip"], [0, 0]) if grouped else afx.Context(ds[0]["params"], ds[0]["key"], ds[0]["ip"])
    items = issue_items(ds, pos[:-1])
    # a request with the wrong number of attributes beside them: MacCreation for its items only (amacs.rs:285-287)
    short = make_credentials(3, "SSP", 2, b"gpu-mixed-short")["creds"]
    items.append(dict(kinds=short[0]["kinds"], values=np.stack([col(short, lambda c, i=i: c["values"][i][:32]) for i in range(3)]),
                      t_wide=col(short, lambda c: c["rnd"][0]), U_wide=col(short, lambda c: c["rnd"][1]), rng_seed=col(short, lambda c: c["rnd"][2]), positions=pos[-1]))
    outs, status = batch.issue_mixed(ctx, items)
    want = np.zeros(total, np.uint8)
    want[pos[-1]] = afx.ST_MAC_CREATION
    assert status.tolist() == want.tolist()
    for d, o in zip(ds, outs):
        for i, c in enumerate(d["creds"]):
            assert (o["t"][i].tobytes(), o["U"][i].tobytes(), o["V"][i].tobytes(), o["challenge"][i].tobytes()) == (c["t"], c["U"], c["V"], c["challenge"])
            assert [o["responses"][k, i].tobytes() for k in range(N + 5)] == c["responses"]
    ctx.close()


@pytest.mark.parametrize("grouped", [False, True])
def test_verify_issuances_mixed_statuses_equal_the_oracles(world, grouped):
    import aeonflux_amd as afx
    from aeonflux_amd import batch
    ds = world
    pos, total = interleaved_positions([len(d["creds"]) for d in ds])
    ctx = afx.Group(ds[0]["params"], None, ds[0]["ip"], [0, 0]) if grouped else afx.Context(ds[0]["params"], None, ds[0]["ip"])
    items, want = [], np.full(total, 255, np.uint8)
    for g, (d, p) in enumerate(zip(ds, pos)):
        cr = d["creds"]
        iss = {k: col(cr, lambda c, k=k: c[k]).copy() for k in ("t", "U", "V", "challenge")}
        iss["responses"] = np.stack([col(cr, lambda c, k=k: c["responses"][k]) for k in range(N + 5)]).copy()
        vals = np.stack([col(cr, lambda c, i=i: c["values"][i][:32]) for i in range(N)]).copy()
        # tamper one item per group, somewhere else each time
        victim = g % len(cr)
        if g % 3 == 0:
            iss["V"][victim, 1] ^= 4
        elif g % 3 == 1:
            iss["responses"][g % (N + 5), victim, 7] ^= 1
        else:
            vals[g % N, victim, 2] ^= 8
        for i, c in enumerate(cr):
            v = [vals[k, i].tobytes() + c["values"][k][32:] for k in range(N)]
            want[p[i]] = d["user"].issuance_verify(c["kinds"], v, iss["t"][i].tobytes(), iss["U"][i].tobytes(), iss["V"][i].tobytes(),
                                                   iss["challenge"][i].tobytes(), [iss["responses"][k, i].tobytes() for k in range(N + 5)])
        items.append(dict(kinds=cr[0]["kinds"], values=vals, issuance=iss, positions=p))
    # one group whose proofs have the wrong number of responses: zkp rejects every one of them
    cr = ds[0]["creds"]
    extra = np.arange(total, total + len(cr), dtype=np.uint64)
    iss = {k: col(cr, lambda c, k=k: c[k]) for k in ("t", "U", "V", "challenge")}
    iss["responses"] = np.stack([col(cr, lambda c, k=k: c["responses"][k]) for k in range(N + 4)])
    items.append(dict(kinds=cr[0]["kinds"], values=np.stack([col(cr, lambda c, i=i: c["values"][i][:32]) for i in range(N)]), issuance=iss, positions=extra))
    status = batch.verify_issuances_mixed(ctx, items)
    assert status[:total].tolist() == want.tolist() and 1 in want.tolist() and 0 in want.tolist()
    assert status[total:].tolist() == [1] * len(cr)
    ctx.close()


@pytest.mark.parametrize("grouped", [False, True])
def test_show_mixed_matches_oracle_bytes_and_verifies_as_one_mixed_call(world, grouped):
    import aeonflux_amd as afx
    from aeonflux_amd import batch
    ds = world
    pos, total = interleaved_positions([len(d["creds"]) for d in ds] + [2])
    uctx = afx.Group(ds[0]["params"], None, ds[0]["ip"], [0, 0]) if grouped else afx.Context(ds[0]["params"], None, ds[0]["ip"])
    items, wants = [], []
    for (layout, hide, cnt), d, p in zip(LAYOUTS, ds, pos):
        cr, take, user = d["creds"], d["take"], d["user"]
        kinds = list(cr[0]["kinds"])
        for i in hide:
            kinds[i] = 1 if kinds[i] == 0 else 4
        nsp = sum(1 for k in kinds if k == 4)
        kps = [user.keypair_derive(take(64)) for _ in range(cnt)]
        zw, sd, es = [take(64) for _ in range(cnt)], [take(32) for _ in range(cnt)], [take(32 * nsp) for _ in range(cnt)]
        want = []
        for c, kp, z, s, e in zip(cr, kps, zw, sd, es):
            st, pr = user.show(kinds, c["values"], c["t"], c["U"], c["V"], kp, z, s, e)
            assert st == 0
            want.append(pr)
        wants.append((kinds, nsp, want))
        items.append(dict(kinds=kinds, values=np.stack([col(cr, lambda c, i=i: c["values"][i][:32]) for i in range(N)]),
                          M2=np.stack([col(cr, lambda c, i=i: c["values"][i][32:64]) for i in range(N)]),
                          m3=np.stack([col(cr, lambda c, i=i: c["values"][i][64:96]) for i in range(N)]),
                          t=col(cr, lambda c: c["t"]), U=col(cr, lambda c: c["U"]), V=col(cr, lambda c: c["V"]),
                          keypairs={f: np.stack([np.frombuffer(k[32 * j:32 * j + 32], np.uint8) for k in kps]) for j, f in enumerate(("a", "a0", "a1", "pk"))},
                          z_wide=np.stack([np.frombuffer(z, np.uint8) for z in zw]), rng_seed=np.stack([np.frombuffer(s, np.uint8) for s in sd]),
                          enc_seeds=np.stack([np.stack([np.frombuffer(e[32 * j:32 * j + 32], np.uint8) for e in es]) for j in range(nsp)]) if nsp else None,
                          positions=p))
    # credentials with a hidden group element and no symmetric key beside them: NoSymmetricKey for those only (presentation.rs:150-157)
    cr = ds[0]["creds"][:2]
    items.append(dict(kinds=[0, 0, 2, 4], values=np.stack([col(cr, lambda c, i=i: c["values"][i][:32]) for i in range(N)]),
                      M2=np.stack([col(cr, lambda c, i=i: c["values"][i][32:64]) for i in range(N)]),
                      m3=np.stack([col(cr, lambda c, i=i: c["values"][i][64:96]) for i in range(N)]),
                      t=col(cr, lambda c: c["t"]), U=col(cr, lambda c: c["U"]), V=col(cr, lambda c: c["V"]), keypairs=None,
                      z_wide=np.zeros((2, 64), np.uint8), rng_seed=np.zeros((2, 32), np.uint8), enc_seeds=np.zeros((1, 2, 32), np.uint8), positions=pos[-1]))
    outs, status = batch.show_mixed(uctx, items)
    want_status = np.zeros(total, np.uint8)
    want_status[pos[-1]] = afx.ST_NO_SYMMETRIC_KEY
    assert status.tolist() == want_status.tolist()
    for (kinds, nsp, want), (o, shape) in zip(wants, outs):
        p0 = want[0]
        assert (shape.n_attributes, shape.n_responses, shape.n_hidden_scalars, shape.n_enc_proofs) == (N, p0.n_responses, p0.n_hidden_scalars, p0.n_enc_proofs)
        assert list(shape.kinds[:N]) == list(p0.kinds[:N]) and list(shape.enc_indices[:nsp]) == [p0.enc[e].index for e in range(nsp)]
        for i, pr in enumerate(want):
            assert o["challenge"][i].tobytes() == bytes(pr.challenge)
            assert all(o["responses"][k, i].tobytes() == bytes(pr.responses[k]) for k in range(pr.n_responses))
            assert (o["C_x_0"][i].tobytes(), o["C_x_1"][i].tobytes(), o["C_V"][i].tobytes()) == (bytes(pr.C_x_0), bytes(pr.C_x_1), bytes(pr.C_V))
            for k in range(N):
                assert o["C_y"][k, i].tobytes() == bytes(pr.C_y[k])
                if pr.kinds[k] in (0, 2):
                    assert o["attr_values"][k, i].tobytes() == bytes(pr.attr_values[k])
            for e in range(nsp):
                q, g = pr.enc[e], o["enc"][e]
                assert g["challenge"][i].tobytes() == bytes(q.challenge)
                assert all(g["responses"][k, i].tobytes() == bytes(q.responses[k]) for k in range(6))
                for f in ("pk", "E1", "E2", "C_y_1", "C_y_2", "C_y_3", "C_y_2p"):
                    assert g[f][i].tobytes() == bytes(getattr(q, f)), f
    uctx.close()
    # the issuer verifies all of it in one mixed call; one item of every group is damaged first
    ictx = afx.Context(ds[0]["params"], ds[0]["key"], ds[0]["ip"])
    issuer = ds[0]["issuer"]
    vitems, vwant = [], []
    for g, ((kinds, nsp, want), (o, shape)) in enumerate(zip(wants, outs[:-1])):
        o = {k: (v.copy() if isinstance(v, np.ndarray) else [{f: a.copy() for f, a in e.items()} for e in v]) for k, v in o.items()}
        o["C_V"][g % len(want), 3] ^= 1
        want[g % len(want)].C_V[3] ^= 1   # the oracle's twin gets the same damage
        vitems.append((shape, o))
        vwant.append([issuer.verify_presentation(pr) for pr in want])
        assert vwant[-1][g % len(want)] == 1 and sum(vwant[-1]) == 1
    got = batch.verify_mixed(ictx, vitems)
    assert [s.tolist() for s in got] == vwant
    ictx.close()
