"""GPU: the SEGMENTED chains of small prover passes give the bytes the whole chains give.

Under secret-independent addressing (the default on the prover side) a pass of up to 2048 items cuts every secret scalar on a
per-item base into 8 segments (4 above 256 items) over the base's powers - one k_powers doubling chain per base and pass - and the
proof's commitments multiply the bases an earlier stage computed part by part on the pass's inputs (engine.cpp Assembler::segments,
SchnorrBuilder::prove_compact).  The other GPU tests compare these calls with the oracle at their sizes; this one runs issue and
show at sizes on both sides of every threshold in this process (default: segments on) and in child processes with
AFX_SEGMENTS=1 (whole chains: the kernels of every larger pass), AFX_SEGMENTS=4, AFX_SEGMENTS=2 and with AFX_QUAD_CHAINS=0 (the
segments on the one-wave kernels, which a merged launch too wide for the four-wave ones takes), and compares a digest of every
byte the calls return."""
import hashlib
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SIZES = (1, 3, 70, 257, 300, 600, 1100)   # (8 segments up to 256 items, 4 above; cached tables up to 512; the one-wave kernels once a stage has more than 1024 blocks)


def digest(check_powers=False):
    import aeonflux_amd as afx
    from tests.helpers import make_credentials
    from tests.test_gpu_prove import gpu_issue, gpu_show
    h = hashlib.sha256()
    n, layout, hide = 6, "SPPEES", [0, 3, 4]
    d = make_credentials(n, layout, max(SIZES), b"gpu-segments")
    take, user = d["take"], d["user"]
    kinds = list(d["creds"][0]["kinds"])
    shown = [1 if (i in hide and k == 0) else 4 if i in hide else k for i, k in enumerate(kinds)]
    nsp = sum(1 for k in shown if k == 4)
    kps = [user.keypair_derive(take(64)) for _ in range(max(SIZES))]
    zw = [take(64) for _ in range(max(SIZES))]
    sd = [take(32) for _ in range(max(SIZES))]
    es = [take(32 * nsp) for _ in range(max(SIZES))]
    ictx = afx.Context(d["params"], d["key"], d["ip"])
    uctx = afx.Context(d["params"], None, d["ip"])
    for count in SIZES:
        cr = d["creds"][:count]
        vals = [[c["values"][i][:32] for c in cr] for i in range(n)]
        if check_powers:
            ictx.set_timing(True)
            uctx.set_timing(True)
        o, st = gpu_issue(afx, ictx, kinds, vals, [c["rnd"][0] for c in cr], [c["rnd"][1] for c in cr], [c["rnd"][2] for c in cr])
        assert not st.any()
        # (also the oracle's: the credentials were issued by it from the same randomness)
        assert bytes(o["V"][:32]) == cr[0]["V"] and bytes(o["challenge"][32 * (count - 1):32 * count]) == cr[-1]["challenge"]
        for f in ("t", "U", "V", "challenge", "responses"):
            h.update(o[f].tobytes())
        p, shape, st = gpu_show(afx, uctx, shown, cr, kps[:count], zw[:count], sd[:count], es[:count])
        assert not st.any()
        for f in ("challenge", "responses", "C_x_0", "C_x_1", "C_V", "C_y"):
            h.update(p[f].tobytes())
        for e in p["enc"]:
            for f in sorted(e):
                h.update(e[f].tobytes())
        if check_powers:
            for ctx in (ictx, uctx):
                ms, launches = ctx.get_timing("k_powers")
                assert launches >= 1, "the pass did not take the segmented chains"
                ctx.set_timing(False)
    ictx.close()
    uctx.close()
    return h.hexdigest()


@pytest.mark.gpu
def test_segmented_and_whole_chains_return_the_same_bytes():
    assert os.environ.get("AFX_SEGMENTS", "8") == "8" and os.environ.get("AFX_QUAD_CHAINS", "1") != "0", "this process is meant to run the default plan"
    here = digest(check_powers=True)
    # (... and with the transcripts on 32 lanes per item instead of a wave each: kernels.hip k_hash_coop / k_hash_coop64)
    for env in ({"AFX_SEGMENTS": "1"}, {"AFX_SEGMENTS": "4"}, {"AFX_SEGMENTS": "2"}, {"AFX_QUAD_CHAINS": "0"}, {"AFX_HASH_WAVE": "0"}):
        child = subprocess.run([sys.executable, "-c", "import tests.test_gpu_segments as t; print(t.digest())"], cwd=ROOT,
                               env=dict(os.environ, **env), capture_output=True, text=True, timeout=900)
        assert child.returncode == 0, (env, child.stderr[-2000:])
        assert child.stdout.strip().splitlines()[-1] == here, env
