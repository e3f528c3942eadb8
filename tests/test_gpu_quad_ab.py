"""GPU: the two ways a SMALL pass runs its chains give the same bytes.  On an idle device a launch of at most 512 blocks takes the
four-wave kernels (k_msm_quad, k_pointsum_quad: kernels.hip) - which is what every small-batch test of this suite therefore runs;
this test runs the same calls once more in a child process with AFX_QUAD_CHAINS=0 (one wave per item: k_msm / k_msm_rows /
k_pointsum, the kernels of every larger pass) and compares a digest of everything the calls return: issued credentials and
proofs (default secret mode and mode 0), statuses of a partly corrupted batch of presentations and the challenges recomputed for
it.  Both are also the oracle's (the other tests); here the point is that neither path goes unexercised at small sizes."""
import hashlib
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def digest():
    import aeonflux_amd as afx
    from aeonflux_amd import batch
    from tests.helpers import corrupt, gpu_verify, make_batch, make_credentials
    h = hashlib.sha256()
    n = 4
    d = make_credentials(n, "SSPE", 12, b"gpu-quad-ab-issue")
    cr = d["creds"]
    col = lambda items, f: np.stack([np.frombuffer(f(c), np.uint8) for c in items])
    ctx = afx.Context(d["params"], d["key"], d["ip"])
    g = dict(kinds=cr[0]["kinds"], values=np.stack([col(cr, lambda c, i=i: c["values"][i][:32]) for i in range(n)]),
             t_wide=col(cr, lambda c: c["rnd"][0]), U_wide=col(cr, lambda c: c["rnd"][1]), rng_seed=col(cr, lambda c: c["rnd"][2]),
             positions=np.arange(len(cr), dtype=np.uint64))
    for mode in ("prover", False):
        ctx.set_secret_independent_addressing(mode)
        outs, status = batch.issue_mixed(ctx, [g])
        h.update(status.tobytes())
        for f in ("t", "U", "V", "challenge", "responses"):
            h.update(np.ascontiguousarray(outs[0][f]).tobytes())
    ctx.close()
    params, key, ip, issuer, pres = make_batch(8, "SSPPEEEE", [4, 5, 6, 7], 20, b"gpu-quad-ab-verify")
    corrupt(pres, b"gpu-quad-ab-corrupt")
    ctx = afx.Context(params, key, ip)
    ctx.set_challenge_trace(5, len(pres))
    h.update(bytes(gpu_verify(afx, ctx, pres)))
    h.update(ctx.get_challenge_trace().tobytes())
    ctx.close()
    return h.hexdigest()


@pytest.mark.gpu
def test_four_wave_and_one_wave_chains_return_the_same_bytes():
    assert os.environ.get("AFX_QUAD_CHAINS", "1") != "0", "this process is meant to run the four-wave kernels"
    here = digest()
    # (second child: the transcripts of these small passes on 32 lanes per item, k_hash_coop, instead of a wave each, k_hash_coop64)
    for env in ({"AFX_QUAD_CHAINS": "0"}, {"AFX_HASH_WAVE": "0"}):
        child = subprocess.run([sys.executable, "-c", "import tests.test_gpu_quad_ab as t; print(t.digest())"], cwd=ROOT,
                               env=dict(os.environ, **env), capture_output=True, text=True, timeout=600)
        assert child.returncode == 0, (env, child.stderr[-2000:])
        assert child.stdout.strip().splitlines()[-1] == here, env
