"""GPU: the library the other GPU tests load was built in the build container and shipped with the tree.  This test
compiles the product FROM SOURCE on the GPU box (a copy of aeonflux_amd/csrc + include in a scratch directory, the same
Makefile __graft_entry__.build() drives), loads that fresh library in a child process and checks a small batch against the
oracle - so what runs on the box is also what the box can build."""
import os
import shutil
import subprocess
import sys
import textwrap

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_built_on_this_box_from_a_clean_tree_matches_the_oracle(tmp_path):
    scratch = tmp_path / "tree"
    shutil.copytree(os.path.join(ROOT, "aeonflux_amd", "csrc"), scratch / "aeonflux_amd" / "csrc", ignore=shutil.ignore_patterns("build", "*.o", "*.so"))
    shutil.copytree(os.path.join(ROOT, "include"), scratch / "include")
    assert not (scratch / "aeonflux_amd" / "lib").exists()
    r = subprocess.run(["make", "-C", str(scratch / "aeonflux_amd" / "csrc"), "ARCH=gfx950"], capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stderr[-2000:]
    lib = scratch / "aeonflux_amd" / "lib" / "libaeonflux_gpu.so"
    assert lib.exists()
    child = textwrap.dedent("""
        import sys
        sys.path.insert(0, %r)
        import numpy as np
        import aeonflux_amd as afx
        afx.LIB_PATH = %r
        from tests.helpers import corrupt, gpu_verify, make_batch
        params, key, ip, issuer, pres = make_batch(4, "SSPE", [0, 3], 40, b"clean-build")
        corrupt(pres, b"clean-build-corrupt")
        want = [issuer.verify_presentation(p) for p in pres]
        ctx = afx.Context(params, key, ip)
        got = gpu_verify(afx, ctx, pres)
        assert got == want and 0 in want and 1 in want, (got, want)
        loaded = [l.split()[-1] for l in open("/proc/self/maps") if "libaeonflux_gpu.so" in l]
        assert loaded and all(p == %r for p in loaded), loaded
        print("clean build ok", sum(want), "rejected of", len(want))
    """) % (ROOT, str(lib), str(lib))
    c = subprocess.run([sys.executable, "-c", child], capture_output=True, text=True, timeout=600)
    assert c.returncode == 0 and "clean build ok" in c.stdout, (c.stdout[-1500:], c.stderr[-3000:])
