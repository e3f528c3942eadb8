import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_sessionstart(session):
    """the suites load the built C-ABI library; build it (hipcc cross-compiles without a GPU) when the tree is fresh"""
    lib = os.path.join(ROOT, "aeonflux_amd", "lib", "libaeonflux_gpu.so")
    if not os.path.exists(lib):
        import subprocess
        subprocess.run(["make", "-C", os.path.join(ROOT, "aeonflux_amd", "csrc"), "ARCH=gfx950"], check=False,
                       stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)


@pytest.fixture(scope="session")
def primitives():
    with open(os.path.join(GOLDEN, "primitives.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def kat():
    with open(os.path.join(GOLDEN, "kat.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def flows():
    with open(os.path.join(GOLDEN, "flows.json")) as f:
        return json.load(f)["flows"]
