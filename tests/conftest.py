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
