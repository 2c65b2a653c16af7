import json
import os
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG_DIR = os.path.join(REPO, "navigation-by-deja-vu_amd")
GOLDEN = os.path.join(REPO, "tests", "golden")
for p in (REPO, PKG_DIR):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def manifest():
    with open(os.path.join(GOLDEN, "manifest.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def golden():
    def load(name):
        return np.load(os.path.join(GOLDEN, name))
    return load


@pytest.fixture(autouse=True, scope="session")
def _steps_end_in_k_finish_where_possible():
    """The engine ends integer-path steps in k_finish only where it measured faster (>= 32768 views, <= 16 headings per
    agent); the test libraries are small, so the GPU suite asks for it wherever it is possible (DEJAVU_FINISH=2) unless
    a fixture says otherwise (the `eng` fixtures run every test in both forms).  Full-size tests meet the default rule
    either way.  Worker processes of multi-process tests inherit the variable."""
    before = os.environ.get("DEJAVU_FINISH")
    os.environ.setdefault("DEJAVU_FINISH", "2")
    yield
    if before is None:
        os.environ.pop("DEJAVU_FINISH", None)
