import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def engine():
    """One engine per session; fails loudly when the HIP library or the GPU is missing."""
    from dsurftomo_amd import build
    from dsurftomo_amd.engine import Engine
    build.build()
    e = Engine(0)
    yield e
    e.close()
