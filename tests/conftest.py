import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DROPIN = os.path.join(ROOT, "cloth-splatting_amd")
for p in (DROPIN, ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _build_checkers():
    """The oracle .so files are git-ignored build products; make sure they exist (gcc, seconds)."""
    from oracle import raster_oracle
    raster_oracle.build()
    yield
