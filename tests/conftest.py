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
    config.addinivalue_line("markers", "allow_fallbacks(*kinds): this gpu test knowingly runs forms the HIP kernels do not cover")


@pytest.fixture(scope="session", autouse=True)
def _build_checkers():
    """The oracle .so files are git-ignored build products; make sure they exist (gcc, seconds)."""
    from oracle import raster_oracle
    raster_oracle.build()
    yield


@pytest.fixture(autouse=True)
def _strict_dispatch(request):
    """`-m gpu` tests run with csplat.native.STRICT: a product function that would leave the HIP path for a GPU tensor (dtype,
    layout, unsupported width) RAISES instead of silently composing torch ops.  A test that knowingly runs such a form (the
    reference-golden GNN fixtures at latent size 32, fp64 cross-checks) says so with `@pytest.mark.allow_fallbacks("shape", ...)`."""
    if request.node.get_closest_marker("gpu") is None:
        yield
        return
    from csplat import native
    old = native.STRICT
    native.STRICT = True
    m = request.node.get_closest_marker("allow_fallbacks")
    try:
        if m is not None:
            with native.allow_fallbacks(*m.args):
                yield
        else:
            yield
    finally:
        native.STRICT = old
