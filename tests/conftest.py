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
def oracle():
    import oracle_lib
    oracle_lib.lib()
    return oracle_lib


@pytest.fixture(scope="session")
def native_lib():
    """libmofreak_hip.so, built in-tree if missing/stale (hipcc cross-compiles gfx950 without a GPU)."""
    from mofreak_amd import build, api
    build.build_native()
    return api.load()


@pytest.fixture(scope="session")
def gpu_ctx(native_lib):
    """A device context; GPU tests fail loudly (no skip, no fallback) when there is no GPU."""
    import mofreak_amd as M
    ctx = M.Context(0)
    yield ctx
    ctx.close()
