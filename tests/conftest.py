import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from oracle import agc_oracle
    agc_oracle.lib()
    return agc_oracle


@pytest.fixture(scope="session")
def hip_ctx():
    """One HIP context for the GPU tests; fails loudly when the extension is missing."""
    from agc_amd import capi
    ctx = capi.Context(0)
    yield ctx
    ctx.close()
