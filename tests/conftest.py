import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def hip_ctx():
    """One spp context for the whole GPU session. Fails loudly if the HIP library is missing."""
    from slam_plus_plus_amd import api
    ctx = api.Context(0, api.FLAG_PROFILE)
    yield ctx
    ctx.close()
