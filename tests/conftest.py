import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box with -m gpu)")


@pytest.fixture(scope="session")
def orc():
    import oracle
    oracle.lib()
    return oracle


@pytest.fixture(scope="session")
def ibl():
    """The HIP library bound to cuda:0.  Fails loudly (no skip, no fallback) when
    the extension or the GPU is missing."""
    import vulkan_renderer_amd as vra
    return vra.Ibl(0)
