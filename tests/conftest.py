import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box with -m gpu)")


@pytest.fixture(scope="session")
def device():
    """One rt355 context on cuda:0 for the whole GPU session (fails loudly without a GPU)."""
    from rsraytracingv3_amd import Device
    dev = Device(0)
    yield dev
    dev.close()
