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
    from oracle import pyoracle

    pyoracle.lib()
    return pyoracle


@pytest.fixture(scope="session")
def product():
    """The HIP library through its C ABI; fails loudly when it is not built or no GPU is visible."""
    import srrg2_slam_interfaces_amd as pkg
    from srrg2_slam_interfaces_amd import _capi

    _capi.lib()
    assert _capi.device_count() > 0, "no HIP device visible: GPU tests cannot run"
    return pkg
