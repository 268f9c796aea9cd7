import os
import sys

import pytest

# torch BEFORE the product library: the PyTorch wheel carries its own copy of the HIP runtime (ROCm 7.0) and the product
# library is linked against the system's (ROCm 7.2), both under the soname libamdhip64.so.7.  Whichever is loaded first
# serves both; torch on top of the system's copy reports "No HIP GPUs are available" (seen when a test module that uses
# torch for device memory ran after modules that had already loaded the library).  The library on top of torch's copy works.
try:
    import torch  # noqa: F401
except Exception:  # (the CPU-only suite does not need it)
    pass

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
