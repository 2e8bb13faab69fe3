import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def hip_lib():
    """The C-ABI library; GPU tests fail loudly (no skip, no fallback) when it is missing."""
    from effocr_amd import _lib
    return _lib.lib()


@pytest.fixture(scope="session")
def dev():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a ROCm device"
    return torch.device("cuda:0")


@pytest.fixture
def ab_lib():
    """The A/B build (effocr_amd/libeffocr_hip_ab.so, `make AB=1`): product library + the kernels only A/B switches reach.  For the
    duration of the test every engine created loads it; the product library comes back afterwards."""
    from effocr_amd import _lib
    if not os.path.exists(_lib.SO_PATH_AB):
        _lib.build()
    prev = _lib.use_library(_lib.SO_PATH_AB)
    try:
        yield _lib.lib()
    finally:
        _lib.use_library(None if prev.endswith("libeffocr_hip.so") else prev)
