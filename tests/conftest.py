import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    import util
    return util.Impl("oracle")


@pytest.fixture(scope="session")
def hip():
    import torch
    import util
    if not torch.cuda.is_available():
        pytest.fail("gpu test selected but no GPU is visible")
    return util.Impl("hip")


@pytest.fixture()
def oracle_backend():
    """Route the package's host layer (mantaflow_amd.core / plugins) through the oracle library (CPU tensors)."""
    import util
    from mantaflow_amd import _lib
    _lib.use_library(util.build_oracle(), "cpu")
    yield
    _lib.reset()


@pytest.fixture()
def hip_backend():
    import torch
    from mantaflow_amd import _lib
    if not torch.cuda.is_available():
        pytest.fail("gpu test selected but no GPU is visible")
    _lib.reset()
    lib = _lib.get()
    assert lib.backend == "hip"
    yield
    _lib.reset()
