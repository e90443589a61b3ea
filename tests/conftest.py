import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")


@pytest.fixture(scope="session")
def orc():
    """The CPU oracle (test infrastructure): builds oracle/libc3dgs_oracle.so with gcc on first use."""
    from oracle import oracle
    oracle.lib()
    return oracle


@pytest.fixture(scope="session")
def hip():
    """The product package, with the HIP library loaded; GPU tests fail loudly if it is missing."""
    import torch
    import c3dgs_amd
    from c3dgs_amd import _lib
    _lib.lib()
    assert torch.cuda.is_available(), "GPU test selected but no GPU is visible"
    return c3dgs_amd
