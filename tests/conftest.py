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
    """CPU oracle (oracle/liboracle.so, built on demand with gcc). Test infrastructure only."""
    import oracle_lib
    oracle_lib.build()
    return oracle_lib


@pytest.fixture(scope="session")
def hip():
    """The product library through its C ABI; fails loudly when the HIP build or the GPU is missing."""
    import torch
    from asif_amd import capi
    assert torch.cuda.is_available(), "gpu-marked test running without a GPU"
    capi.load()
    return capi
