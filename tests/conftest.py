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
def oracle_mod():
    from oracle import oracle as O
    O.build()
    return O


@pytest.fixture(scope="session")
def cuda_device():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("this test is marked gpu and needs a GPU; none is visible")
    from gaussiancity_amd import _native
    _native.lib()  # fail loudly if libgcr_hip.so is not built
    from gaussiancity_amd import ext
    ext.poison_outputs = True  # gcr_backward must write every gradient element itself (include/gcr.h)
    return torch.device("cuda:0")
