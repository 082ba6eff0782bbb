import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_sessionstart(session):
    # One HIP runtime per process: torch must bring its own up before liblsdhip.so loads the system one, otherwise a
    # later torch.cuda call (the band tests use torch tensors as halo buffers) finds "no HIP GPUs".
    try:
        import torch
        if torch.cuda.is_available():
            torch.cuda.init()
    except Exception:
        pass


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from oracle import pyoracle
    pyoracle.build()
    return pyoracle
