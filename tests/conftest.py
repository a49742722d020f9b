import importlib
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
PKG = "retinanet-for-table-detection_amd"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box with -m gpu)")


@pytest.fixture(scope="session")
def pkg():
    return importlib.import_module(PKG)


@pytest.fixture(scope="session")
def golden():
    import numpy as np
    return np.load(os.path.join(ROOT, "tests", "golden", "ref_numpy_golden.npz"))


@pytest.fixture(scope="session")
def handle(pkg):
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    h = pkg.Handle(0)
    h.set_stream(torch.cuda.current_stream().cuda_stream)
    yield h
    torch.cuda.synchronize()
    h.close()
