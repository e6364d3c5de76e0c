import os
import sys

import pytest

try:                      # torch bundles its own HIP runtime: it has to be the first one loaded into the process, or torch.cuda
    import torch  # noqa: F401   # finds no device once libbiodgpu.so has pulled in the system's libamdhip64 (tests that hand torch tensors over need both)
except ImportError:
    pass

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))            # bai_oracle.py: the checker, imported by tests only

GOLDEN = os.path.join(ROOT, "tests", "golden")
DATA = os.path.join(GOLDEN, "data")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def data_dir():
    return DATA


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
