import json
import os
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)
GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _torch_hip_runtime_first():
    """PyTorch ships its own HIP runtime; librbf_hip.so links the system one.  Both can live in one process as long as
    torch's initialises FIRST (bench.py does the same) -- the other order leaves torch with 'No HIP GPUs are available'.
    No-op on the CPU-only box."""
    try:
        import torch
        if torch.cuda.is_available():
            torch.cuda.init()
    except Exception:
        pass
    yield


def load_json(name):
    with open(os.path.join(GOLDEN, name)) as f:
        return json.load(f)


def load_npz(name):
    return np.load(os.path.join(GOLDEN, name))


@pytest.fixture(scope="session")
def oracle():
    """CPU oracle (test infrastructure): scalar C restatement + numpy host code."""
    from oracle import oracle as orc
    orc.build()
    return orc


def has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False
