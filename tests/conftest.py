"""pytest configuration: registers the `gpu` marker; `-m "not gpu"` runs on CPU only."""
import os
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

GOLDEN = ROOT/"tests"/"golden"
# code objects of run-time translated fragments stay inside the repository (build/ is git-ignored and travels to the GPU box)
os.environ.setdefault("SHADERFLOW_JIT_CACHE", str(ROOT/"build"/"jit"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    import numpy as np

    def load(name: str):
        return np.load(GOLDEN/f"{name}.npz")
    return load
