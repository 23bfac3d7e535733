import os
import sys

import pytest

try:                      # PyTorch bundles its own HIP runtime: when a process uses both, torch must be loaded first
    import torch          # noqa: F401  (the product itself does not need torch; bench.py / megagta_amd.dist do)
except Exception:         # pragma: no cover
    torch = None

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (test infrastructure).  Built on demand with g++."""
    from oracle import oracle as O
    O.build()
    O.lib()
    return O
