import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def pkg():
    """the product package (directory arctic-renderer_amd/, module name arctic_renderer_amd)."""
    import __graft_entry__ as entry
    return entry.load_package()


@pytest.fixture(scope="session")
def oracle():
    """the CPU oracle binding (test infrastructure); builds liboracle.so on first use."""
    from oracle import oracle as O
    O.build()
    return O


@pytest.fixture(scope="session")
def hip(pkg):
    """the HIP renderer binding; skips (never falls back) when no GPU is present."""
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no HIP device")
    return pkg.renderer
