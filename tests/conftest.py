import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def engine_lib():
    """libsd_engine.so, built in-tree if missing (hipcc cross-compiles without a GPU)."""
    from stablediffusion_amd import _lib, build
    if not os.path.exists(_lib.LIB_PATH):
        build.build()
    return _lib.load()


def rel_l2(a, b):
    import torch
    a = a.detach().float().cpu()
    b = b.detach().float().cpu()
    return (torch.linalg.vector_norm(a - b) / torch.linalg.vector_norm(b).clamp_min(1e-12)).item()
