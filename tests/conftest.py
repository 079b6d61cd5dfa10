import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    # `-m gpu` on a box without a GPU must fail loudly rather than skip silently; plain runs on
    # the CPU container deselect via -m "not gpu".
    pass


@pytest.fixture(scope="session")
def load_image():
    from golden_graphs import INPUTS
    from pngio import read_png
    cache = {}

    def _load(name):
        if name not in cache:
            cache[name] = read_png(os.path.join(INPUTS, os.path.basename(name)))
        return cache[name]

    return _load
