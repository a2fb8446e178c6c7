import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_sessionstart(session):
    """The tests load the in-tree libpixelnerf_hip.so; build it once if it is missing or stale
    (hipcc cross-compiles gfx950 without a GPU).  The product path itself never builds or falls back."""
    from pixelnerf_amd import _lib
    _lib.ensure_built()


@pytest.fixture(scope="session")
def repo_root():
    return ROOT
