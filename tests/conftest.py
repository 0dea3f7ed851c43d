import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for sub in ("riffusion-hobby_amd", "oracle", ""):
    path = os.path.join(ROOT, sub) if sub else ROOT
    if path not in sys.path:
        sys.path.insert(0, path)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def repo_root():
    return ROOT


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")


def pytest_sessionstart(session):
    """A fresh checkout has no librfx.so (it is git-ignored): compile it once, never fall back to anything else."""
    lib = os.path.join(ROOT, "riffusion-hobby_amd", "librfx.so")
    if not os.path.exists(lib) and os.path.exists(os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")):
        import __graft_entry__

        __graft_entry__.build()
