import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_sessionstart(session):
    """A fresh checkout has no built artefacts (they are git-ignored): build them once (hipcc cross-compiles
    without a GPU; the GPU box receives the prebuilt files with the snapshot)."""
    lib = os.path.join(ROOT, "uspace_amd", "libuspace_hip.so")
    orc = os.path.join(ROOT, "oracle", "_build", "liboracle_generic.so")
    if not (os.path.exists(lib) and os.path.exists(orc)):
        import __graft_entry__
        __graft_entry__.build()


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
