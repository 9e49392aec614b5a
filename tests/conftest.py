import os
import subprocess
import sys

import pytest

# host layer self-check: carried undistorted coordinates are re-derived on the host and compared bit for bit
os.environ.setdefault("ICG_HOST_CHECK", "1")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ic-gvins_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    """CPU restatement of the reference (oracle/liboracle.so) — the checker, never the thing under test."""
    import oracle_lib
    return oracle_lib.load()


@pytest.fixture(scope="session")
def hiplib():
    import icgvins
    return icgvins.load_library()
