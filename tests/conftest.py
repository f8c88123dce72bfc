import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")


@pytest.fixture(scope="session", autouse=True)
def built_libraries():
    """The suites need the in-tree HIP library (ABI tests, GPU tests) and the C oracle.  Both are git-ignored build products:
    build them once per session when they are missing (hipcc cross-compiles gfx950 without a GPU)."""
    import subprocess
    if not os.path.exists(os.path.join(ROOT, "mocopci_amd", "libmocopci_hip.so")):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "mocopci_amd", "csrc"), "-s", "-j8"])
    if not os.path.exists(os.path.join(ROOT, "oracle", "libpointset_oracle.so")):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "-s"])
