import glob
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def golden_files(pattern="*"):
    return sorted(glob.glob(os.path.join(GOLDEN_DIR, pattern + ".npz")))


@pytest.fixture(scope="session")
def checkers():
    """Build (if needed) and return the oracle/ checker module."""
    from oracle import checker
    if not checker.have_oracle() or (not checker.have_ref()
                                     and os.path.isdir("/root/reference")):
        checker.build(ref=os.path.isdir("/root/reference"))
    return checker
