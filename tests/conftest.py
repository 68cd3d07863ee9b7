import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


# The differential guard (sunode_amd/_native.py NativeSolver) is ON in the product and exercised by tests/test_guard.py,
# which switches it on per test; everywhere else the suite compares with the oracle already, and the guard would
# compile the conservative partner of every test model on the GPU box.
os.environ.setdefault("SA_GUARD", "0")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")
