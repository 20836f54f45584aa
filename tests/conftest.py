import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run by the driver with -m gpu)")


@pytest.fixture(autouse=True, scope="session")
def _list_path_for_small_outputs():
    """Outputs of at most 1024 columns take the cursor kernel by default (the apply case).  Most parity tests
    are that small, so the suite pins the neighbour-list path (the one the benchmarked sizes run); the tests
    that exercise the cursor kernel select it explicitly."""
    old = os.environ.get("SKM_COSINE_PATH")
    os.environ["SKM_COSINE_PATH"] = "lists"
    yield
    if old is None:
        os.environ.pop("SKM_COSINE_PATH", None)
    else:
        os.environ["SKM_COSINE_PATH"] = old
