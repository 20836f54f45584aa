import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# skm_cosine_csr routes tall outputs of at most 1024 columns (rows >= 8 x columns) to the cursor kernel and all others to
# the neighbour-list kernels; SKM_COSINE_PATH=lists / cursor force one of them (all exact).  Tests marked `cosine_paths` run once per
# routing: "default" is the product's own dispatch (no variable set), the other two cover the kernel the default
# would not pick at the test's size.  Unmarked tests run the default dispatch.
COSINE_PATHS = ("default", "lists", "cursor")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run by the driver with -m gpu)")
    config.addinivalue_line("markers", "cosine_paths: run under every routing of skm_cosine_csr (default / lists / cursor)")


def pytest_generate_tests(metafunc):
    if metafunc.definition.get_closest_marker("cosine_paths") and "_cosine_path" in metafunc.fixturenames:
        metafunc.parametrize("_cosine_path", COSINE_PATHS, indirect=True)


@pytest.fixture
def skm_option():
    """set(name, value): skm_set_option for the rest of the test (the library reads the SKM_* environment variables once,
    when it is loaded, so a test switches kernels through the API); everything goes back to the environment's values at
    teardown.  Unknown names / values raise: a test cannot silently run the default kernel instead of the one it names."""
    from snekmer_amd import _hip

    touched = []

    def set_(name, value):
        touched.append(name)
        _hip.set_option(name, value)

    yield set_
    for name in touched:
        _hip.set_option(name, None)


@pytest.fixture(autouse=True)
def _cosine_path(request):
    path = getattr(request, "param", "default")
    from snekmer_amd import _hip

    try:
        _hip.set_option("SKM_COSINE_PATH", None if path == "default" else path)
    except _hip.HipUnavailable:
        if path != "default":
            raise
        yield path
        return
    yield path
    _hip.set_option("SKM_COSINE_PATH", None)
