import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    # a GPU test that hangs (a kernel that never ends) must fail, not sit on the box until the caller's limit: pytest-timeout
    # (installed in this image) ends it after ten minutes; the whole GPU suite takes under two
    for item in items:
        if item.get_closest_marker("gpu") and not item.get_closest_marker("timeout"):
            item.add_marker(pytest.mark.timeout(600))


@pytest.fixture(scope="session")
def oc():
    import oracle_c
    oracle_c.lib()
    return oracle_c


@pytest.fixture(scope="session")
def npo():
    import np_oracle
    return np_oracle


@pytest.fixture(scope="session")
def sim():
    import slam.net_amd.sim as s
    return s
