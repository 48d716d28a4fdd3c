import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


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
