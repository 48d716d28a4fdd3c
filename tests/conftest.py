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


@pytest.fixture(scope="session")
def checksum_np():
    """The replica-check word of include/slamhip.h (slamhip_cs_maps_checksum / slamhip_hs_checksum), restated with NumPy:
    sum over i of mix64(i << 32 | element_i) mod 2^64, elements zero-extended from their own width."""
    import numpy as np

    def f(a):
        a = np.ascontiguousarray(a).ravel()
        u = a.view({1: np.uint8, 2: np.uint16, 4: np.uint32}[a.dtype.itemsize]).astype(np.uint64)
        with np.errstate(over="ignore"):
            x = (np.arange(u.size, dtype=np.uint64) << np.uint64(32)) | u
            x = x + np.uint64(0x9E3779B97F4A7C15)
            x = (x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
            x = (x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
            x = x ^ (x >> np.uint64(31))
            return int(np.add.reduce(x, dtype=np.uint64))
    return f
