"""C oracle vs the committed golden fixtures (tests/golden/, produced by oracle/gen_golden.py from the
independent NumPy restatement).  CPU only.  The same fixtures gate the HIP path in test_gpu_*.py."""
import glob
import os

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    return np.load(os.path.join(GOLD, name))


@pytest.mark.parametrize("name", sorted(os.path.basename(p) for p in glob.glob(os.path.join(GOLD, "cs_distance_*.npz"))))
def test_distance_fixture(oc, name):
    g = load(name)
    d, bi, bd = oc.distance_batch_pxcs(g["pixels"], int(g["size"]), g["xy"], g["pxcs"])
    assert (d == g["dist"]).all()
    assert bi == int(g["best"]) and bd == int(g["dist"][bi])
    # the same through poses + deterministic trig (candidate 0 = base pose)
    oc.set_trig_mode(oc.TRIG_DET)
    try:
        bi2, pose, bd2, alld = oc.search(g["pixels"], int(g["size"]), float(g["scale"]), g["xy"], g["base"], g["offs"])
    finally:
        oc.set_trig_mode(oc.TRIG_LIBM)
    assert (alld == g["dist"]).all() and bi2 == bi


@pytest.mark.parametrize("name", ["cs_holemap_64_r90.npz", "cs_holemap_256_r360_hw2.npz"])
def test_holemap_fixture(oc, name):
    g = load(name)
    size = int(g["size"])
    pix = np.full(size * size, 32750, np.uint16)
    for i in range(g["xy"].shape[0]):
        n = oc.update_holemap_pxcs(pix, size, float(g["scale"]), g["xy"][i], g["pxcs"][i],
                                   float(g["hole_width"]), int(g["quality"]))
        assert n == int(g["counts"][i])
        if i == 0:
            assert (pix == g["after1"]).all()
    assert (pix == g["after_all"]).all()


def test_obstacle_fixture(oc):
    g = load("cs_obstacle_64_r360.npz")
    size = int(g["size"])
    pix = np.full((size, size), -5, np.int8)
    for i in range(g["xy"].shape[0]):
        oc.update_obstaclemap_pxcs(pix, size, g["xy"][i], g["pxcs"][i], int(g["max_hits"]))
        if i == 0:
            assert (pix == g["after1"]).all()
    assert (pix == g["after_all"]).all()


def test_hector_fixture(oc):
    g = load("hs_grid_200_r180.npz")
    side = int(g["side"])
    oc.set_trig_mode(oc.TRIG_DET)
    try:
        grid = oc.Grid(float(g["cell"]), side, side)
        for i in range(g["xy"].shape[0]):
            grid.update_by_scan(g["xy"][i], g["poses"][i])
        assert (grid.cells["update_index"] == g["upd"]).all()
        assert (grid.cells["value"] == g["value"]).all()
        for T, Hk, dk in ((1, "H1", "d1"), (4, "H4", "d4")):
            H, d = grid.hessian(g["match_xy"], g["est_map"], T)
            assert np.allclose(H, g[Hk], rtol=2e-5, atol=1e-5)
            assert np.allclose(d, g[dk], rtol=2e-5, atol=1e-4)
    finally:
        oc.set_trig_mode(oc.TRIG_LIBM)
