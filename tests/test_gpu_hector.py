"""GPU parity tests for the HectorSLAM hot path (K4 matcher, K5 grid update) vs the CPU oracle.
Grid cells (fp32 log-odds + update indices) must be bit-exact; matcher outputs are floating point
whose summation order differs from the reference's thread chunks, so poses are compared within the
north-star tolerance of 1e-4 m / 1e-4 rad."""
import math
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
POS_TOL = 1e-4      # metres
ANG_TOL = 1e-4      # radians


@pytest.fixture(scope="module")
def hs_mod():
    import slam.net_amd.hector as m
    return m


@pytest.fixture(scope="module")
def ctx(hs_mod):
    c = hs_mod.Context(0)
    yield c
    c.close()


@pytest.fixture(scope="module")
def det(oc):
    oc.set_trig_mode(oc.TRIG_DET)
    yield oc
    oc.set_trig_mode(oc.TRIG_LIBM)


def cells_equal(got, ref_cells):
    return (got["update_index"] == ref_cells["update_index"]).all() and (got["value"] == ref_cells["value"]).all()


def test_grid_golden(hs_mod, ctx):
    g = np.load(os.path.join(GOLD, "hs_grid_200_r180.npz"))
    side = int(g["side"])
    rep = hs_mod.MapRepMultiMap(float(g["cell"]), (side, side), 1, ctx=ctx)
    for i in range(g["xy"].shape[0]):
        rep.UpdateByScan(hs_mod.ScanCloud(g["xy"][i]), g["poses"][i])
    cells = rep.Maps[0].GetCells()
    assert (cells["update_index"] == g["upd"]).all()
    assert (cells["value"] == g["value"]).all()
    rep.set_scan(hs_mod.ScanCloud(g["match_xy"]))
    H, d = rep.Maps[0].Hessian(g["est_map"])
    for Hk, dk in (("H1", "d1"), ("H4", "d4")):
        assert np.allclose(H, g[Hk], rtol=1e-4, atol=1e-4) and np.allclose(d, g[dk], rtol=1e-4, atol=1e-3)
    rep.close()


@pytest.mark.parametrize("side,cell,levels,R", [(400, 0.1, 4, 400), (2048, 40.0 / 2048, 3, 1080), (301, 0.13, 2, 360),
                                                (512, 0.08, 3, 3500)])     # more lines than the cell kernel keeps in LDS
def test_grid_update_vs_oracle(hs_mod, ctx, det, sim, side, cell, levels, R):
    oc = det
    segs = sim.default_field()
    rep = hs_mod.MapRepMultiMap(cell, (side, side), levels, ctx=ctx)
    ref = oc.make_pyramid(cell, side, side, levels)
    for l in range(levels):
        assert rep.Maps[l].Dimensions == (ref[l].w, ref[l].h)
        assert rep.Maps[l].CellLength == ref[l].cell_len
    rng = sim.PCG32(side)
    for it in range(6):
        p = np.array([20 + 0.25 * it, 20 - 0.1 * it, 0.15 * it], np.float32)
        rays, xy = sim.make_scan(segs, p, R, rng)
        rep.UpdateByScan(hs_mod.ScanCloud(xy), p)
        for l in range(levels):
            ref[l].update_by_scan(xy, p)
    for l in range(levels):
        got = rep.Maps[l].GetCells()
        assert cells_equal(got, ref[l].cells), l
        assert (rep.Maps[l].GetBitmapData() == ref[l].bitmap()).all()
        assert rep.Maps[l].GetMapExtends() == ref[l].map_extends()
    # probabilities (expf on the device vs libm expf: <= 1 ulp)
    idx = np.flatnonzero(ref[0].cells["value"] != 0)[:500].astype(np.int32)
    pg = rep.Maps[0].GetCachedProbability(idx)
    pr = np.array([ref[0].prob(i) for i in idx], np.float32)
    assert np.allclose(pg, pr, rtol=0, atol=2e-7)
    # scan origin offset, factors, degenerate points, upload/download round trip
    rep.SetUpdateFactorFree(0.3); rep.SetUpdateFactorOccupied(0.8)
    for g in ref:
        g.set_factors(0.3, 0.8)
    xy = np.array([[0.0, 0.0], [0.01, 0.0], [500.0, 0.0], [np.nan, 1.0], [3.0, 4.0], [3.0, 4.0], [-6.0, 2.5]], np.float32)
    rep.UpdateByScan(hs_mod.ScanCloud(xy, (0.5, -0.25, 0.0)), [20.0, 20.0, 0.7])
    for l in range(levels):
        ref[l].update_by_scan(xy, [20.0, 20.0, 0.7], origin=(0.5, -0.25))
        assert cells_equal(rep.Maps[l].GetCells(), ref[l].cells), l
    cells = rep.Maps[levels - 1].GetCells().copy()
    rep.Reset()
    assert (rep.Maps[0].GetCells()["value"] == 0).all() and (rep.Maps[0].GetCells()["update_index"] == -1).all()
    rep.Maps[levels - 1].SetCells(cells)
    assert cells_equal(rep.Maps[levels - 1].GetCells(), cells)
    rep.close()


def test_map_extends_quirks(hs_mod, ctx, det):
    """GetMapExtends on the device: empty map, NaN cells, and the reference's 10000 start value for the minima."""
    oc = det
    rep = hs_mod.MapRepMultiMap(1.0, (10300, 8), 2, ctx=ctx)
    ref = oc.make_pyramid(1.0, 10300, 8, 2)
    for l in range(2):
        assert rep.Maps[l].GetMapExtends() == ref[l].map_extends() == (False, 0, 0, 0, 0)
    w = 10300
    for cells_at in ([(10100, 2, 1.0)], [(10000, 1, -2.0)], [(9, 7, np.nan)], [(0, 0, 0.5)]):
        for x, y, v in cells_at:
            ref[0].cells["value"][y * w + x] = v
        rep.Maps[0].SetCells(ref[0].cells)
        assert rep.Maps[0].GetMapExtends() == ref[0].map_extends()
    assert rep.Maps[0].GetMapExtends() == (True, 10100, 7, 0, 0)
    assert rep.Maps[0].GetCell(10100, 2)["value"] == 1.0 and rep.Maps[0].GetCell(7 * w + 9)["value"] != rep.Maps[0].GetCell(7 * w + 9)["value"]
    rep.close()


def test_grid_update_order_dependence(hs_mod, ctx, det):
    """Free-then-occupied vs occupied-then-free within one scan (OccGridMap.cs:192-218)."""
    oc = det
    for xy in (np.array([[8.0, 0.0], [5.0, 0.0]], np.float32), np.array([[5.0, 0.0], [8.0, 0.0]], np.float32)):
        rep = hs_mod.MapRepMultiMap(1.0, (32, 32), 1, ctx=ctx)
        ref = oc.Grid(1.0, 32, 32)
        c0 = ref.cells.copy(); c0["value"] = np.linspace(-3, 3, 1024).astype(np.float32)
        ref.cells[:] = c0
        rep.Maps[0].SetCells(c0)
        for _ in range(3):
            rep.UpdateByScan(hs_mod.ScanCloud(xy), [10.0, 10.0, 0.0])
            ref.update_by_scan(xy, [10.0, 10.0, 0.0])
            assert cells_equal(rep.Maps[0].GetCells(), ref.cells)
        rep.close()


def test_grid_update_unordered_dense_scan(hs_mod, ctx, det):
    """Lines in random order, many per cell: the first free / first occupied line of a cell is the smallest index."""
    oc = det
    rep = hs_mod.MapRepMultiMap(0.2, (200, 200), 2, ctx=ctx)
    ref = oc.make_pyramid(0.2, 200, 200, 2)
    rng = np.random.default_rng(5)
    for it in range(4):
        ang = rng.uniform(-np.pi, np.pi, 2500)
        rad = rng.uniform(0.3, 15.0, 2500)
        xy = np.stack([rad * np.cos(ang), rad * np.sin(ang)], 1).astype(np.float32)
        p = [20.0 + 0.3 * it, 20.0, 0.2 * it]
        rep.UpdateByScan(hs_mod.ScanCloud(xy), p)
        for l in range(2):
            ref[l].update_by_scan(xy, p)
            assert cells_equal(rep.Maps[l].GetCells(), ref[l].cells), (it, l)
    rep.close()


def build_pair(hs_mod, ctx, oc, sim, side, cell, levels, R, n_scans):
    segs = sim.default_field()
    rep = hs_mod.MapRepMultiMap(cell, (side, side), levels, ctx=ctx)
    ref = oc.make_pyramid(cell, side, side, levels)
    rng = sim.PCG32(3)
    for it in range(n_scans):
        p = np.array([20 + 0.05 * it, 20 + 0.02 * it, 0.01 * it], np.float32)
        rays, xy = sim.make_scan(segs, p, R, rng)
        rep.UpdateByScan(hs_mod.ScanCloud(xy), p)
        for g in ref:
            g.update_by_scan(xy, p)
    return rep, ref, segs, rng


@pytest.mark.parametrize("side,cell,levels,R,iters", [(400, 0.1, 4, 400, [7, 4, 4, 4]), (2048, 40.0 / 2048, 3, 1080, [3, 3, 3])])
def test_match_vs_oracle(hs_mod, ctx, det, sim, side, cell, levels, R, iters):
    oc = det
    rep, ref, segs, rng = build_pair(hs_mod, ctx, oc, sim, side, cell, levels, R, 12)
    for l, it in enumerate(iters):
        rep.Maps[l].EstimateIterations = it
    matcher = hs_mod.ScanMatcher(4)
    true_pose = np.array([20.6, 20.25, 0.12], np.float32)
    rays, xy = sim.make_scan(segs, true_pose, R, rng)
    scan = hs_mod.ScanCloud(xy)
    hints = [true_pose + np.array(d, np.float32) for d in ((0, 0, 0), (0.1, -0.08, 0.03), (-0.15, 0.1, -0.05), (0.02, 0.3, 0.0))]
    for hint in hints:
        got = matcher.MatchData(rep, scan, hint)
        for T in (1, 4):
            want = oc.match_pyramid(ref, xy, hint, iters, n_threads=T)
            assert abs(got[0] - want[0]) < POS_TOL and abs(got[1] - want[1]) < POS_TOL, (hint, T, got, want)
            assert abs(math.remainder(float(got[2]) - float(want[2]), 2 * math.pi)) < ANG_TOL, (hint, T, got, want)
    # H / dTr at a fixed map pose (single level API) within fp32 summation noise.  The entries are binary32 sums of ~R
    # products whose order differs from the reference's thread chunks (ScanMatcher.cs:149-195); an entry's error scales with
    # the sum of the magnitudes of its terms, i.e. with the LARGEST entry of the matrix (the small off-diagonal entries are
    # differences of large sums), not with the entry itself.  Measured on MI355X (tools: 8 poses, 400^2 and 2048^2):
    # <= 7e-7 of the largest entry -- the same as the reference's own spread between 1 and 4 threads (<= 7e-7 for H, 1.7e-6
    # for dTr).  The bound below is 4e-6 of the largest entry.
    rep.set_scan(scan)
    for hint in hints:
        est_map = ref[0].map_pose(hint)
        H, d = rep.Maps[0].Hessian(est_map)
        Hr, dr = ref[0].hessian(xy, est_map, 1)
        assert np.abs(H - Hr).max() <= 4e-6 * np.abs(Hr).max(), (hint, np.abs(H - Hr).max(), np.abs(Hr).max())
        assert np.abs(d - dr).max() <= 4e-6 * max(np.abs(dr).max(), 1.0), (hint, np.abs(d - dr).max(), np.abs(dr).max())
    # MatchData(OccGridMap) on one level (:64-84)
    got = matcher.MatchData(rep.Maps[1], scan, hints[1])
    want = ref[1].match(xy, hints[1], iters[1], 1)
    assert np.allclose(got[:2], want[:2], atol=POS_TOL) and abs(got[2] - want[2]) < ANG_TOL
    # batched hints against single-hint launches: up to 8 hints run the single match's kernel (512 lanes per hint) -- the same
    # floats; larger batches run 256 lanes per hint, whose wave partials are summed in another order -- the same pose within the
    # noise of a binary32 sum (1e-5 of a cell)
    singles = [matcher.MatchData(rep, scan, hint) for hint in hints]
    small = matcher.MatchDataBatch(rep, scan, np.stack(hints))
    for i in range(len(hints)):
        assert (small[i] == singles[i]).all()
    batch = matcher.MatchDataBatch(rep, scan, np.stack(hints * 8))
    for i in range(len(hints) * 8):
        assert np.abs(np.asarray(batch[i]) - np.asarray(singles[i % len(hints)])).max() < 2e-5, (i, batch[i], singles[i % len(hints)])
    # empty scan returns the hint (:82-83); a hint far outside the map leaves the estimate unchanged (:97,:124)
    assert (matcher.MatchData(rep, hs_mod.ScanCloud(np.zeros((0, 2), np.float32)), hints[1]) == hints[1]).all()
    far = np.array([500.0, 500.0, 0.3], np.float32)
    got = matcher.MatchData(rep, scan, far)
    want = oc.match_pyramid(ref, xy, far, iters, 1)
    assert (np.asarray(got) == np.asarray(want)).all()                 # (no iteration moves it: the same float transforms there and back)
    rep.close()


def test_match_long_scan_vs_oracle(hs_mod, ctx, det, sim):
    """Scans beyond the 2048 points the matcher keeps in LDS are read from global memory (hs_hessian_block<.., false>): single
    match and a batch, against the oracle (ScanMatcher.cs:41-125)."""
    oc = det
    rep, ref, segs, rng = build_pair(hs_mod, ctx, oc, sim, 400, 0.1, 3, 400, 10)
    true_pose = np.array([20.6, 20.25, 0.12], np.float32)
    rays, xy = sim.make_scan(segs, true_pose, 2500, rng)
    scan = hs_mod.ScanCloud(xy)
    matcher = hs_mod.ScanMatcher(4)
    hints = [true_pose + np.array(d, np.float32) for d in ((0.1, -0.08, 0.03), (-0.15, 0.1, -0.05))]
    for hint in hints:
        got = matcher.MatchData(rep, scan, hint)
        want = oc.match_pyramid(ref, xy, hint, [3, 3, 3], n_threads=1)
        assert abs(got[0] - want[0]) < POS_TOL and abs(got[1] - want[1]) < POS_TOL, (hint, got, want)
        assert abs(math.remainder(float(got[2]) - float(want[2]), 2 * math.pi)) < ANG_TOL, (hint, got, want)
    batch = matcher.MatchDataBatch(rep, scan, np.stack(hints * 6))
    for i in range(12):
        want = oc.match_pyramid(ref, xy, hints[i % 2], [3, 3, 3], n_threads=1)
        assert np.abs(np.asarray(batch[i][:2]) - np.asarray(want[:2])).max() < POS_TOL and abs(batch[i][2] - want[2]) < ANG_TOL
    rep.close()


def test_hector_processor_gating(hs_mod, ctx, det, sim):
    """HectorSLAMProcessor.Update (:86-126): map-without-matching, distance / angle thresholds, DegDiff quirk."""
    oc = det
    segs = sim.default_field()
    start = np.array([20.0, 20.0, 0.0], np.float32)
    proc = hs_mod.HectorSLAMProcessor(0.1, (400, 400), start, 3, 4, ctx=ctx)
    proc.MinDistanceDiffForMapUpdate = 0.4                              # Simulation/MainWindow.xaml.cs:78-79
    proc.MinAngleDiffForMapUpdate = math.radians(8)
    ref = oc.make_pyramid(0.1, 400, 400, 3)
    rng = sim.PCG32(8)
    pose = start.copy()
    assert (proc.LastMapUpdatePose < -3e38).all()
    last = None
    for loop in range(16):
        true_pose = np.array([20 + 0.06 * loop, 20 + 0.01 * loop, 0.004 * loop], np.float32)
        rays, xy = sim.make_scan(segs, true_pose, 400, rng)
        scan = hs_mod.ScanCloud(xy)
        hint = proc.MatchPose if loop else start
        map_only = loop < 10                                            # MainWindow.xaml.cs:179
        updated = proc.Update(scan, hint, map_only)
        if map_only:
            want = np.asarray(hint, np.float32)
        else:
            want = oc.match_pyramid(ref, xy, hint, [3, 3, 3], 1)
        got = proc.MatchPose
        assert np.allclose(got, want, atol=POS_TOL), loop
        d2 = np.inf if last is None else float((got[0] - last[0]) ** 2 + (got[1] - last[1]) ** 2)
        ang = -np.inf if last is None else oc.deg_diff(float(got[2]), float(last[2]))
        expect = bool(map_only or d2 > 0.4 ** 2 or ang > math.radians(8))
        assert updated == expect, (loop, d2, ang)
        if updated:
            for g in ref:
                g.update_by_scan(xy, got)
            last = got.copy()
            assert (proc.LastMapUpdatePose == got).all()
        if map_only:
            for l in range(3):
                c = proc.MapRep.Maps[l].GetCells()
                assert (c["update_index"] == ref[l].cells["update_index"]).all() and (c["value"] == ref[l].cells["value"]).all()
    assert proc.MatchTiming > 0 and proc.UpdateTiming > 0
    proc.Reset()
    assert (proc.MatchPose == start).all() and (proc.MapRep.Maps[0].GetCells()["value"] == 0).all()
    proc.Dispose()


def test_hector_processor_long_run(hs_mod, ctx, det, sim):
    """HectorSLAMProcessor over 120 scans of the lap (simulator settings, :76-86): every match within tolerance of the
    oracle's match on the oracle's maps; the oracle maps are advanced with the device's pose so that the grids stay
    comparable bit for bit (a 1e-7 pose difference may legitimately round a beam end into the neighbouring cell)."""
    oc = det
    segs = sim.default_field()
    traj, _ = sim.lap_trajectory(110, 0.1)
    traj = np.concatenate([np.repeat(traj[:1], 10, axis=0), traj[1:]])
    start = traj[0].copy()
    proc = hs_mod.HectorSLAMProcessor(0.1, (400, 400), start, 4, 4, ctx=ctx)
    proc.MinDistanceDiffForMapUpdate = 0.4
    proc.MinAngleDiffForMapUpdate = math.radians(8)
    its = [7, 4, 4, 4]
    for l in range(4):
        proc.MapRep.Maps[l].EstimateIterations = its[l]
    ref = oc.make_pyramid(0.1, 400, 400, 4)
    rng = sim.PCG32(31)
    n_updates = 0
    for loop, tp in enumerate(traj):
        rays, xy = sim.make_scan(segs, tp, 400, rng)
        hint = proc.MatchPose.copy()
        updated = proc.Update(hs_mod.ScanCloud(xy), hint, loop < 10)
        got = proc.MatchPose
        want = np.asarray(hint, np.float32) if loop < 10 else oc.match_pyramid(ref, xy, hint, its, 4)
        assert abs(got[0] - want[0]) < POS_TOL and abs(got[1] - want[1]) < POS_TOL and abs(got[2] - want[2]) < ANG_TOL, loop
        if updated:
            n_updates += 1
            for g in ref:
                g.update_by_scan(xy, got)
        if loop % 20 == 19 or loop == len(traj) - 1:
            for l in range(4):
                assert cells_equal(proc.MapRep.Maps[l].GetCells(), ref[l].cells), (loop, l)
                assert proc.MapRep.Maps[l].GetMapExtends() == ref[l].map_extends()
    assert 10 < n_updates < len(traj)                                  # the distance / angle gate did skip scans
    err = proc.MatchPose - traj[-1]
    assert math.hypot(err[0], err[1]) < 0.1 and abs(err[2]) < math.radians(1)
    proc.Dispose()



def test_hector_processor_wait_update_mode():
    """HectorSLAMProcessor.Update enqueues the grid update and returns; SLAMHIP_HS_WAIT_UPDATE=1 (wait for it, the former
    behaviour) and SLAMHIP_NO_HOSTWAIT=1 (no host mailbox) must give the same results."""
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    sel = "test_hector_processor_gating or test_hector_processor_long_run or test_match"
    # (round 6: ... and without the single match's helper workgroup -- a prefetch, never a result -- or with two of them)
    for env_extra in ({"SLAMHIP_HS_WAIT_UPDATE": "1"}, {"SLAMHIP_NO_HOSTWAIT": "1"}, {"SLAMHIP_K4_HELPERS": "0"}, {"SLAMHIP_K4_HELPERS": "2"}):
        env = dict(os.environ); env.update(env_extra)
        r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(here, "test_gpu_hector.py"), "-m", "gpu", "-x", "-q",
                            "-k", sel], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
        assert r.returncode == 0, (env_extra, r.stdout.decode(errors="replace")[-3000:])


def test_grid_checksum(hs_mod, ctx, sim, checksum_np):
    """slamhip_hs_checksum (the replica check of the grids) equals the NumPy restatement over the downloaded
    cells on every level, and moves with an update."""
    segs = sim.default_field()
    rep = hs_mod.MapRepMultiMap(0.1, (400, 400), 3, ctx=ctx)
    rng = sim.PCG32(8)
    before = [rep.Maps[l].checksum() for l in range(3)]
    for p in sim.trajectory(3):
        _, xy = sim.make_scan(segs, p, 360, rng)
        rep.UpdateByScan(hs_mod.ScanCloud(xy), p)
    for l in range(3):
        cells = rep.Maps[l].GetCells()
        got = rep.Maps[l].checksum()
        assert got == (checksum_np(cells["value"]), checksum_np(cells["update_index"])), l
        assert got != before[l]
    rep.close()


def test_grid_update_changing_scan_sizes(hs_mod, ctx, det, sim):
    """The cell kernel deals its lines to the XCDs by the sector bounds the PREVIOUS update left on the device (valid while the
    scan keeps its line count, else by count): updates whose scans change size and order from one to the next -- with repeats,
    so that a record is used, dropped and used again -- must equal the oracle cell for cell."""
    oc = det
    side, cell, levels = 400, 0.1, 3
    segs = sim.default_field()
    rep = hs_mod.MapRepMultiMap(cell, (side, side), levels, ctx=ctx)
    ref = oc.make_pyramid(cell, side, side, levels)
    rng = sim.PCG32(19)
    perm = np.random.default_rng(2)
    traj = sim.trajectory(14)
    for it, (R, shuffle) in enumerate([(360, False), (360, False), (360, True), (361, False), (90, False), (90, False), (1500, False), (1500, True),
                                       (1500, False), (1, False), (360, False), (360, False), (3100, False), (360, False)]):
        p = traj[it]
        _, xy = sim.make_scan(segs, p, R, rng)
        if shuffle:
            xy = xy[perm.permutation(xy.shape[0])]
        rep.UpdateByScan(hs_mod.ScanCloud(xy), p)
        for l in range(levels):
            ref[l].update_by_scan(xy, p)
    for l in range(levels):
        assert cells_equal(rep.Maps[l].GetCells(), ref[l].cells), l
    rep.close()


def test_grid_update_large_scan_path():
    """The grid update of scans with more lines than the cell kernel keeps in LDS (k5_prepare + the cell kernel reading its
    tables from memory), forced on the ordinary test scans with SLAMHIP_K5_TWO_LAUNCHES=1: same cells.  (Natively the path
    runs in the 3500-point case of test_grid_update_vs_oracle.)"""
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    # (SLAMHIP_K5_EQUAL_SECTORS=1: the one-launch path with its lines dealt to the XCDs by count instead of by the work the
    # last update's first workgroups measured -- the dealing must never show in the cells)
    for extra in ({"SLAMHIP_K5_TWO_LAUNCHES": "1"}, {"SLAMHIP_K5_EQUAL_SECTORS": "1"}):
        env = dict(os.environ, **extra)
        r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(here, "test_gpu_hector.py"), "-m", "gpu", "-x", "-q", "-k",
                            "grid_golden or grid_update_vs_oracle or map_extends or order_dependence or unordered_dense or processor_gating"],
                           env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
        assert r.returncode == 0, (extra, r.stdout.decode(errors="replace")[-3000:])


def test_reset_probabilities_are_current_d5(hs_mod, ctx, det, sim):
    """Deviation D5 (OccGridMap.cs:97-107,147,244-252; tests/test_oracle_kat.py shows the C# behaviour): after Reset the
    reference can serve pre-reset probabilities out of its cache.  The library's probability grids follow the cells: the
    hand case (0.9 before the Reset, 0.4 after Reset + a scan that crosses the cell as free), and a match after
    Reset + remapping equals the oracle's match on a freshly built map."""
    rep = hs_mod.MapRepMultiMap(1.0, (32, 32), 1, ctx=ctx)
    cell = 10 * 32 + 15
    pose = np.array([10.0, 10.0, 0.0], np.float32)
    rep.UpdateByScan(hs_mod.ScanCloud(np.array([[5.0, 0.0]], np.float32)), pose)
    assert abs(float(rep.Maps[0].GetCachedProbability(cell)[0]) - 0.9) < 1e-6
    rep.Reset()
    assert float(rep.Maps[0].GetCachedProbability(cell)[0]) == 0.5
    rep.UpdateByScan(hs_mod.ScanCloud(np.array([[8.0, 0.0]], np.float32)), pose)
    assert abs(float(rep.Maps[0].GetCachedProbability(cell)[0]) - 0.4) < 1e-6       # the C# cache would say 0.9 here (same epoch 1)
    g = det.Grid(1.0, 32, 32)
    g.update_by_scan(np.array([[8.0, 0.0]], np.float32), pose)
    assert float(rep.Maps[0].GetCachedProbability(cell)[0]) == np.float32(g.prob(cell))
    rep.close()

    # a pyramid that is mapped, matched (the match touches -- in the C# -- the cache), reset and mapped elsewhere: the match on the
    # new map equals the oracle's match on a map that never saw the first one
    segs = sim.default_field()
    rng = sim.PCG32(11)
    side, cellm, R = 400, 0.1, 360
    rep = hs_mod.MapRepMultiMap(cellm, (side, side), 2, ctx=ctx)
    m = hs_mod.ScanMatcher(1)
    first = [np.array([12.0 + 0.05 * i, 14.0, 0.02 * i], np.float32) for i in range(4)]
    for p in first:
        rep.UpdateByScan(hs_mod.ScanCloud(sim.make_scan(segs, p, R, rng)[1]), p)
    m.MatchData(rep, hs_mod.ScanCloud(sim.make_scan(segs, first[-1], R, rng)[1]), first[-1])
    rep.Reset()
    levels = det.make_pyramid(cellm, side, side, 2)
    second = [np.array([20.0 + 0.05 * i, 20.0 + 0.02 * i, 0.01 * i], np.float32) for i in range(4)]
    for p in second:
        xy = sim.make_scan(segs, p, R, rng)[1]
        rep.UpdateByScan(hs_mod.ScanCloud(xy), p)
        for g in levels:
            g.update_by_scan(xy, p)
    for l, g in enumerate(levels):
        assert cells_equal(rep.Maps[l].GetCells(), g.cells)
    xy = sim.make_scan(segs, second[-1], R, rng)[1]
    hint = second[-1] + np.array([0.05, -0.04, 0.01], np.float32)
    got = m.MatchData(rep, hs_mod.ScanCloud(xy), hint)
    ref = det.match_pyramid(levels, xy, hint, [3, 3])
    assert abs(got[0] - ref[0]) < POS_TOL and abs(got[1] - ref[1]) < POS_TOL and abs(got[2] - ref[2]) < ANG_TOL
    m.Dispose()
    rep.close()
