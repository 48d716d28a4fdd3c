"""GPU parity tests for the CoreSLAM hot path: HIP kernels (through the C-ABI) vs the CPU oracle and the
committed golden fixtures.  Integer outputs (distances, arg-min, HoleMap / ObstacleMap cells) must be
bit-exact; poses must be identical floats (same IEEE adds on both sides)."""
import glob
import math
import os
import time

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
INT_MAX = 2 ** 31 - 1


@pytest.fixture(scope="module")
def cs_mod():
    import slam.net_amd.coreslam as m
    return m


@pytest.fixture(scope="module")
def ctx(cs_mod):
    c = cs_mod.Context(0)
    yield c
    c.close()


@pytest.fixture(scope="module")
def det(oc):
    oc.set_trig_mode(oc.TRIG_DET)
    yield oc
    oc.set_trig_mode(oc.TRIG_LIBM)


def make_dev(cs_mod, ctx, size, obst=64):
    return cs_mod.CoreSlamDevice(ctx, 40.0, size, obst)


# ---- K1 distance ------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", sorted(os.path.basename(p) for p in glob.glob(os.path.join(GOLD, "cs_distance_*.npz"))))
def test_distance_golden(cs_mod, ctx, name):
    g = np.load(os.path.join(GOLD, name))
    dev = make_dev(cs_mod, ctx, int(g["size"]))
    assert dev.hole_scale == float(g["scale"])
    dev.holemap_upload(g["pixels"])
    dev.set_scan(g["xy"])
    d, bi, bd = dev.distance_pxcs(g["pxcs"])
    assert (d == g["dist"]).all()
    assert bi == int(g["best"]) and bd == int(g["dist"][bi])
    # poses + device deterministic trig
    poses = np.vstack([g["base"][None], g["base"][None] + g["offs"]]).astype(np.float32)
    d2, bi2, bd2 = dev.distance_poses(poses)
    assert (d2 == g["dist"]).all() and bi2 == bi
    # search over the jitter list: candidate 0 = base pose
    dev.set_offsets(g["offs"])
    pose, dist, idx = dev.search(g["base"])
    assert idx == bi and dist == bd
    assert (pose == poses[bi]).all()
    dev.close()


def test_distance_quirks(cs_mod, ctx, oc):
    size = 128
    dev = make_dev(cs_mod, ctx, size)
    pix = np.full(size * size, 32750, np.uint16)
    dev.holemap_upload(pix)
    R = 1080
    ang = np.arange(R) * (2 * math.pi / R)
    xy = np.stack([3.0 * np.cos(ang), 3.0 * np.sin(ang)], 1).astype(np.float32)
    scale = dev.hole_scale
    pxcs = oc.pose_to_pxcs([20, 20, 0.1], scale)
    dev.set_scan(xy)
    assert dev.distance_pxcs(pxcs[None])[0][0] == 33536000            # uniform map (:253)
    xy_half = xy.copy(); xy_half[: R // 2] += 1000.0
    dev.set_scan(xy_half)
    assert dev.distance_pxcs(pxcs[None])[0][0] == 16768000            # divides by ALL points
    dev.set_scan(xy + 1000.0)
    d, bi, bd = dev.distance_pxcs(np.stack([pxcs, pxcs]))
    assert d[0] == INT_MAX and bi == 0 and bd == INT_MAX              # :257, earliest wins ties
    # truncation toward zero / NaN (SAFE kernel path)
    pix2 = np.zeros(size * size, np.uint16); pix2[3 * size] = 77
    dev.holemap_upload(pix2)
    dev.set_scan(np.array([[0.0, 0.0]], np.float32))
    cands = np.array([[-0.75, 3.2, 1, 0], [-1.0, 3.2, 1, 0], [np.nan, 3.2, 1, 0], [np.inf, 3.2, 1, 0],
                      [0.5, 3.2, np.nan, 0], [3e9, 3.2, 1, 0]], np.float32)
    d, bi, bd = dev.distance_pxcs(cands)
    ref, rbi, rbd = oc.distance_batch_pxcs(pix2, size, np.array([[0.0, 0.0]], np.float32), cands)
    assert (d == ref).all() and d[0] == 77 * 1024 and (d[1:] == INT_MAX).all() and bi == rbi
    # non-finite scan point: SAFE path on the point side
    dev.set_scan(np.array([[0.0, 0.0], [np.nan, 1.0], [1e30, -1e30]], np.float32))
    d = dev.distance_pxcs(cands[:2])[0]
    ref = oc.distance_batch_pxcs(pix2, size, np.array([[0.0, 0.0], [np.nan, 1.0], [1e30, -1e30]], np.float32), cands[:2])[0]
    assert (d == ref).all()
    dev.close()


def test_distance_64bit_and_ragged(cs_mod, ctx, oc):
    size = 64
    dev = make_dev(cs_mod, ctx, size)
    dev.holemap_upload(np.full(size * size, 65535, np.uint16))
    pxcs = oc.pose_to_pxcs([20, 20, 0], dev.hole_scale)
    for R in (1, 2, 31, 32, 33, 63, 64, 65, 1079, 1080, 1081, 4097):
        dev.set_scan(np.zeros((R, 2), np.float32))
        assert dev.distance_pxcs(pxcs[None])[0][0] == 65535 * 1024    # sum*1024 > 2^32 at R >= 65 (H3)
    rng = np.random.default_rng(5)
    pix = rng.integers(0, 65536, size * size).astype(np.uint16)
    dev.holemap_upload(pix)
    for R, K in ((7, 1), (100, 255), (333, 257), (1080, 1000)):
        xy = rng.uniform(-25, 25, (R, 2)).astype(np.float32)
        poses = np.stack([rng.uniform(-5, 45, K), rng.uniform(-5, 45, K), rng.uniform(-7, 7, K)], 1).astype(np.float32)
        pxcs = np.stack([oc.pose_to_pxcs(p, dev.hole_scale) for p in poses])
        dev.set_scan(xy)
        d, bi, bd = dev.distance_pxcs(pxcs)
        ref, rbi, rbd = oc.distance_batch_pxcs(pix, size, xy, pxcs)
        assert (d == ref).all() and bi == rbi and bd == rbd
    dev.close()


def test_empty_scan_is_state_error(cs_mod, ctx):
    import slam.net_amd.capi as capi
    dev = make_dev(cs_mod, ctx, 64)
    with pytest.raises(capi.SlamhipError):
        dev.distance_pxcs(np.zeros((1, 4), np.float32))
    dev.close()


@pytest.mark.parametrize("size,R,K", [(400, 360, 4001), (1024, 1080, 16384), (2048, 1080, 16384),
                                      (512, 360, 70001),       # more than 64 candidate groups: listed theta tails + uniform middle
                                      (1024, 500, 140000),
                                      (400, 360, 12288), (400, 360, 12289),    # the candidate counts at which the group size changes (512 | 1024 | 2048 candidates per group)
                                      (256, 200, 65535), (256, 200, 65536)])
def test_search_full_size_vs_oracle(cs_mod, ctx, det, sim, size, R, K):
    """BASELINE configs C1/C2/C3 sizes: every candidate's distance and the arg-min vs the C oracle."""
    oc = det
    segs = sim.default_field()
    dev = make_dev(cs_mod, ctx, size, size // 4)
    scale = dev.hole_scale
    # build a realistic map with the HIP mapping path itself, then hand the SAME map to the oracle
    rng = sim.PCG32(1234)
    traj = sim.trajectory(8)
    for p in traj:
        rays, xy = sim.make_scan(segs, p, R, rng)
        dev.set_scan(xy)
        dev.update_holemap(p)
    pix = dev.holemap_download()
    true_pose = sim.trajectory(9)[-1]
    rays, xy = sim.make_scan(segs, true_pose, R, sim.PCG32(99))
    base = (true_pose + np.array([0.03, -0.02, math.radians(1.0)], np.float32)).astype(np.float32)
    offs = sim.gaussian_offsets(K - 1)
    dev.set_scan(xy)
    dev.set_offsets(offs)
    pose, dist, idx = dev.search(base)
    rbi, rpose, rbd, rall = oc.search(pix, size, scale, xy, base, offs)
    assert idx == rbi and dist == rbd and (pose == rpose).all()
    poses = np.vstack([base[None], base[None] + offs]).astype(np.float32)
    d, bi, bd = dev.distance_poses(poses)
    assert (d == rall).all() and bi == rbi
    # sharded search: min over shards == full search, for uneven shard counts
    for n in (2, 3, 8):
        keys = [dev.search_shard(base, K * r // n, K * (r + 1) // n - K * r // n) for r in range(n)]
        p2, d2, i2 = dev.pose_from_key(base, min(keys))
        assert i2 == rbi and d2 == rbd and (p2 == rpose).all()
    # device-generated (stratified) offsets: feed the SAME list to the oracle
    dev.generate_offsets(K - 1, 0.1, math.radians(10.0), seed=7, stream=3)
    goffs = dev.offsets_download()
    assert np.isfinite(goffs).all() and (np.diff(goffs[:, 2]) >= 0).all()
    assert abs(goffs[:, 0].std() - 0.1) < 0.01 and abs(goffs[:, 2].std() - math.radians(10.0)) < 0.01
    pose, dist, idx = dev.search(base)
    rbi, rpose, rbd, _ = oc.search(pix, size, scale, xy, base, goffs)
    assert idx == rbi and dist == rbd and (pose == rpose).all()
    for n in (2, 5):                               # shards of the device-sorted list (the base pose sits mid-list in shard 0)
        keys = [dev.search_shard(base, K * r // n, K * (r + 1) // n - K * r // n) for r in range(n)]
        p2, d2, i2 = dev.pose_from_key(base, min(keys))
        assert i2 == rbi and d2 == rbd and (p2 == rpose).all()
    # a list that is searched before anyone reads it is produced inside the search's gather launch: same list, same answer
    dev.generate_offsets(K - 1, 0.07, math.radians(6.0), seed=11, stream=5)
    pose, dist, idx = dev.search(base)
    goffs2 = dev.offsets_download()
    dev.generate_offsets(K - 1, 0.07, math.radians(6.0), seed=11, stream=5)
    assert (dev.offsets_download() == goffs2).all() and not (goffs2 == goffs).all()
    rbi, rpose, rbd, _ = oc.search(pix, size, scale, xy, base, goffs2)
    assert idx == rbi and dist == rbd and (pose == rpose).all()
    assert dev.selfcheck_failures == 0
    dev.close()


def ulp_diff32(a, b):
    """distance in units in the last place between binary32 arrays (finite values of one sign or around zero)"""
    ia = np.asarray(a, np.float32).view(np.int32).astype(np.int64)
    ib = np.asarray(b, np.float32).view(np.int32).astype(np.int64)
    ia = np.where(ia < 0, -(ia & 0x7FFFFFFF), ia)
    ib = np.where(ib < 0, -(ib & 0x7FFFFFFF), ib)
    return np.abs(ia - ib)


def test_generated_offsets_match_the_philox_specification(cs_mod, ctx, npo):
    """The device generator that stands where FillRandomQueues stood (CoreSLAMProcessor.cs:599-612) against its specification:
    (a) one Philox4x32-10 block computed ON THE DEVICE equals Random123's published known answers (the integer stream, bit for bit);
    (b) a downloaded list equals the NumPy restatement of k_jitter (oracle/np_oracle.py: philox_jitters, binary64) to within the
    accuracy of the device's logf / sincosf / normcdfinvf -- dtheta to a few units in the last place, dx and dy to ~1e-6 sigma (the
    device's sincosf carries an absolute error of a few 1e-7) -- while a wrong word of the integer stream anywhere would move a
    value by about one sigma; three (n, sigmas, seed, stream) with 64-bit seeds and streams."""
    for ctr, key, want in npo.PHILOX4X32_10_KAT:
        assert ctx.philox4x32_10(ctr, key) == want
    assert ctx.philox4x32_10((5, 0, 3, 0), (42, 0)) == tuple(int(x) for x in npo.philox4x32_10(np.array([5, 0, 3, 0], np.uint64), (42, 0)))
    dev = make_dev(cs_mod, ctx, 256)
    worst_xy, worst_th = 0.0, 0
    for n, sxy, sth, seed, stream in ((4000, 0.1, math.radians(10.0), 42, 1), (16383, 0.05, math.radians(3.0), 7, 3),
                                      (70001, 0.2, math.radians(20.0), (1 << 40) + 5, (1 << 33) + 2)):
        dev.generate_offsets(n, sxy, sth, seed=seed, stream=stream)
        got = dev.offsets_download()
        want = npo.philox_jitters(n, sxy, sth, seed=seed, stream=stream)
        assert got.shape == want.shape
        # dx, dy = sigma * rad * cos / sin(angle): the device's sincosf reduces its argument in binary32 -- an ABSOLUTE error of a few
        # 1e-7 in cos / sin (measured on MI355X: 7e-7 at worst), so the bound is absolute, in units of sigma * rad (rad <= 5.8)
        _, u = npo.philox_jitter_words(n, seed, stream)
        rad = np.sqrt(-2.0 * np.log(u[:, 0]))                  # (0 where u is exactly 1.0)
        for k in (0, 1):
            e = np.abs(got[:, k].astype(np.float64) - want[:, k]) / (sxy * np.maximum(rad, 0.05))
            worst_xy = max(worst_xy, float(e.max()))
        # dtheta = sigma_theta * normcdfinvf(quantile): a few units in the last place (relative), 1e-9 sigma around zero
        d = ulp_diff32(got[:, 2], want[:, 2].astype(np.float32))
        near0 = np.abs(want[:, 2]) < 1e-3 * sth
        assert (np.abs(got[near0, 2].astype(np.float64) - want[near0, 2]) < 1e-8 * sth).all()
        worst_th = max(worst_th, int(d[~near0].max()))
    print("generated list vs the binary64 restatement: worst |error| of dx, dy in units of sigma_xy * rad: %.2e; worst ulp distance of dtheta: %d" % (worst_xy, worst_th))
    assert worst_xy < 2.5e-6 and worst_th <= 32, (worst_xy, worst_th)      # (a wrong Philox word anywhere moves a value by ~1 sigma)
    dev.close()


def test_search_changing_scans_one_candidate_list(cs_mod, ctx, det, sim):
    """The per-scan flow: one candidate list, a new scan before every search.  The search launch keeps the layout made for the
    previous scan while its counts of ray ranges are legal for the new scan's ray blocks, and makes the new one while the host
    waits for the result (cs_launch_distance / cs_layout_idle_refresh).  Scans that differ wildly in size and order -- ordered,
    shuffled (blocks of one ray), tiny, > 2000 rays -- with the tile self-check on (in the SLAMHIP_K1_VERIFY re-run below): every
    winner equals the oracle's, through the blocking search, the fused call and the shard call alike."""
    oc = det
    size, K = 1024, 6000
    segs = sim.default_field()
    dev = make_dev(cs_mod, ctx, size, 64)
    ref = np.full(size * size, 32750, np.uint16)
    rng = sim.PCG32(41)
    for p in sim.trajectory(6):
        _, xy = sim.make_scan(segs, p, 720, rng)
        dev.set_scan(xy); dev.update_holemap(p)
        oc.update_holemap(ref, size, dev.hole_scale, xy, p)
    offs = sim.gaussian_offsets(K - 1, seed=3)
    dev.set_offsets(offs)
    perm = np.random.default_rng(5)
    traj = sim.trajectory(22)[6:]
    for it, (R, shuffle) in enumerate([(720, False), (725, False), (720, True), (90, False), (2300, True), (2300, False), (1, False), (1080, False),
                                       (1079, True), (360, False), (1080, False), (7, True), (1500, False), (1500, True), (720, False), (64, False)]):
        p = traj[it]
        _, xy = sim.make_scan(segs, p, R, rng)
        if shuffle:
            xy = xy[perm.permutation(xy.shape[0])]
        dev.set_scan(xy)
        base = (p + np.array([0.02, -0.01, 0.01], np.float32)).astype(np.float32)
        rbi, rpose, rbd, _ = oc.search(ref, size, dev.hole_scale, xy, base, offs)
        how = it % 3
        if how == 0:
            pose, dist, idx = dev.search(base)
        elif how == 1:
            pose, dist, idx = dev.pose_from_key(base, dev.search_shard(base, 0, K))
        else:
            pose, dist, idx = dev.search_and_update(base, 0.6, 50, 10)
            rpose[2] = oc.normalize_angle(rpose[2])
        assert idx == rbi and dist == rbd and (pose == rpose).all(), (it, R, shuffle)
        if how == 2:
            oc.update_holemap(ref, size, dev.hole_scale, xy, rpose)
    assert (dev.holemap_download() == ref).all()
    if os.environ.get("SLAMHIP_EXPECT_SELFCHECK"):
        assert dev.selfcheck_failures == 0
    dev.close()


@pytest.mark.parametrize("size,R,K", [(512, 720, 3001), (1024, 1080, 16385), (400, 360, 12289), (512, 500, 20000), (256, 200, 65536), (512, 360, 70001)])
def test_heading_lattice(cs_mod, ctx, det, sim, size, R, K):
    """The opt-in heading lattice (slamhip_cs_generate_offsets_lattice): the candidates a lane of the search kernel evaluates share
    their dtheta bit for bit (2 per lane below 65 536 candidates, 4 from there on), the un-jittered pose's stratum has dtheta = 0;
    the search over it -- through the kernel variant that forms the ray products once per lane -- returns what the oracle returns
    for the downloaded list, and what the ordinary kernel returns for it (SLAMHIP_K1_NO_LATTICE is read once per process, so the
    comparison is with the same list handed back through slamhip_cs_set_offsets: an explicit list never takes the lattice kernel)."""
    oc = det
    segs = sim.default_field()
    dev = make_dev(cs_mod, ctx, size, size // 4)
    rng = sim.PCG32(4321)
    traj = sim.trajectory(8)
    for p in traj:
        _, xy = sim.make_scan(segs, p, R, rng)
        dev.set_scan(xy); dev.update_holemap(p)
    pix = dev.holemap_download()
    true_pose = sim.trajectory(9)[-1]
    _, xy = sim.make_scan(segs, true_pose, R, sim.PCG32(99))
    base = (true_pose + np.array([0.03, -0.02, math.radians(1.0)], np.float32)).astype(np.float32)
    dev.set_scan(xy)
    n = K - 1
    dev.generate_offsets(n, 0.1, math.radians(10.0), seed=77, stream=4, lattice=True)
    pose, dist, idx = dev.search(base)                       # (the list is produced inside the search's gather launch)
    offs = dev.offsets_download()
    assert np.isfinite(offs).all()
    if K <= 12288:                                           # small searches: one candidate per lane, the plain (sorted) list
        assert (np.diff(offs[:, 2]) >= 0).all()
        rbi, rpose, rbd, _ = oc.search(pix, size, dev.hole_scale, xy, base, offs)
        assert idx == rbi and dist == rbd and (pose == rpose).all()
        dev.close()
        return
    # structure: evaluation position j of flat f (the un-jittered pose, flat 0, sits at zero_pos); lanes = group / candidates per lane
    grp, cpl = (2048, 4) if K >= 65536 else (1024, 2)
    lanes = grp // cpl
    zero_pos = min(K - 1, n // 2)
    dth = np.empty(K, np.float32)
    j = np.arange(K)
    flat = np.where(j < zero_pos, j + 1, np.where(j == zero_pos, 0, j))
    dth[:] = np.where(flat > 0, offs[np.maximum(flat - 1, 0), 2], np.float32(0.0))
    stratum = (j // grp) * lanes + (j % lanes)
    bits = dth.view(np.uint32)
    for u in np.unique(stratum)[:: max(1, len(np.unique(stratum)) // 400)]:
        b = bits[stratum == u]
        assert (b == b[0]).all(), u                          # one heading per lane position, bit for bit
    us = np.unique(stratum)
    first_of = np.array([dth[stratum == u][0] for u in us if u != stratum[zero_pos]])
    assert (np.diff(first_of) >= 0).all()                    # the strata ascend (the un-jittered pose's own stratum is pinned to 0)
    assert dth[stratum == stratum[zero_pos]].tolist() == [0.0] * int((stratum == stratum[zero_pos]).sum())
    assert abs(offs[:, 0].std() - 0.1) < 0.01 and abs(offs[:, 2].std() - math.radians(10.0)) < 0.012
    # the same list from the stand-alone generator launch (the list read before it is searched)
    dev.generate_offsets(n, 0.1, math.radians(10.0), seed=77, stream=4, lattice=True)
    assert (dev.offsets_download() == offs).all()
    # parity: oracle on the downloaded list; and the ordinary kernel on the same list as an explicit one
    rbi, rpose, rbd, _ = oc.search(pix, size, dev.hole_scale, xy, base, offs)
    assert idx == rbi and dist == rbd and (pose == rpose).all()
    pose_l, dist_l, idx_l = dev.search(base)                 # (searched again after the download: the lattice kernel once more)
    assert (idx_l, dist_l) == (idx, dist) and (pose_l == pose).all()
    dev.set_offsets(offs)
    pose_e, dist_e, idx_e = dev.search(base)
    assert (idx_e, dist_e) == (idx, dist) and (pose_e == pose).all()
    assert dev.selfcheck_failures == 0                       # (the lattice kernel counts lanes whose candidates do not share a heading)
    # a plain generated list afterwards is sorted again, and a sharded search over a lattice list (ordinary kernel) agrees
    dev.generate_offsets(n, 0.1, math.radians(10.0), seed=77, stream=4, lattice=True)
    keys = [dev.search_shard(base, K * r // 3, K * (r + 1) // 3 - K * r // 3) for r in range(3)]
    p2, d2, i2 = dev.pose_from_key(base, min(keys))
    assert i2 == rbi and d2 == rbd and (p2 == rpose).all()
    dev.generate_offsets(n, 0.1, math.radians(10.0), seed=77, stream=4)
    assert (np.diff(dev.offsets_download()[:, 2]) >= 0).all()
    dev.close()


def test_processor_with_lattice(cs_mod, ctx, det, sim):
    """CoreSLAMProcessor with SetLattice(True): every scan's candidates are a heading lattice (prepared ahead like the plain
    lists); the poses equal those of the oracle state machine fed the downloaded lists."""
    oc = det
    segs = sim.default_field()
    start = np.array([20.0, 20.0, 0.0], np.float32)
    proc = cs_mod.CoreSLAMProcessor(40.0, 256, 64, start, 0.1, math.radians(10), 3200, 4, ctx=ctx)   # (12 801 candidates: a lattice from 12 289 on)
    proc.HoleWidth = 2.0
    proc.SetLattice(True)
    ref = oc.CSProc(40.0, 256, 64, start)
    ref.set_params(hole_width=2.0)
    rng = sim.PCG32(21)
    true_traj = sim.trajectory(14, step=(0.08, 0.03, math.radians(1.0)))
    for i, tp in enumerate(true_traj):
        rays, _ = sim.make_scan(segs, tp, 400, rng)
        est = proc.Pose.copy()
        assert (est == ref.pose).all()
        seg_pose = (est + np.array([0.08, 0.03, math.radians(1.0)], np.float32)).astype(np.float32) if i else est
        proc.Update([cs_mod.ScanSegment(rays, seg_pose)])
        offs = proc.device.offsets_download(12800) if i >= 5 else None      # (the list the searching Update just used)
        ref.update(seg_pose[None], [0, rays.shape[0]], rays, offs)
        assert (proc.Pose == ref.pose).all(), i
    assert (proc.HoleMap.Pixels == ref.holemap).all()
    served, prepared = proc.device.prepared_lists()
    if not any(k in os.environ for k in ("SLAMHIP_NO_SPECULATION", "SLAMHIP_NO_HOSTWAIT", "SLAMHIP_FUSED_WAIT_UPDATES")):
        assert served >= 7, (served, prepared)               # (the lattice lists are prepared ahead like the plain ones)
    assert proc.device.selfcheck_failures == 0
    dth = proc.device.offsets_download(12800)[:, 2]
    assert not (np.diff(dth) >= 0).all()                     # (a lattice, not the sorted plain list)
    proc.Dispose()


def test_prepared_candidate_list(cs_mod, ctx, det, sim):
    """The per-scan flow's candidate list prepared ahead (cs_speculate_next, coreslam.hip): a fused scan on a generated list prepares
    the list of stream + 1 on a side stream; generate_offsets(stream + 1) then swaps it in -- it must be the very list a fresh
    handle generates, the searches on it must equal the oracle's, and a request for any other list (other stream, other sigma) must
    get THAT list, not the prepared one.  Scans change in between: the scan blocks alternate and are stored by the CPU."""
    oc = det
    size, R, K = 512, 720, 3001
    segs = sim.default_field()
    rng = sim.PCG32(99)
    dev = make_dev(cs_mod, ctx, size, 128)
    fresh = make_dev(cs_mod, ctx, size, 128)
    ref_h = dev.holemap_download().copy()
    ref_o = dev.obstaclemap_download().copy()
    traj = sim.trajectory(16)
    scans = [sim.make_scan(segs, p, R, rng)[1] for p in traj]
    for p, xy in zip(traj[:5], scans[:5]):
        dev.set_scan(xy); dev.update_holemap(p); dev.update_obstaclemap(p)
        oc.update_holemap(ref_h, size, dev.hole_scale, xy, p); oc.update_obstaclemap(ref_o, 128, dev.obst_scale, xy, p)

    def expected_list(sxy, sth, seed, stream):
        fresh.generate_offsets(K - 1, sxy, sth, seed=seed, stream=stream)
        return fresh.offsets_download().copy()

    # (stream, sigma_xy, sigma_theta): 5, 6, 7 hit the prepared list from the second on; 9 skips one (miss); then another sigma
    # (miss), its successor (hit), and a repeat of an old stream (miss)
    plan = [(5, 0.1, 0.17), (6, 0.1, 0.17), (7, 0.1, 0.17), (9, 0.1, 0.17), (10, 0.05, 0.17), (11, 0.05, 0.17), (6, 0.05, 0.17),
            (7, 0.05, 0.17), (8, 0.05, 0.17), (9, 0.05, 0.17)]
    for (stream, sxy, sth), p, xy in zip(plan, traj[5:], scans[5:]):
        base = (p + np.array([0.02, -0.03, 0.01], np.float32)).astype(np.float32)
        dev.set_scan(xy)
        dev.generate_offsets(K - 1, sxy, sth, seed=42, stream=stream)
        pose, dist, idx = dev.search_and_update(base, 0.6, 50, 10)
        offs = expected_list(sxy, sth, 42, stream)
        assert (dev.offsets_download() == offs).all(), (stream, sxy)
        rbi, rpose, rbd, _ = oc.search(ref_h, size, dev.hole_scale, xy, base, offs)
        rpose[2] = oc.normalize_angle(rpose[2])
        assert idx == rbi and dist == rbd and (pose == rpose).all(), (stream, sxy)
        # the host-side decode of the same key (slamhip_cs_pose_from_key fetches a host copy of the jitters and keeps it): behind a
        # SERVED list the copy of the previous list must not be what decodes the key (round-4 advisor finding)
        kp, kd, ki = dev.pose_from_key(base, (dist << 32) | idx)
        kp[2] = oc.normalize_angle(kp[2])
        assert ki == idx and kd == dist and (kp == pose).all(), (stream, sxy, kp, pose)
        oc.update_holemap(ref_h, size, dev.hole_scale, xy, rpose); oc.update_obstaclemap(ref_o, 128, dev.obst_scale, xy, rpose)
    assert (dev.holemap_download() == ref_h).all() and (dev.obstaclemap_download() == ref_o).all()
    assert dev.selfcheck_failures == 0
    served, prepared = dev.prepared_lists()
    if "SLAMHIP_NO_SPECULATION" not in os.environ and "SLAMHIP_NO_HOSTWAIT" not in os.environ and "SLAMHIP_FUSED_WAIT_UPDATES" not in os.environ:
        assert prepared == len(plan) and served == 6, (served, prepared)     # streams 6, 7 | 11 | 7, 8, 9 of the plan
    dev.close(); fresh.close()


def test_search_enqueue_ring(cs_mod, ctx, det, sim):
    """slamhip_cs_search_shard_enqueue (the ring of result words: finishing workgroups min straight into the word, no final
    arriver): every key equals the blocking search's and the oracle's -- back to back launches from different poses (each launch
    must find its word all ones: the previous one rests it), interleaved with blocking searches, shards, and a changed list."""
    oc = det
    size, R, K = 1024, 720, 6000
    dev = make_dev(cs_mod, ctx, size)
    segs = sim.default_field()
    rng = sim.PCG32(77)
    ref = np.full(size * size, 32750, np.uint16)
    for p in sim.trajectory(6):
        _, xy = sim.make_scan(segs, p, R, rng)
        dev.set_scan(xy); dev.update_holemap(p)
        oc.update_holemap(ref, size, dev.hole_scale, xy, p)
    true_pose = sim.trajectory(7)[-1]
    _, xy = sim.make_scan(segs, true_pose, R, rng)
    offs = sim.gaussian_offsets(K - 1, 0.1, math.radians(10.0), seed=5)
    dev.set_scan(xy); dev.set_offsets(offs)
    poses = [(true_pose + np.array([0.01 * i, -0.02 * i, 0.004 * i], np.float32)).astype(np.float32) for i in range(9)]
    want = []
    for p in poses:
        bi, _, bd, _ = oc.search(ref, size, dev.hole_scale, xy, p, offs)
        want.append((int(bd) << 32) | int(bi))
    # nine launches back to back, results read afterwards: slots are reused after four launches, so read the last three late
    slots = [dev.search_shard_enqueue(p, 0, K) for p in poses]
    assert len(set(slots)) == 4
    for i in (6, 7, 8):
        assert dev.key_read(slots[i]) == want[i], i
    # one at a time, interleaved with the blocking form (which does not touch the ring) and with shards of the list
    for i, p in enumerate(poses):
        s = dev.search_shard_enqueue(p, 0, K)
        assert dev.search_shard(p, 0, K) == want[i]
        assert dev.key_read(s) == want[i], i
    cuts = [0, 1, 700, 2048, 4097, K]
    keys = []
    for a, b in zip(cuts[:-1], cuts[1:]):
        keys.append(dev.key_read(dev.search_shard_enqueue(poses[3], a, b - a)))
        assert keys[-1] == dev.search_shard(poses[3], a, b - a)
    assert min(keys) == want[3]
    # another list (another group size: 512 lanes x 1), same ring
    offs2 = sim.gaussian_offsets(2999, 0.05, math.radians(3.0), seed=6)
    dev.set_offsets(offs2)
    bi, _, bd, _ = oc.search(ref, size, dev.hole_scale, xy, poses[1], offs2)
    assert dev.key_read(dev.search_shard_enqueue(poses[1], 0, 3000)) == (int(bd) << 32) | int(bi)
    if os.environ.get("SLAMHIP_EXPECT_SELFCHECK"):
        assert dev.selfcheck_failures == 0
    dev.close()


def test_search_plan_launches(cs_mod, ctx, det, sim):
    """The search's plan (k1_plan, slamhip_cs_plan_stats): a queue of enqueue-only searches from EIGHT DIFFERENT poses over two scans
    (a new scan in the middle of the queue: the plan launch must not read tables a launch in the operator's stream is still
    writing) -- once the host runs ahead of the device every search launch is accompanied by its plan launch on a stream of its
    own, the workgroups take their tile steps from the stamped records (or plan for themselves when a record is late) -- and
    every key equals the oracle's.  A plan costs time, never a result."""
    oc = det
    size, R, K = 2048, 1080, 16384
    dev = make_dev(cs_mod, ctx, size, 256)
    segs = sim.default_field()
    rng = sim.PCG32(31)
    ref = np.full(size * size, 32750, np.uint16)
    for p in sim.trajectory(6):
        _, xy = sim.make_scan(segs, p, R, rng)
        dev.set_scan(xy); dev.update_holemap(p)
        oc.update_holemap(ref, size, dev.hole_scale, xy, p)
    tp = sim.trajectory(7)[-1]
    scans = [sim.make_scan(segs, (tp + np.array([0.03 * k, 0.0, 0.004 * k], np.float32)).astype(np.float32), R, rng)[1] for k in range(2)]
    offs = sim.gaussian_offsets(K - 1, 0.1, math.radians(10.0), seed=9)
    dev.set_offsets(offs)
    poses = [(tp + np.array([0.01 * i, -0.015 * i, 0.003 * i], np.float32)).astype(np.float32) for i in range(8)]
    want = {}
    for si, xy in enumerate(scans):
        for pi, p in enumerate(poses):
            bi, _, bd, _ = oc.search(ref, size, dev.hole_scale, xy, p, offs)
            want[(si, pi)] = (int(bd) << 32) | int(bi)
    before = dev.plan_stats
    got = []
    for rep in range(6):
        for si, xy in enumerate(scans):
            dev.set_scan(xy)
            slots = []
            for i in range(24):                                        # 24 launches back to back per scan: the ring holds four keys
                slots.append((dev.search_shard_enqueue(poses[i % 8], 0, K), (si, i % 8)))
                if len(slots) == 3:
                    s0, k0 = slots.pop(0)
                    got.append((dev.key_read(s0), want[k0])) if (i % 5) == 0 else None
            for s0, k0 in slots:
                got.append((dev.key_read(s0), want[k0]))
    assert len(got) > 50 and all(a == b for a, b in got), [(hex(a), hex(b)) for a, b in got if a != b][:4]
    after = dev.plan_stats
    if os.environ.get("SLAMHIP_K1_PLAN", "1") != "0" and not os.environ.get("SLAMHIP_K1_GLOBAL") and not os.environ.get("SLAMHIP_K1_NOBOUNDS"):
        assert after[0] - before[0] > 100, (before, after)             # most launches of the queues had a plan
    assert dev.selfcheck_failures == 0
    dev.close()


def test_k1_tile_boxes_selfcheck():
    """Re-run the distance tests with SLAMHIP_K1_VERIFY=1 (every end point is checked against its LDS tile
    box, every staged pixel against the map), with SLAMHIP_K1_GLOBAL=1 (global-gather fallback kernels) and
    with tile budgets / layouts that force every step kind: all must stay bit-exact."""
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    sel = "test_distance_golden or test_distance_quirks or test_distance_64bit or test_search_full_size or test_search_changing_scans or test_search_enqueue_ring or test_heading_lattice or test_search_plan_launches"
    for env_extra in ({"SLAMHIP_K1_VERIFY": "1", "SLAMHIP_K1_CPL": "1"}, {"SLAMHIP_K1_VERIFY": "1", "SLAMHIP_K1_CPL": "2"},
                      {"SLAMHIP_K1_LAYOUT_SYNC": "1", "SLAMHIP_K1_VERIFY": "1"},   # every scan's launch layout made before its launch
                      {"SLAMHIP_K1_VERIFY": "1", "SLAMHIP_K1_CPL": "4"}, {"SLAMHIP_K1_GLOBAL": "1"},
                      {"SLAMHIP_K1_TILE_KB": "8", "SLAMHIP_K1_VERIFY": "1"},       # banded tiles and global gathers
                      {"SLAMHIP_K1_TILE_KB": "24", "SLAMHIP_K1_CPL": "1"}, {"SLAMHIP_K1_TILE_KB": "1"},
                      {"SLAMHIP_K1_NOTABLE": "1"},                                 # uniform chunk-major layout
                      {"SLAMHIP_K1_NODEN": "1", "SLAMHIP_K1_VERIFY": "1"},         # tile addresses from the integer coordinates everywhere
                      {"SLAMHIP_K1_GROUP": "2048", "SLAMHIP_K1_VERIFY": "1"},      # groups of 2048 candidates (512 lanes x 4), as large searches use
                      {"SLAMHIP_K1_GROUP": "2048"},
                      {"SLAMHIP_K1_GROUP": "512", "SLAMHIP_K1_VERIFY": "1"},       # groups of 512 candidates (512 lanes x 1), as small searches use
                      {"SLAMHIP_K1_GROUP": "512"}, {"SLAMHIP_K1_GROUP": "1024"},
                      {"SLAMHIP_K1_NOBOUNDS": "1", "SLAMHIP_K1_VERIFY": "1"},      # search-mode bounds from the in-kernel reduction
                      {"SLAMHIP_K1_CUT_ALWAYS": "1", "SLAMHIP_K1_VERIFY": "1"},    # ray ranges cut by cost from the first launch of a scan on
                      {"SLAMHIP_K1_CUT_ALWAYS": "1", "SLAMHIP_K1_CUT_TAB": "1", "SLAMHIP_K1_CUT_WFIX": "40"},   # ... the listed groups' too, heavy weights
                      {"SLAMHIP_K1_CUT_ALWAYS": "1", "SLAMHIP_K1_CUT_WFIX": "3", "SLAMHIP_K1_CUT_WKB": "1.5", "SLAMHIP_K1_CUT_KEEP": "100"},
                      {"SLAMHIP_K1_CUT_WFIX": "0"},                                # the equal-count formula everywhere
                      {"SLAMHIP_K1_NOSPLIT": "1", "SLAMHIP_K1_VERIFY": "1"},       # banded pieces never planned as two halves
                      {"SLAMHIP_K1_NOPAD": "1", "SLAMHIP_K1_VERIFY": "1"},         # tiles as wide as their box
                      {"SLAMHIP_K1_TILE_KB": "16", "SLAMHIP_K1_VERIFY": "1"},      # many banded pieces: the halves' plans under the self-check
                      {"SLAMHIP_K1_TILE_KB": "16"},                                # ... and the lattice kernel over them
                      {"SLAMHIP_NO_DIRECT_UPLOAD": "1"},                           # scans through the upload launch (no CPU stores into device memory)
                      {"SLAMHIP_K1_PLAN": "0"},                                    # no plan launches: every workgroup plans for itself
                      {"SLAMHIP_K1_PLAN_ALWAYS": "1", "SLAMHIP_K1_VERIFY": "1"},   # a plan launch beside EVERY search: most arrive while their search runs
                      {"SLAMHIP_K1_PLAN_ALWAYS": "1", "SLAMHIP_K1_TILE_KB": "16"},  # ... with many banded pieces (records of many steps, some too long for a record)
                      {"SLAMHIP_K1_TARGET_WGS": "64", "SLAMHIP_K1_TARGET_WGS_UNIFORM": "64"},
                      {"SLAMHIP_K1_TARGET_WGS": "100000", "SLAMHIP_K1_TARGET_WGS_UNIFORM": "100000", "SLAMHIP_K1_CPL": "1"}):
        env = dict(os.environ); env.update(env_extra); env["SLAMHIP_EXPECT_SELFCHECK"] = "1"
        r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(here, "test_gpu_coreslam.py"), "-m", "gpu", "-x", "-q",
                            "-k", sel], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
        assert r.returncode == 0, (env_extra, r.stdout.decode(errors="replace")[-3000:])


# ---- K2 HoleMap ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", ["cs_holemap_64_r90.npz", "cs_holemap_256_r360_hw2.npz"])
def test_holemap_golden(cs_mod, ctx, name):
    g = np.load(os.path.join(GOLD, name))
    dev = make_dev(cs_mod, ctx, int(g["size"]))
    for i in range(g["xy"].shape[0]):
        dev.set_scan(g["xy"][i])
        dev.update_holemap_pxcs(g["pxcs"][i], float(g["hole_width"]), int(g["quality"]))
        assert dev.last_holemap_pixels == int(g["counts"][i])
        if i == 0:
            assert (dev.holemap_download() == g["after1"]).all()
    assert (dev.holemap_download() == g["after_all"]).all()
    assert (dev.holemap_download_packed() == ((g["after_all"][0::2] >> 12) << 4 | (g["after_all"][1::2] >> 12))).all()
    dev.close()


@pytest.mark.parametrize("size,R,hw,pose", [
    (400, 360, 0.6, (20.0, 20.0, 0.3)),
    (1024, 1080, 0.6, (12.5, 30.2, -2.0)),
    (2048, 1080, 0.6, (20.0, 20.0, 0.0)),
    (2048, 1080, 2.0, (20.3, 19.1, 1.1)),
    (512, 720, 0.6, (34.9, 20.0, 1.0)),       # close to the east wall: dense overlapping hole zones
    (256, 1080, 5.0, (6.0, 6.0, 0.77)),       # corner, very wide holes: many conflicting fragments
    (300, 500, 0.6, (39.9, 39.9, 2.0)),       # robot at the map edge: clipping on most rays
    (2048, 2000, 0.6, (5.5, 5.5, 0.4)),
    (1024, 4001, 0.6, (20.0, 20.0, 0.2)),     # more rays than the pixel kernel keeps in LDS: global ray table
    (16392, 360, 0.6, (20.0, 20.0, 0.1)),     # sides above 16384: 64-bit hit test
    (128, 90, 3.0e7, (20.0, 20.0, 0.5)),      # absurd hole width (half-width > 2^24 px): literal wrapping recurrence
    (128, 90, 2.0e9, (21.0, 20.0, 0.5)),      # extension end beyond the int32 range
])
def test_holemap_vs_oracle(cs_mod, ctx, det, sim, size, R, hw, pose):
    oc = det
    segs = sim.default_field()
    dev = make_dev(cs_mod, ctx, size)
    ref = np.full(size * size, 32750, np.uint16)
    rng = sim.PCG32(size + R)
    for it in range(4):
        p = np.array([pose[0] + 0.07 * it, pose[1] - 0.05 * it, pose[2] + 0.03 * it], np.float32)
        rays, xy = sim.make_scan(segs, p, R, rng)
        dev.set_scan(xy)
        dev.update_holemap(p, hw, 50)
        n = oc.update_holemap(ref, size, dev.hole_scale, xy, p, hw, 50)
        assert dev.last_holemap_pixels == n
        got = dev.holemap_download()
        bad = np.flatnonzero(got != ref)
        assert bad.size == 0, (it, bad[:10], got[bad[:10]], ref[bad[:10]])
    dev.close()


def test_holemap_unordered_dense_scan(cs_mod, ctx, det):
    """Rays in random order and far denser than one per pixel: most pixels collect more than four fragments whose
    blend order is the (random) ray index -- the conflict list, the rank sort and the all-rays scan of the zone."""
    oc = det
    size = 256
    dev = make_dev(cs_mod, ctx, size)
    ref = np.full(size * size, 32750, np.uint16)
    rng = np.random.default_rng(11)
    for it in range(3):
        ang = rng.uniform(-np.pi, np.pi, 3000)
        rad = rng.uniform(0.5, 18.0, 3000)
        xy = np.stack([rad * np.cos(ang), rad * np.sin(ang)], 1).astype(np.float32)
        pose = [20.0 + it, 19.0, 0.3 * it]
        dev.set_scan(xy)
        dev.update_holemap(pose, 1.5, 77)
        n = oc.update_holemap(ref, size, dev.hole_scale, xy, pose, 1.5, 77)
        assert dev.last_holemap_pixels == n
        assert (dev.holemap_download() == ref).all()
    dev.close()


@pytest.mark.parametrize("size,R,fov,shuffle,rmin,rmax,hw", [
    (2048, 1080, 360.0, False, 0.3, 1.2, 0.6),     # every obstacle INSIDE the zone: central and zone pixels inside the rays' V (marked pixels, the workgroups' queues, the lanes' own ordered draws)
    (2048, 1080, 360.0, True, 2.0, 19.0, 0.6),     # a full scan in random ray order: index order is not angular order (the selection is by direction, the blend order by index)
    (1024, 720, 180.0, False, 1.0, 15.0, 0.6),     # half a circle: four octants hold no ray at all
    (2048, 1000, 45.0, False, 3.0, 18.0, 0.6),     # one octant holds every ray: its rays' far steps are one XCD's, the own-ray records overflow nothing
    (2048, 2040, 30.0, True, 3.0, 18.0, 0.6),      # ... and more own rays than records fit side by side (the indirect own list), shuffled
    (512, 64, 360.0, False, 0.2, 19.0, 2.5),       # a handful of rays, wide holes
    (2048, 1080, 360.0, False, 2.0, 19.0, -0.6),   # a NEGATIVE hole width turns the extension round: no arcs (every workgroup holds every ray)
])
def test_holemap_arcs_inputs(cs_mod, ctx, det, size, R, fov, shuffle, rmin, rmax, hw):
    """Round 5's HoleMap update gives every workgroup the rays of its octant and a margin, chosen by the direction of the float end
    point (k2_arc_member, holemap.hip).  Scans that stress that choice -- against the oracle, pixel for pixel, three updates each from
    poses that turn and move (the octants' borders sweep over the rays)."""
    oc = det
    dev = make_dev(cs_mod, ctx, size)
    ref = np.full(size * size, 32750, np.uint16)
    rng = np.random.default_rng(size + R)
    for it in range(3):
        ang = np.radians(np.linspace(0.0, fov, R, endpoint=False) + rng.uniform(-0.02, 0.02, R) + 17.0 * it)
        rad = rng.uniform(rmin, rmax, R)
        xy = np.stack([rad * np.cos(ang), rad * np.sin(ang)], 1).astype(np.float32)
        if shuffle:
            xy = xy[rng.permutation(R)]
        pose = [20.0 + 0.4 * it, 19.5 - 0.3 * it, 0.7 * it - 0.4]
        dev.set_scan(xy)
        dev.update_holemap(pose, hw, 60)
        n = oc.update_holemap(ref, size, dev.hole_scale, xy, pose, hw, 60)
        assert dev.last_holemap_pixels == n
        got = dev.holemap_download()
        bad = np.flatnonzero(got != ref)
        assert bad.size == 0, (it, bad.size, bad[:8], got[bad[:8]], ref[bad[:8]])
    dev.close()


@pytest.mark.parametrize("env", [{"SLAMHIP_K2_NCORE": "0"}, {"SLAMHIP_K2_ZONE": "48", "SLAMHIP_K2_RB": "5"}, {"SLAMHIP_K2_ZONE": "200", "SLAMHIP_K2_RB": "30"},
                                 {"SLAMHIP_K2_GRID": "64"}])
def test_holemap_variants(env):
    """The HoleMap update's other forms on the ordinary test scans: without arcs (every workgroup holds every ray: what a kept host
    mirror, a developer grid or an absurd hole width select), other zone radii and central radii, a grid of 64 workgroups (seven
    sector workgroups per XCD: several zone items each, full queues)."""
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(here, "test_gpu_coreslam.py"), "-m", "gpu", "-x", "-q", "-k",
                        "holemap_golden or holemap_vs_oracle or holemap_unordered or holemap_degenerate or holemap_arcs_inputs or search_and_update_fused or fused_scans"],
                       env=dict(os.environ, **env), stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    assert r.returncode == 0, (env, r.stdout.decode(errors="replace")[-3000:])


def test_holemap_degenerate_inputs(cs_mod, ctx, det):
    oc = det
    size = 128
    dev = make_dev(cs_mod, ctx, size)
    scale = dev.hole_scale
    rng = np.random.default_rng(3)
    cases = [
        np.array([[0.0, 0.0], [2.0, 1.0], [np.nan, 1.0], [1e30, 0.0], [-1e20, 1e20]], np.float32),   # D1
        np.array([[500.0, 300.0], [-700.0, 650.0], [0.01, 0.0], [1e-9, 1e-9]], np.float32),          # far outside + tiny
        rng.uniform(-60, 60, (300, 2)).astype(np.float32),
        np.tile(np.array([[5.0, 0.0]], np.float32), (50, 1)),                                       # 50 identical rays
    ]
    for pose in ([20, 20, 0.0], [0.1, 0.1, 0.5], [-3.0, 20.0, 0.0], [39.99, 0.0, 3.0]):
        ref = np.full(size * size, 32750, np.uint16)
        dev.reset()
        for xy in cases:
            dev.set_scan(xy)
            dev.update_holemap(pose, 0.6, 50)
            n = oc.update_holemap(ref, size, scale, xy, pose, 0.6, 50)
            assert dev.last_holemap_pixels == n
            assert (dev.holemap_download() == ref).all()
    # alpha extremes (Quality 1..255, :74-80) and ties dx == dy
    for q in (1, 128, 255):
        ref = np.full(size * size, 32750, np.uint16)
        dev.reset()
        xy = np.array([[3.0, 3.0], [3.0, -3.0], [-3.0, 3.0], [0.0, 4.0], [4.0, 0.0]], np.float32)
        dev.set_scan(xy)
        for _ in range(3):
            dev.update_holemap([20, 20, 0.0], 0.6, q)
            oc.update_holemap(ref, size, scale, xy, [20, 20, 0.0], 0.6, q)
        assert (dev.holemap_download() == ref).all()
    dev.close()


# ---- K3 ObstacleMap -------------------------------------------------------------------------------------------
def test_obstacle_golden(cs_mod, ctx):
    g = np.load(os.path.join(GOLD, "cs_obstacle_64_r360.npz"))
    dev = cs_mod.CoreSlamDevice(ctx, 40.0, 64, int(g["size"]))
    assert dev.obst_scale == float(g["scale"])
    for i in range(g["xy"].shape[0]):
        dev.set_scan(g["xy"][i])
        dev.update_obstaclemap_pxcs(g["pxcs"][i], int(g["max_hits"]))
        if i == 0:
            assert (dev.obstaclemap_download() == g["after1"]).all()
    assert (dev.obstaclemap_download() == g["after_all"]).all()
    dev.close()


@pytest.mark.parametrize("osize,R", [(64, 400), (100, 360), (512, 1080), (1024, 1080)])
def test_obstacle_vs_oracle(cs_mod, ctx, det, sim, osize, R):
    oc = det
    segs = sim.default_field()
    dev = cs_mod.CoreSlamDevice(ctx, 40.0, 64, osize)
    ref = np.full((osize, osize), -5, np.int8)
    rng = sim.PCG32(osize)
    for it in range(14):
        p = np.array([20 + 0.2 * it, 20 - 0.1 * it, 0.3 + 0.1 * it], np.float32)
        rays, xy = sim.make_scan(segs, p, R, rng)
        dev.set_scan(xy)
        dev.update_obstaclemap(p, 10)
        oc.update_obstaclemap(ref, osize, dev.obst_scale, xy, p, 10)
        assert (dev.obstaclemap_download() == ref).all()
    assert ref.max() == 10 and ref.min() == -5
    # robot outside the map: no-op (:557-560); garbage points
    dev.update_obstaclemap([-1.0, 5.0, 0.0], 10)
    assert (dev.obstaclemap_download() == ref).all()
    xy = np.array([[np.nan, 0.0], [1e30, 1.0], [0.0, 0.0], [300.0, -200.0]], np.float32)
    dev.set_scan(xy)
    dev.update_obstaclemap([20.0, 20.0, 0.0], 10)
    oc.update_obstaclemap(ref, osize, dev.obst_scale, xy, [20.0, 20.0, 0.0], 10)
    assert (dev.obstaclemap_download() == ref).all()
    dev.close()


# ---- fused path + processor state machine -----------------------------------------------------------------------
def test_search_and_update_fused(cs_mod, ctx, det, sim):
    oc = det
    size, osize, R, K = 1024, 256, 1080, 4096
    segs = sim.default_field()
    dev = cs_mod.CoreSlamDevice(ctx, 40.0, size, osize)
    ref_h = np.full(size * size, 32750, np.uint16)
    ref_o = np.full((osize, osize), -5, np.int8)
    rng = sim.PCG32(11)
    for p in sim.trajectory(6):
        rays, xy = sim.make_scan(segs, p, R, rng)
        dev.set_scan(xy)
        dev.update_holemap(p); dev.update_obstaclemap(p)
        oc.update_holemap(ref_h, size, dev.hole_scale, xy, p); oc.update_obstaclemap(ref_o, osize, dev.obst_scale, xy, p)
    true_pose = sim.trajectory(7)[-1]
    rays, xy = sim.make_scan(segs, true_pose, R, rng)
    base = (true_pose + np.array([0.05, 0.04, 6.25], np.float32)).astype(np.float32)     # theta needs NormalizeAngle
    offs = sim.gaussian_offsets(K - 1, seed=5)
    dev.set_scan(xy); dev.set_offsets(offs)
    pose, dist, idx = dev.search_and_update(base, 0.6, 50, 10)
    rbi, rpose, rbd, _ = oc.search(ref_h, size, dev.hole_scale, xy, base, offs)
    rpose[2] = oc.normalize_angle(rpose[2])                                               # :746
    assert idx == rbi and dist == rbd and (pose == rpose).all()
    n_px = oc.update_holemap(ref_h, size, dev.hole_scale, xy, rpose); oc.update_obstaclemap(ref_o, osize, dev.obst_scale, xy, rpose)
    # the call returns with the pose while the map updates run on: whatever reads the maps next is ordered behind them
    assert dev.last_holemap_pixels == n_px
    assert (dev.holemap_download() == ref_h).all()
    assert (dev.obstaclemap_download() == ref_o).all()
    # a second fused scan straight behind the first (its search reads the map the first one's update is still writing)
    pose2, dist2, idx2 = dev.search_and_update(base, 0.6, 50, 10)
    rbi2, rpose2, rbd2, _ = oc.search(ref_h, size, dev.hole_scale, xy, base, offs)
    rpose2[2] = oc.normalize_angle(rpose2[2])
    assert idx2 == rbi2 and dist2 == rbd2 and (pose2 == rpose2).all()
    oc.update_holemap(ref_h, size, dev.hole_scale, xy, rpose2); oc.update_obstaclemap(ref_o, osize, dev.obst_scale, xy, rpose2)
    assert (dev.holemap_download() == ref_h).all()
    assert (dev.obstaclemap_download() == ref_o).all()
    dev.close()


def test_fused_robot_outside_holemap_only(cs_mod, ctx, det, sim):
    """The fused call with the robot just left of both maps' first column: (int)(x * scale + 0.5) is -3 at the HoleMap's scale
    -- UpdateHoleMap draws nothing (:509-512) -- but truncates to 0 at the ObstacleMap's (:553-560 test its own pixel), whose
    update, riding on the HoleMap update's launch, must still run; and a third scan from well inside behind it."""
    oc = det
    size, osize, R, K = 512, 100, 360, 1024
    segs = sim.default_field()
    dev = cs_mod.CoreSlamDevice(ctx, 40.0, size, osize)
    ref_h = np.full(size * size, 32750, np.uint16)
    ref_o = np.full((osize, osize), -5, np.int8)
    rng = sim.PCG32(23)
    offs = np.zeros((K - 1, 3), np.float32)                              # (every candidate is the base pose: the winner is known)
    dev.set_offsets(offs)
    for base in (np.array([20.0, 20.0, 0.3], np.float32), np.array([-0.3, 18.07, -0.9], np.float32), np.array([-0.3, 18.07, -0.9], np.float32),
                 np.array([21.0, 19.0, 1.0], np.float32)):
        _, xy = sim.make_scan(segs, np.array([20.0, 20.0, float(base[2])], np.float32), R, rng)
        dev.set_scan(xy)
        pose, dist, idx = dev.search_and_update(base, 0.6, 50, 10)
        rbi, rpose, rbd, _ = oc.search(ref_h, size, dev.hole_scale, xy, base, offs)
        rpose[2] = oc.normalize_angle(rpose[2])
        assert idx == rbi and dist == rbd and (pose == rpose).all()
        before = ref_o.copy()
        n_px = oc.update_holemap(ref_h, size, dev.hole_scale, xy, rpose); oc.update_obstaclemap(ref_o, osize, dev.obst_scale, xy, rpose)
        if base[0] < 0:
            assert n_px == 0 and (before != ref_o).any()                 # the case this test is about
    assert (dev.holemap_download() == ref_h).all()
    assert (dev.obstaclemap_download() == ref_o).all()
    dev.close()


def test_fused_scans_back_to_back(cs_mod, ctx, det, sim):
    """Twelve scans through set_scan + the fused call with nothing in between that waits for the device: every call
    returns with its pose while its map updates still run, the next scan's upload and search queue behind them.  Poses
    scan by scan and both maps at the end equal the oracle's (2048^2: the updates outlast the host's preparation)."""
    oc = det
    size, osize, R, K = 2048, 512, 1080, 2048
    segs = sim.default_field()
    dev = cs_mod.CoreSlamDevice(ctx, 40.0, size, osize)
    ref_h = np.full(size * size, 32750, np.uint16)
    ref_o = np.full((osize, osize), -5, np.int8)
    rng = sim.PCG32(17)
    traj = sim.trajectory(18)
    scans = [sim.make_scan(segs, p, R, rng)[1] for p in traj]
    for p, xy in zip(traj[:6], scans[:6]):
        dev.set_scan(xy)
        dev.update_holemap(p); dev.update_obstaclemap(p)
        oc.update_holemap(ref_h, size, dev.hole_scale, xy, p); oc.update_obstaclemap(ref_o, osize, dev.obst_scale, xy, p)
    offs = sim.gaussian_offsets(K - 1, 0.05, math.radians(2.0), seed=9)
    dev.set_offsets(offs)
    got = []
    bases = [(p + np.array([0.02, -0.03, 0.01], np.float32)).astype(np.float32) for p in traj[6:]]
    for base, xy in zip(bases, scans[6:]):
        dev.set_scan(xy)
        got.append(dev.search_and_update(base, 0.6, 50, 10))
    for (pose, dist, idx), base, xy in zip(got, bases, scans[6:]):
        rbi, rpose, rbd, _ = oc.search(ref_h, size, dev.hole_scale, xy, base, offs)
        rpose[2] = oc.normalize_angle(rpose[2])
        assert idx == rbi and dist == rbd and (pose == rpose).all()
        oc.update_holemap(ref_h, size, dev.hole_scale, xy, rpose); oc.update_obstaclemap(ref_o, osize, dev.obst_scale, xy, rpose)
    assert (dev.holemap_download() == ref_h).all()
    assert (dev.obstaclemap_download() == ref_o).all()
    dev.close()


def test_fused_completion_modes():
    """The fused call and the processor with the result block after the updates (SLAMHIP_FUSED_WAIT_UPDATES=1), without the host
    mailbox (SLAMHIP_NO_HOSTWAIT=1: copy + synchronise), on the fallback search kernels and with the pose delivered by the search's
    final arriver (SLAMHIP_FUSED_K1_DELIVERS=1): same results as the default (the winner decoded and delivered by the map update)."""
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    sel = "test_search_and_update_fused or test_processor_vs_oracle or test_fused_scans_back_to_back"
    for env_extra in ({"SLAMHIP_FUSED_WAIT_UPDATES": "1"}, {"SLAMHIP_NO_HOSTWAIT": "1"}, {"SLAMHIP_K1_GLOBAL": "1"}, {"SLAMHIP_FUSED_K1_DELIVERS": "1"}):
        env = dict(os.environ); env.update(env_extra)
        r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(here, "test_gpu_coreslam.py"), "-m", "gpu", "-x", "-q",
                            "-k", sel], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
        assert r.returncode == 0, (env_extra, r.stdout.decode(errors="replace")[-3000:])


def test_processor_vs_oracle(cs_mod, ctx, det, sim):
    """CoreSLAMProcessor.Update over 12 scans (5 map-only, then searching) vs the oracle state machine."""
    oc = det
    segs = sim.default_field()
    start = np.array([20.0, 20.0, 0.0], np.float32)
    proc = cs_mod.CoreSLAMProcessor(40.0, 256, 64, start, 0.1, math.radians(10), 500, 4, ctx=ctx)
    proc.HoleWidth = 2.0                                                # Simulation/MainWindow.xaml.cs:69-72
    ref = oc.CSProc(40.0, 256, 64, start)
    ref.set_params(hole_width=2.0)
    rng = sim.PCG32(21)
    true_traj = sim.trajectory(12, step=(0.08, 0.03, math.radians(1.0)))
    for i, tp in enumerate(true_traj):
        rays, _ = sim.make_scan(segs, tp, 400, rng)
        est = proc.Pose.copy()                                          # the simulator stores the last estimate (:159,:387)
        assert (est == ref.pose).all()
        seg_pose = (est + np.array([0.08, 0.03, math.radians(1.0)], np.float32)).astype(np.float32) if i else est
        offs = sim.gaussian_offsets(2000, seed=100 + i)
        proc.SetOffsets(offs)
        proc.Update([cs_mod.ScanSegment(rays, seg_pose)])
        ref.update(seg_pose[None], [0, rays.shape[0]], rays, offs)
        assert (proc.Pose == ref.pose).all(), i
        assert (proc.HoleMap.Pixels == ref.holemap).all(), i
        assert (proc.ObstacleMap.Pixels == ref.obstaclemap).all(), i
    err = proc.Pose - true_traj[-1]
    assert abs(err[0]) < 0.5 and abs(err[1]) < 0.5                      # sanity: it localises
    # two segments per scan + Reset
    proc.Reset(); ref.reset()
    rays, _ = sim.make_scan(segs, (20, 20, 0), 200, rng)
    segp = np.array([[19.9, 20.0, -0.01], [20.0, 20.0, 0.0]], np.float32)
    proc.Update([cs_mod.ScanSegment(rays[:100], segp[0], False), cs_mod.ScanSegment(rays[100:], segp[1], True)])
    ref.update(segp, [0, 100, 200], rays, None)
    assert (proc.Pose == ref.pose).all() and (proc.HoleMap.Pixels == ref.holemap).all()
    # device-generated candidates: runs, stays close to the odometry prediction
    proc.Reset()
    proc2 = cs_mod.CoreSLAMProcessor(40.0, 256, 64, start, 0.1, math.radians(10), 1000, 4, ctx=ctx)
    for i, tp in enumerate(true_traj):
        rays, _ = sim.make_scan(segs, tp, 400, rng)
        proc2.Update([cs_mod.ScanSegment(rays, proc2.Pose)])
    assert np.isfinite(proc2.Pose).all()
    proc2.Dispose(); proc.Dispose()


def test_processor_search_launched_ahead_of_the_scan(cs_mod, ctx, det, sim):
    """CoreSLAMProcessor.Update (CoreSLAMProcessor.cs:717-752) with the search launch put into the stream before the scan's tables
    (cs_search_and_update_prelaunched): one candidate list kept over many scans of one ray count, so that the launch-ahead path is
    taken; in between scans whose ray blocks the last layout does not serve (every ray far from the next: one block per ray) and a
    scan with a NaN point (the search then runs on the bounds-checked kernels) -- those launches are abandoned on the device and the
    scan is searched again in the ordinary order.  Pose and both maps equal the oracle state machine after every scan; the
    library's counters say that every path was taken."""
    oc = det
    segs = sim.default_field()
    start = np.array([20.0, 20.0, 0.0], np.float32)
    proc = cs_mod.CoreSLAMProcessor(40.0, 1024, 128, start, 0.1, math.radians(10), 500, 4, ctx=ctx)
    proc.HoleWidth = 2.0
    ref = oc.CSProc(40.0, 1024, 128, start)
    ref.set_params(hole_width=2.0)
    rng = sim.PCG32(33)
    R = 720
    offs = sim.gaussian_offsets(4000, seed=77)
    proc.SetOffsets(offs)
    true_traj = sim.trajectory(40, step=(0.05, 0.02, math.radians(0.4)))
    nprng = np.random.default_rng(5)
    for i, tp in enumerate(true_traj):
        rays, _ = sim.make_scan(segs, tp, R, rng)
        if i in (17, 18, 29):                                           # a scan scattered all over the room: every ray its own block
            rays = rays.copy()
            rays[:, 0] = nprng.uniform(-math.pi, math.pi, R).astype(np.float32)
            rays[:, 1] = nprng.uniform(0.5, 18.0, R).astype(np.float32)
        if i == 24:
            rays = rays.copy(); rays[5, 1] = np.nan
        est = proc.Pose.copy()
        assert (est == ref.pose).all()
        seg_pose = (est + np.array([0.05, 0.02, math.radians(0.4)], np.float32)).astype(np.float32) if i else est
        proc.Update([cs_mod.ScanSegment(rays, seg_pose)])
        ref.update(seg_pose[None], [0, rays.shape[0]], rays, offs)
        assert (proc.Pose == ref.pose).all() or (np.isnan(proc.Pose).any() and np.isnan(ref.pose).any()), (i, proc.Pose, ref.pose)
        if i % 4 == 3 or i in (17, 18, 19, 24, 25, 29, 30):
            assert (proc.HoleMap.Pixels == ref.holemap).all(), i
            assert (proc.ObstacleMap.Pixels == ref.obstaclemap).all(), i
    ahead, abandoned, remade, refused = proc.device.prelaunch_stats
    if os.environ.get("SLAMHIP_PRELAUNCH", "1") != "0" and not os.environ.get("SLAMHIP_NO_HOSTWAIT"):
        assert ahead >= 10 and abandoned >= 1 and refused >= 1, (ahead, abandoned, remade, refused)
    assert proc.device.selfcheck_failures == 0
    proc.Dispose()


def test_scan_search_and_update_equals_the_two_calls(cs_mod, ctx, sim):
    """slamhip_cs_scan_search_and_update == slamhip_cs_set_scan + slamhip_cs_search_and_update (CoreSLAMProcessor.cs:723, :732,
    :746-751): two operators fed the same scans and the same candidate list, one through the two calls, one through the one call
    (whose search launch precedes the scan's tables from the second scan on): pose, distance, index and both maps equal after
    every scan, and the counters say the launch-ahead path was taken."""
    segs = sim.default_field()
    a = cs_mod.CoreSlamDevice(ctx, 40.0, 1024, 256)
    b = cs_mod.CoreSlamDevice(ctx, 40.0, 1024, 256)
    rng = sim.PCG32(91)
    traj = sim.trajectory(16, step=(0.06, 0.02, math.radians(0.5)))
    offs = sim.gaussian_offsets(8191, 0.1, math.radians(8.0), seed=5)
    for d in (a, b):
        d.set_offsets(offs)
    for i, p in enumerate(traj):
        _, xy = sim.make_scan(segs, p, 900, rng)
        if i < 4:                                                          # (mapping first)
            for d in (a, b):
                d.set_scan(xy); d.update_holemap(p, 0.6, 50); d.update_obstaclemap(p, 10)
            continue
        search = (p + np.array([0.03, -0.02, math.radians(0.8)], np.float32)).astype(np.float32)
        a.set_scan(xy)
        pa, da, ia = a.search_and_update(search, 0.6, 50, 10)
        pb, db, ib = b.scan_search_and_update(xy, search, 0.6, 50, 10)
        assert (pa == pb).all() and da == db and ia == ib, (i, pa, pb, da, db, ia, ib)
        if i % 3 == 0 or i == len(traj) - 1:
            assert (a.holemap_download() == b.holemap_download()).all(), i
            assert (a.obstaclemap_download() == b.obstaclemap_download()).all(), i
    ahead, abandoned, remade, refused = b.prelaunch_stats
    if os.environ.get("SLAMHIP_PRELAUNCH", "1") != "0" and not os.environ.get("SLAMHIP_NO_HOSTWAIT"):
        assert ahead >= 6, (ahead, abandoned, remade, refused)
    assert a.prelaunch_stats[0] == 0
    a.close(); b.close()


def test_processor_long_run_vs_oracle(cs_mod, ctx, det, sim):
    """The simulator's loop (Simulation/MainWindow.xaml.cs:136-210) headless for 160 scans around the inner obstacle: the
    estimate after every Update and both maps along the way must equal the oracle state machine's bit for bit -- any
    single differing pixel or a lost tie-break would compound over the following scans."""
    oc = det
    segs = sim.default_field()
    traj, _ = sim.lap_trajectory(150, 0.1)
    traj = np.concatenate([np.repeat(traj[:1], 10, axis=0), traj[1:]])
    start = traj[0].copy()
    proc = cs_mod.CoreSLAMProcessor(40.0, 256, 64, start, 0.1, math.radians(10), 500, 4, ctx=ctx)
    proc.HoleWidth = 2.0
    ref = oc.CSProc(40.0, 256, 64, start)
    ref.set_params(hole_width=2.0)
    rng = sim.PCG32(5)
    for i, tp in enumerate(traj):
        rays, _ = sim.make_scan(segs, tp, 400, rng)
        est = proc.Pose.copy()
        offs = sim.gaussian_offsets(2000, seed=1000 + i)
        proc.SetOffsets(offs)
        proc.Update([cs_mod.ScanSegment(rays, est)])                   # the simulator poses the segment at the last estimate
        ref.update(est[None], [0, rays.shape[0]], rays, offs)
        assert (proc.Pose == ref.pose).all(), i
        if i % 20 == 19 or i == len(traj) - 1:
            assert (proc.HoleMap.Pixels == ref.holemap).all(), i
            assert (proc.ObstacleMap.Pixels == ref.obstaclemap).all(), i
    err = proc.Pose - traj[-1]
    assert math.hypot(err[0], err[1]) < 0.3 and abs(err[2]) < math.radians(3)      # and it tracks the true pose
    proc.Dispose()


def test_group_single_gpu(cs_mod, ctx, det, sim):
    """slamhip_group_* (the single-process multi-GPU form: RCCL communicator, sharded search, replicated updates) with a
    one-GPU group: the same answers as the plain operator object and the oracle."""
    import ctypes as C
    import slam.net_amd.capi as capi
    oc = det
    size, R, K = 512, 720, 5000
    segs = sim.default_field()
    rng = sim.PCG32(77)
    g = C.c_void_p()
    dev_ids = (C.c_int32 * 1)(0)
    capi.call("slamhip_group_create", dev_ids, 1, C.c_float(40.0), size, 64, C.byref(g))
    try:
        n = C.c_int32()
        capi.call("slamhip_group_size", g, C.byref(n))
        assert n.value == 1
        capi.call("slamhip_group_reset", g, -5)
        ref = np.full(size * size, 32750, np.uint16)
        scale = size / 40.0
        for p in sim.trajectory(6)[:-1]:
            _, xy = sim.make_scan(segs, p, R, rng)
            capi.call("slamhip_group_set_scan", g, capi.fptr(capi.f32(xy)), xy.shape[0])
            capi.call("slamhip_group_update_maps", g, capi.fptr(capi.f32(p)), C.c_float(0.6), 50, 60)
            oc.update_holemap(ref, size, scale, xy, p, 0.6, 50)
        pose = sim.trajectory(6)[-1]
        _, xy = sim.make_scan(segs, pose, R, rng)
        base = (pose + np.array([0.02, -0.03, 0.01], np.float32)).astype(np.float32)
        offs = sim.gaussian_offsets(K - 1)
        capi.call("slamhip_group_set_scan", g, capi.fptr(capi.f32(xy)), xy.shape[0])
        capi.call("slamhip_group_set_offsets", g, capi.fptr(capi.f32(offs)), offs.shape[0])
        out_pose = np.zeros(3, np.float32)
        dist, idx = C.c_int32(), C.c_int32()
        capi.call("slamhip_group_search", g, capi.fptr(base), capi.fptr(out_pose), C.byref(dist), C.byref(idx))
        rbi, rpose, rbd, _ = oc.search(ref, size, scale, xy, base, offs)
        assert idx.value == rbi and dist.value == rbd and (out_pose == rpose).all()
        # the whole scan in one call: search, exchange, winner decoded on the device, the replica's map updates behind it
        fp = np.zeros(3, np.float32)
        capi.call("slamhip_group_search_and_update", g, capi.fptr(base), C.c_float(0.6), 50, 60, capi.fptr(fp), C.byref(dist), C.byref(idx))
        wp = np.array([rpose[0], rpose[1], oc.normalize_angle(float(rpose[2]))], np.float32)
        assert idx.value == rbi and dist.value == rbd and (fp == wp).all()
        oc.update_holemap(ref, size, scale, xy, wp, 0.6, 50)
        cs0 = C.c_void_p()
        capi.call("slamhip_group_cs", g, 0, C.byref(cs0))
        got = np.empty(size * size, np.uint16)
        capi.call("slamhip_cs_holemap_download", cs0, got.ctypes.data_as(C.POINTER(C.c_uint16)), got.size)
        assert (got == ref).all()
        # a list GENERATED on every GPU of the group (what the C# shim's multi-GPU constructor asks for per scan): the same scan once more
        capi.call("slamhip_group_generate_offsets", g, K - 1, C.c_float(0.1), C.c_float(0.17), C.c_uint64(9), C.c_uint64(3))
        gen = np.empty((K - 1, 3), np.float32)
        capi.call("slamhip_cs_offsets_download", cs0, capi.fptr(gen), K - 1)
        capi.call("slamhip_group_search_and_update", g, capi.fptr(base), C.c_float(0.6), 50, 60, capi.fptr(fp), C.byref(dist), C.byref(idx))
        rbi, rpose, rbd, _ = oc.search(ref, size, scale, xy, base, gen)
        wp = np.array([rpose[0], rpose[1], oc.normalize_angle(float(rpose[2]))], np.float32)
        assert idx.value == rbi and dist.value == rbd and (fp == wp).all()
        oc.update_holemap(ref, size, scale, xy, wp, 0.6, 50)
        capi.call("slamhip_cs_holemap_download", cs0, got.ctypes.data_as(C.POINTER(C.c_uint16)), got.size)
        assert (got == ref).all()
        eq = C.c_int32()
        capi.call("slamhip_group_replicas_equal", g, C.byref(eq))
        assert eq.value == 1
    finally:
        capi.call("slamhip_group_destroy", g)


def test_bench_two_ranks_share_one_gpu():
    """bench.py's N > 1 flow (shard ranges, per-step key all-reduce on the library's stream, max-over-ranks timing, the
    per-launch timing pass) with two ranks on this box's single GPU and gloo carrying the key: the JSON line is complete
    and the winner equals that of one rank searching the whole 2 x 4096 list."""
    import json
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    env = dict(os.environ, SLAMHIP_BENCH_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    common = ["--steps", "20", "--warmup", "3", "--size", "1024", "--map-updates", "8", "--no-cpu-baseline"]
    r2 = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                         "--master-port", str(port), os.path.join(root, "bench.py"), "--gpus", "2", "--cands", "4096"] + common,
                        env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert r2.returncode == 0, r2.stderr.decode(errors="replace")[-3000:]
    line2 = [l for l in r2.stdout.decode().splitlines() if l.startswith("{")]
    assert len(line2) == 1                                             # rank 0 prints exactly one JSON line
    j2 = json.loads(line2[0])
    r1 = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--cands", "8192"] + common,
                        stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert r1.returncode == 0, r1.stderr.decode(errors="replace")[-3000:]
    j1 = json.loads([l for l in r1.stdout.decode().splitlines() if l.startswith("{")][0])
    assert j2["n_gpus"] == 2 and j2["scaling"] == "weak" and j2["config"]["candidates_total"] == 8192
    assert j2["config"]["collective"].startswith("gloo")
    assert (j2["config"]["best_index"], j2["config"]["best_distance"]) == (j1["config"]["best_index"], j1["config"]["best_distance"])
    assert j2["roofline"]["launches"] == 50 and j2["roofline"]["avg_launch_us"] > 0 and j2["value"] > 0
    assert j2["multi_gpu"]["replicas_equal"] is True                   # (both ranks built their HoleMap by the same updates)
    assert j2["config"]["winner_matches_oracle"] is True               # (N > 1: rank 0 checks the reduced key against one oracle search over the whole list)
    assert j1["roofline"]["launches"] == 20
    # ONE form at every N: `value` is the enqueue-only search (+ the exchange) at N = 2 as at N = 1, and the blocking per-scan figure
    # stands beside it at both
    m = j2["multi_gpu"]
    for k in ("us_per_step", "single_rank_same_form_us_per_step", "efficiency_same_form", "per_scan_blocking_us_per_step"):
        assert m[k] > 0, k
    assert abs(m["us_per_step"] - j2["ms_per_step"] * 1e3) < 1e-6
    assert j2["config"]["per_scan_blocking_us_per_step"] > 0 and j1["config"]["per_scan_blocking_us_per_step"] > 0
    assert "its key equals the timed region's: True" in j1["config"]["per_scan_blocking_is"]
    assert "its key equals the timed region's: True" in j2["config"]["per_scan_blocking_is"]
    assert "multi_gpu" not in j1
    # the N = 1 line's value is reproduced by --gpus 1 under torch.distributed.run (the launcher the driver uses for N > 1)
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port1 = so.getsockname()[1]
    r1d = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
                          "--master-port", str(port1), os.path.join(root, "bench.py"), "--gpus", "1", "--cands", "8192"] + common,
                         env=dict(os.environ, MASTER_ADDR="127.0.0.1"), stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert r1d.returncode == 0, r1d.stderr.decode(errors="replace")[-3000:]
    j1d = json.loads([l for l in r1d.stdout.decode().splitlines() if l.startswith("{")][0])
    assert (j1d["config"]["best_index"], j1d["config"]["best_distance"]) == (j1["config"]["best_index"], j1["config"]["best_distance"])
    assert j1d["n_gpus"] == 1 and "multi_gpu" not in j1d and j1d["config"]["timed_region"] == j1["config"]["timed_region"]
    assert 0.5 < j1d["value"] / j1["value"] < 2.0                      # (the same form: the same figure, up to the noise of a 20-step region)


def test_fuzz_parity_short():
    """Twenty seconds of tests/fuzz_parity.py (random sizes, poses at and beyond the map edge, 1 .. 2500 rays in random
    order, hole widths, candidate lists with tiny to huge sigmas, Hector pyramids): everything equal to the oracle."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tests", "fuzz_parity.py"), "--seconds", "20", "--seed", "12345"],
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
    out = r.stdout.decode(errors="replace")
    assert r.returncode == 0 and "all equal to the oracle" in out, out[-3000:]



def test_lib_comm_single_rank(cs_mod, ctx, sim):
    """slamhip_comm_* (one process per GPU; here one rank): a search step = K1 + the 8-byte RCCL min all-reduce issued by the
    library on its own stream.  More steps than key slots, shards of every kind, the same keys as the blocking search."""
    import slam.net_amd.distributed as D
    size, R, K = 512, 360, 5000
    dev = make_dev(cs_mod, ctx, size, 64)
    segs = sim.default_field()
    rng = sim.PCG32(5)
    for p in sim.trajectory(6):
        _, xy = sim.make_scan(segs, p, R, rng)
        dev.set_scan(xy)
        dev.update_holemap(p)
    base = sim.trajectory(7)[-1]
    dev.set_offsets(sim.gaussian_offsets(K - 1))
    comm = D.LibComm(ctx, 0, 1)
    try:
        for first, count in ((0, K), (0, K // 3), (K // 3, K - K // 3), (K - 1, 1)):
            want = dev.search_shard(base, first, count)
            step = comm.bind_step(dev, base, first, count)
            for _ in range(19):                                    # (the ring of key slots wraps twice)
                step()
            assert comm.wait() == want
    finally:
        comm.close()
        dev.close()


def test_maps_checksum_and_replica_checks(cs_mod, ctx, det, sim, checksum_np):
    """The replica check (map updates run as replicas on every GPU): slamhip_cs_maps_checksum equals the NumPy
    restatement over the downloaded maps -- also right behind a fused call, whose ObstacleMap cell pass is still pending --, two
    handles fed the same scans agree, one scan more on one of them and they do not; the one-rank communicator and the
    one-GPU group report equal replicas."""
    import ctypes as C
    import slam.net_amd.capi as capi
    import slam.net_amd.distributed as D
    size, osize, R = 512, 100, 360
    segs = sim.default_field()
    a, b = make_dev(cs_mod, ctx, size, osize), make_dev(cs_mod, ctx, size, osize)
    rng = sim.PCG32(3)
    scans = [(p, sim.make_scan(segs, p, R, rng)[1]) for p in sim.trajectory(5)]
    try:
        assert a.maps_checksum() == b.maps_checksum() == (checksum_np(a.holemap_download()), checksum_np(a.obstaclemap_download()))
        for p, xy in scans[:4]:
            for d in (a, b):
                d.set_scan(xy); d.update_holemap(p); d.update_obstaclemap(p)
        ca = a.maps_checksum()
        assert ca == b.maps_checksum() and ca == (checksum_np(a.holemap_download()), checksum_np(a.obstaclemap_download()))
        a.set_offsets(sim.gaussian_offsets(499)); b.set_offsets(sim.gaussian_offsets(499))
        p, xy = scans[4]
        a.set_scan(xy); b.set_scan(xy)
        a.search_and_update(p, 0.6, 50, 10)                         # (returns with the pose: the updates are still in the stream)
        cf = a.maps_checksum()
        assert cf != ca and cf != b.maps_checksum()
        assert cf == (checksum_np(a.holemap_download()), checksum_np(a.obstaclemap_download()))
        b.search_and_update(p, 0.6, 50, 10)
        assert b.maps_checksum() == cf
        comm = D.LibComm(ctx, 0, 1)
        try:
            assert comm.replicas_equal(a) and comm.replicas_equal(b)
            step = comm.bind_step(a, p, 0, 500)                     # (behind asynchronous steps, too)
            step(); step()
            assert comm.replicas_equal(a)
            comm.wait()
        finally:
            comm.close()
    finally:
        a.close(); b.close()
    g = C.c_void_p()
    dev_ids = (C.c_int32 * 1)(0)
    capi.call("slamhip_group_create", dev_ids, 1, C.c_float(40.0), size, osize, C.byref(g))
    try:
        eq = C.c_int32(-1)
        capi.call("slamhip_group_replicas_equal", g, C.byref(eq))
        assert eq.value == 1
    finally:
        capi.call("slamhip_group_destroy", g)


@pytest.mark.parametrize("size,K", [(1024, 16384), (2048, 16384), (400, 4001)])
def test_search_and_update_host_trig(cs_mod, ctx, oc, sim, size, K):
    """slamhip_cs_search_and_update_pxcs: the fused scan with the CALLER's trigonometry at both map scales -- the only form in which a
    .NET host is identical to CoreSLAMProcessor.cs:232-235, :499-502, :545-548 rather than to this library's deterministic routine.
    The oracle runs in TRIG_LIBM mode (glibc's cosf / sinf standing in for the host's MathF), the candidates' (px, py, c, s) come from
    its libm path, and distances' arg-min, index and BOTH maps must be bit-exact over several scans (configurations C2 and C3 sizes,
    and the simulator's)."""
    oc.set_trig_mode(oc.TRIG_LIBM)
    try:
        osize = size // 4
        R = 1080 if size >= 1024 else 360
        dev = cs_mod.CoreSlamDevice(ctx, 40.0, size, osize)
        segs = sim.default_field()
        rng = sim.PCG32(size + K)
        ref_h = np.full(size * size, 32750, np.uint16)
        ref_o = np.full((osize, osize), -5, np.int8)
        traj = sim.trajectory(9)
        for p in traj[:5]:                                            # (mapping updates from the host's own trigonometry)
            _, xy = sim.make_scan(segs, p, R, rng)
            dev.set_scan(xy)
            dev.update_holemap_pxcs(oc.pose_to_pxcs(p, dev.hole_scale)); dev.update_obstaclemap_pxcs(oc.pose_to_pxcs(p, dev.obst_scale))
            oc.update_holemap(ref_h, size, dev.hole_scale, xy, p); oc.update_obstaclemap(ref_o, osize, dev.obst_scale, xy, p)
        assert (dev.holemap_download() == ref_h).all() and (dev.obstaclemap_download() == ref_o).all()
        for it, p in enumerate(traj[5:]):
            _, xy = sim.make_scan(segs, p, R, rng)
            base = (p + np.array([0.03, -0.02, math.radians(1.0)], np.float32)).astype(np.float32)
            offs = sim.gaussian_offsets(K - 1, 0.1, math.radians(10.0), seed=100 + it)
            poses = np.vstack([base[None], (base[None] + offs).astype(np.float32)]).astype(np.float32)      # (:635-637: search + jitter, binary32 adds)
            ps = oc.poses_to_pxcs(poses, dev.hole_scale)
            norm = poses.copy()
            norm[:, 2] = [oc.normalize_angle(a) for a in poses[:, 2]]                                          # (:746: the updates' pose)
            dev.set_scan(xy)
            rbi, rpose, rbd, _ = oc.search(ref_h, size, dev.hole_scale, xy, base, offs)
            if it % 2 == 0:                                              # the one-call form: update rows of every candidate
                ph, po = oc.poses_to_pxcs(norm, dev.hole_scale), oc.poses_to_pxcs(norm, dev.obst_scale)
                idx, dist = dev.search_and_update_pxcs(ps, ph, po, 0.6, 50, 10)
            else:                                                        # the two-call form: the winner's rows formed after the search
                _, idx, dist = dev.distance_pxcs(ps, want_all=False)
                dev.update_maps_pxcs(oc.pose_to_pxcs(norm[idx], dev.hole_scale), oc.pose_to_pxcs(norm[idx], dev.obst_scale), 0.6, 50, 10)
            assert (idx, dist) == (rbi, rbd), (it, idx, dist, rbi, rbd)
            assert (poses[idx] == rpose).all()
            rpose[2] = oc.normalize_angle(rpose[2])
            oc.update_holemap(ref_h, size, dev.hole_scale, xy, rpose); oc.update_obstaclemap(ref_o, osize, dev.obst_scale, xy, rpose)
            assert (dev.holemap_download() == ref_h).all(), it
            assert (dev.obstaclemap_download() == ref_o).all(), it
        dev.close()
    finally:
        oc.set_trig_mode(oc.TRIG_DET)


def test_blocking_wait_times_out_and_poisons_the_context(cs_mod, sim):
    """A blocking call whose completion word does not arrive within the context's bound (slamhip_ctx_set_wait_timeout) returns
    SLAMHIP_ERR_TIMEOUT and poisons the context: every later blocking call on it fails with the same code at once, nothing is
    re-executed, and the handles are still destroyed cleanly.  (The 'kernel that never ends' is a queue of one-million-candidate
    searches in front of a blocking one, against a bound of 1 ms: the enqueue-only searches hold the host back to seven launches
    ahead of the device -- the plan slots' backpressure, slamhip_cs_plan_stats -- so the blocking call finds about six searches
    of ~0.4 ms each in front of its own.)"""
    import slam.net_amd.capi as capi
    ctx2 = cs_mod.Context(0)
    dev = cs_mod.CoreSlamDevice(ctx2, 40.0, 1024, 256)
    try:
        segs = sim.default_field()
        rng = sim.PCG32(5)
        p = sim.trajectory(2)[-1]
        _, xy = sim.make_scan(segs, p, 1080, rng)
        dev.set_scan(xy); dev.update_holemap(p)
        dev.generate_offsets((1 << 20) - 1, 0.1, 0.17, seed=3, stream=1)
        pose, dist, idx = dev.search(p)                              # (a sound call first: the bound is generous by default)
        assert not ctx2.poisoned
        ctx2.set_wait_timeout(1)
        for _ in range(40):
            dev.search_shard_enqueue(p, 0, 1 << 20)
        t0 = time.perf_counter()
        with pytest.raises(capi.SlamhipError) as e:
            dev.search(p)
        assert e.value.code == capi.ERR_TIMEOUT and time.perf_counter() - t0 < 1.0
        assert ctx2.poisoned
        for _ in range(3):                                           # poisoned: fails at once, with the same code
            t1 = time.perf_counter()
            with pytest.raises(capi.SlamhipError) as e2:
                dev.search(p)
            assert e2.value.code == capi.ERR_TIMEOUT and time.perf_counter() - t1 < 0.05
    finally:
        dev.close(); ctx2.close()                                    # (destroy waits for the queue to drain: no bound there)


def test_holemap_large_scan_path():
    """The HoleMap update of scans too large for the in-kernel tables (more than 2048 rays: k2_prepare + the pixel kernel reading
    its tables from memory, the ObstacleMap update in launches of its own), forced on the ordinary test scans with
    SLAMHIP_K2_TWO_LAUNCHES=1: same maps, same fused results.  (Natively the path runs in the 3000- and 4097-ray cases below and
    in the soak.)"""
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    env = dict(os.environ, SLAMHIP_K2_TWO_LAUNCHES="1")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(here, "test_gpu_coreslam.py"), "-m", "gpu", "-x", "-q", "-k",
                        "holemap_golden or holemap_vs_oracle or holemap_unordered or holemap_degenerate or search_and_update_fused or fused_scans or partial_mirror"],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    assert r.returncode == 0, r.stdout.decode(errors="replace")[-3000:]


def test_holemap_more_rays_than_lds_tables(cs_mod, ctx, det, sim):
    """3000 rays (> K2_LDS_RAYS): the update takes the two-launch path by itself, the ObstacleMap update rides nowhere."""
    oc = det
    size, osize, R = 1024, 256, 3000
    segs = sim.default_field()
    dev = cs_mod.CoreSlamDevice(ctx, 40.0, size, osize)
    ref_h = np.full(size * size, 32750, np.uint16)
    ref_o = np.full((osize, osize), -5, np.int8)
    rng = sim.PCG32(61)
    for p in sim.trajectory(4, step=(0.3, 0.1, 0.2)):
        _, xy = sim.make_scan(segs, p, R, rng)
        dev.set_scan(xy)
        dev.update_holemap(p); dev.update_obstaclemap(p)
        oc.update_holemap(ref_h, size, dev.hole_scale, xy, p); oc.update_obstaclemap(ref_o, osize, dev.obst_scale, xy, p)
    assert (dev.holemap_download() == ref_h).all() and (dev.obstaclemap_download() == ref_o).all()
    dev.set_offsets(sim.gaussian_offsets(999))
    base = sim.trajectory(5, step=(0.3, 0.1, 0.2))[-1]
    for _ in range(2):                                                  # fused: search + both updates, twice in a row
        pose, dist, idx = dev.search_and_update(base, 0.6, 50, 10)
        rbi, rpose, rbd, _ = oc.search(ref_h, size, dev.hole_scale, xy, base, sim.gaussian_offsets(999))
        rpose[2] = oc.normalize_angle(rpose[2])
        assert idx == rbi and dist == rbd and (pose == rpose).all()
        oc.update_holemap(ref_h, size, dev.hole_scale, xy, rpose); oc.update_obstaclemap(ref_o, osize, dev.obst_scale, xy, rpose)
    assert (dev.holemap_download() == ref_h).all() and (dev.obstaclemap_download() == ref_o).all()
    dev.close()


# ---- round 3: the BASELINE configurations at their own sizes, the host mirror, threads, two GPUs -------------------------
def _mapped_dev(cs_mod, ctx, sim, size, R, updates, osize=None):
    segs = sim.default_field()
    dev = cs_mod.CoreSlamDevice(ctx, 40.0, size, osize or max(size // 4, 1))
    rng = sim.PCG32(1234)
    traj = sim.trajectory(updates + 1)
    for p in traj[:-1]:
        _, xy = sim.make_scan(segs, p, R, rng)
        dev.set_scan(xy)
        dev.update_holemap(p, 0.6, 50)
    _, xy = sim.make_scan(segs, traj[-1], R, rng)
    base = (traj[-1] + np.array([0.03, -0.02, math.radians(1.0)], np.float32)).astype(np.float32)
    return dev, segs, rng, xy, base


def test_c5_4096_map_262144_candidates_eight_shards(cs_mod, ctx, det, sim):
    """BASELINE config C5 at its own size: 4096^2 HoleMap, 1080 rays, 262 144 candidates in eight blocks of 32 768 (one per
    GPU of the 8-GPU configuration; this GPU plays every rank in turn).  Every distance of rank 0's block, every block's key and
    the min over the eight keys -- what the RCCL min all-reduce delivers -- equal the oracle's single search of the whole list."""
    oc = det
    size, R, K, n = 4096, 1080, 262144, 8
    dev, segs, rng, xy, base = _mapped_dev(cs_mod, ctx, sim, size, R, 10)
    pix = dev.holemap_download()
    offs = sim.gaussian_offsets(K - 1, 0.1, math.radians(10.0), seed=42)
    dev.set_scan(xy)
    dev.set_offsets(offs)
    rbi, rpose, rbd, rall = oc.search(pix, size, dev.hole_scale, xy, base, offs)
    keys = []
    for r in range(n):
        first, count = K * r // n, K * (r + 1) // n - K * r // n
        assert count == 32768
        k = dev.search_shard(base, first, count)
        blk = rall[first:first + count]
        j = int(np.argmin(blk))                                    # (first strictly smaller wins: argmin returns the first minimum)
        assert (k >> 32, k & 0xFFFFFFFF) == (int(blk[j]), first + j), r
        keys.append(k)
    pose, dist, idx = dev.pose_from_key(base, min(keys))
    assert idx == rbi and dist == rbd and (pose == rpose).all()
    # every distance of one GPU's share (rank 0's block: the base pose and the first 32 767 jitters)
    poses = np.vstack([base[None], base[None] + offs[:32767]]).astype(np.float32)
    d, bi, bd = dev.distance_poses(poses)
    assert (d == rall[:32768]).all()
    assert dev.selfcheck_failures == 0
    dev.close()


def test_c3_fused_at_bench_size(cs_mod, ctx, det, sim):
    """BASELINE config C3 as bench.py times it: the fused search + HoleMap / ObstacleMap update at 2048^2 with 1080 rays and
    16 384 candidates, four consecutive scans (each search reads the map the previous call's update wrote): winner, pose and
    both maps equal the oracle's after every scan."""
    oc = det
    size, osize, R, K = 2048, 512, 1080, 16384
    dev, segs, rng, xy, base = _mapped_dev(cs_mod, ctx, sim, size, R, 12, osize)
    ref_h = dev.holemap_download()
    ref_o = dev.obstaclemap_download()
    offs = sim.gaussian_offsets(K - 1, 0.1, math.radians(10.0), seed=42)
    dev.set_offsets(offs)
    traj = sim.trajectory(18)
    est = base.copy()
    for i in range(4):
        if i > 0:
            _, xy = sim.make_scan(segs, traj[12 + i], R, rng)
        dev.set_scan(xy)
        search = (est + np.array([0.01 * i, -0.005 * i, 0.002 * i], np.float32)).astype(np.float32)
        pose, dist, idx = dev.search_and_update(search, 0.6, 50, 10)
        rbi, rpose, rbd, _ = oc.search(ref_h, size, dev.hole_scale, xy, search, offs)
        rpose[2] = oc.normalize_angle(rpose[2])
        assert idx == rbi and dist == rbd and (pose == rpose).all(), i
        n_px = oc.update_holemap(ref_h, size, dev.hole_scale, xy, rpose, 0.6, 50)
        oc.update_obstaclemap(ref_o, osize, dev.obst_scale, xy, rpose, 10)
        assert dev.last_holemap_pixels == n_px
        assert (dev.holemap_download() == ref_h).all(), i
        assert (dev.obstaclemap_download() == ref_o).all(), i
        est = np.asarray(pose, np.float32)
    dev.close()


def test_holemap_partial_mirror(cs_mod, ctx, sim):
    """slamhip_cs_holemap_mirror (live HoleMap.Pixels at the price of what changed): after every update the mirror -- brought up
    to date by copying the scan's bounding rectangle only -- equals a full download; the rectangle is smaller than the map; a
    second call without an update copies nothing; reset and upload make everything dirty again."""
    size, R = 1024, 720
    segs = sim.default_field()
    dev = make_dev(cs_mod, ctx, size, 64)
    rng = sim.PCG32(8)
    mirror = np.zeros(size * size, np.uint16)
    assert dev.holemap_mirror(mirror) == (0, 0, size - 1, size - 1)           # first call: everything
    assert (mirror == 32750).all()
    assert dev.holemap_mirror(mirror) == (0, 0, -1, -1)
    small = 0
    for k, p in enumerate(sim.trajectory(10, step=(0.35, 0.2, 0.05))):
        pose = (p + np.array([-6.0 + k, 4.0 - k, 0.0], np.float32)).astype(np.float32)     # wander: the rectangles move
        _, xy = sim.make_scan(segs, pose, R, rng)
        dev.set_scan(xy * (0.35 if k % 3 == 0 else 1.0))                      # (some short scans: small rectangles)
        dev.update_holemap(pose, 0.6, 50)
        if k % 2:                                                              # sometimes two updates between mirror calls: the union
            dev.update_holemap((pose + np.array([1.5, -1.0, 0.3], np.float32)).astype(np.float32), 0.6, 50)
        x0, y0, x1, y1 = dev.holemap_mirror(mirror)
        assert 0 <= x0 <= x1 < size and 0 <= y0 <= y1 < size
        small += (x1 - x0 + 1) * (y1 - y0 + 1) < size * size
        assert (mirror == dev.holemap_download()).all(), k
    assert small > 0
    # the fused call's update marks its rectangle too
    dev.set_offsets(sim.gaussian_offsets(255))
    dev.search_and_update(np.array([20.0, 20.0, 0.0], np.float32), 0.6, 50, 10)
    dev.holemap_mirror(mirror)
    assert (mirror == dev.holemap_download()).all()
    dev.reset()
    assert dev.holemap_mirror(mirror) == (0, 0, size - 1, size - 1) and (mirror == 32750).all()
    up = np.arange(size * size, dtype=np.uint32).astype(np.uint16)
    dev.holemap_upload(up)
    assert dev.holemap_mirror(mirror) == (0, 0, size - 1, size - 1) and (mirror == up).all()
    dev.close()


@pytest.mark.parametrize("size,R", [(1024, 720), (2048, 1080), (300, 360), (512, 2600)])   # (300: rows that are not whole 16-byte units; 2600 rays: the large-scan tables)
def test_holemap_async_mirror(cs_mod, ctx, sim, size, R):
    """slamhip_cs_holemap_mirror_async / _wait (live HoleMap.Pixels without stalling the scan): twelve fused
    scans back to back, a mirror request after each -- after _wait the mirror equals a full download taken at the same point,
    every scan; what is pushed is a fraction of the map; several updates between two requests come together; a request without
    an update pushes nothing; reset, upload and another array make everything news again."""
    segs = sim.default_field()
    dev = make_dev(cs_mod, ctx, size, 64)
    rng = sim.PCG32(8)
    if size == 2048:
        # an array that OWNS its pages (starts on a page boundary, whole pages long): the device writes it directly; the other
        # sizes run the staged form (an ordinary NumPy array shares its first and last page with the heap: slamhip.h)
        raw = np.zeros(size * size + 4096, np.uint16)
        ofs = (-raw.ctypes.data % 4096) // 2
        mirror = raw[ofs:ofs + size * size]
        assert mirror.ctypes.data % 4096 == 0
    else:
        mirror = np.zeros(size * size, np.uint16)
    dev.holemap_mirror_async(mirror)
    rect, px = dev.holemap_mirror_wait()
    assert rect == (0, 0, size - 1, size - 1) and px >= size * size and (mirror == 32750).all()     # the first request: everything (in 8-pixel units)
    dev.holemap_mirror_async(mirror)
    assert dev.holemap_mirror_wait() == ((0, 0, -1, -1), 0)
    dev.set_offsets(sim.gaussian_offsets(1500, 0.05, 0.05, seed=3))
    fractions = []
    for k, p in enumerate(sim.trajectory(12, step=(0.35, 0.2, 0.05))):
        pose = (p + np.array([-6.0 + k, 4.0 - k, 0.0], np.float32)).astype(np.float32)     # wander, also towards the map's edge
        _, xy = sim.make_scan(segs, pose, R, rng)
        dev.set_scan(xy * (0.35 if k % 3 == 0 else 1.0))
        if k % 4 == 3:
            dev.update_holemap(pose, 2.0, 50)                                              # (a plain update with wide holes in between)
        dev.search_and_update(pose, 0.6, 50, 10)                                           # returns with the pose: the updates run on
        dev.holemap_mirror_async(mirror)
        want = dev.holemap_download()                                                      # (ordered behind the updates, like the snapshot)
        rect, px = dev.holemap_mirror_wait()
        assert (mirror == want).all(), (k, int((mirror != want).sum()))
        assert 0 < px <= size * (size + 7) and 0 <= rect[0] <= rect[2] < size and 0 <= rect[1] <= rect[3] < size
        fractions.append(px / float(size * size))
    assert min(fractions) < 0.6, fractions
    # two handles on one context do not disturb each other's mirrors; another array starts from everything
    other = np.zeros(size * size, np.uint16)
    dev.holemap_mirror_async(other)
    rect, px = dev.holemap_mirror_wait()
    assert px >= size * size and (other == dev.holemap_download()).all()
    dev.reset()
    dev.holemap_mirror_async(other)
    dev.holemap_mirror_wait()
    assert (other == 32750).all()                          # (only the units that differ from the last snapshot travel)
    up = np.arange(size * size, dtype=np.uint32).astype(np.uint16)
    dev.holemap_upload(up)
    dev.holemap_mirror_async(other)
    dev.holemap_mirror_wait()
    assert (other == up).all()
    # the blocking rectangle mirror and the asynchronous one on the same array, in turns
    for k in range(4):
        p = np.array([18.0 + k, 21.0, 0.3 * k], np.float32)
        _, xy = sim.make_scan(segs, p, min(R, 1080), rng)
        dev.set_scan(xy); dev.update_holemap(p, 0.6, 50)
        if k % 2:
            dev.holemap_mirror(other)
        else:
            dev.holemap_mirror_async(other); dev.holemap_mirror_wait()
        assert (other == dev.holemap_download()).all(), k
    dev.holemap_mirror_release()
    dev.close()


def test_two_handles_one_context_two_threads(cs_mod, ctx, det, sim):
    """Handles that share a context may be driven from different threads (slamhip.h): their blocking calls deliver results
    through the context's one mailbox, under its lock.  Two operator objects with different maps / scans / lists are hammered
    from two threads; every call must return ITS answer."""
    import threading
    oc = det
    segs = sim.default_field()
    devs, wants, args = [], [], []
    for t, (size, R, K) in enumerate(((256, 200, 700), (512, 360, 1500))):
        dev = make_dev(cs_mod, ctx, size, 64)
        rng = sim.PCG32(40 + t)
        for p in sim.trajectory(4):
            _, xy = sim.make_scan(segs, p, R, rng)
            dev.set_scan(xy)
            dev.update_holemap(p)
        pix = dev.holemap_download()
        base = (sim.trajectory(5)[-1] + np.array([0.02 * t, -0.01, 0.01], np.float32)).astype(np.float32)
        _, xy = sim.make_scan(segs, base, R, rng)
        offs = sim.gaussian_offsets(K - 1, seed=3 + t)
        dev.set_scan(xy)
        dev.set_offsets(offs)
        rbi, rpose, rbd, _ = oc.search(pix, size, dev.hole_scale, xy, base, offs)
        devs.append(dev); wants.append((rbi, rbd, rpose)); args.append(base)
    errors = []

    def worker(i):
        try:
            for _ in range(300):
                pose, dist, idx = devs[i].search(args[i])
                if (idx, dist) != wants[i][:2] or not (pose == wants[i][2]).all():
                    errors.append((i, idx, dist))
                    return
        except Exception as e:                                     # noqa: BLE001
            errors.append((i, repr(e)))

    th = [threading.Thread(target=worker, args=(i,)) for i in range(2)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errors, errors[:3]
    for d in devs:
        d.close()


def _device_count():
    import torch
    return torch.cuda.device_count()


def test_group_two_gpus(det, sim):
    """slamhip_group_* on TWO devices (skipped on a one-GPU box): the candidates are block-sharded over the GPUs, the packed keys
    go through one grouped ncclAllReduce(min) over xGMI, the map updates are replicated -- winner, pose and both replicas' maps
    equal the oracle's."""
    if _device_count() < 2:
        pytest.skip("needs two GPUs")
    import ctypes as C
    import slam.net_amd.capi as capi
    import slam.net_amd.coreslam as cs_mod
    oc = det
    size, R, K = 1024, 1080, 16384
    segs = sim.default_field()
    rng = sim.PCG32(77)
    g = C.c_void_p()
    dev_ids = (C.c_int32 * 2)(0, 1)
    capi.call("slamhip_group_create", dev_ids, 2, C.c_float(40.0), size, 64, C.byref(g))
    try:
        capi.call("slamhip_group_reset", g, -5)
        ref = np.full(size * size, 32750, np.uint16)
        scale = size / 40.0
        for p in sim.trajectory(6)[:-1]:
            _, xy = sim.make_scan(segs, p, R, rng)
            capi.call("slamhip_group_set_scan", g, capi.fptr(capi.f32(xy)), xy.shape[0])
            capi.call("slamhip_group_update_maps", g, capi.fptr(capi.f32(p)), C.c_float(0.6), 50, 60)
            oc.update_holemap(ref, size, scale, xy, p, 0.6, 50)
        pose = sim.trajectory(6)[-1]
        _, xy = sim.make_scan(segs, pose, R, rng)
        base = (pose + np.array([0.02, -0.03, 0.01], np.float32)).astype(np.float32)
        offs = sim.gaussian_offsets(K - 1)
        capi.call("slamhip_group_set_scan", g, capi.fptr(capi.f32(xy)), xy.shape[0])
        capi.call("slamhip_group_set_offsets", g, capi.fptr(capi.f32(offs)), offs.shape[0])
        rbi, rpose, rbd, _ = oc.search(ref, size, scale, xy, base, offs)
        for _ in range(3):
            out_pose = np.zeros(3, np.float32)
            dist, idx = C.c_int32(), C.c_int32()
            capi.call("slamhip_group_search", g, capi.fptr(base), capi.fptr(out_pose), C.byref(dist), C.byref(idx))
            assert idx.value == rbi and dist.value == rbd and (out_pose == rpose).all()
        eq = C.c_int32(-1)
        capi.call("slamhip_group_replicas_equal", g, C.byref(eq))  # the library's own replica check
        assert eq.value == 1
        for r in range(2):                                         # the replicas hold the same, oracle-equal map
            h = C.c_void_p()
            capi.call("slamhip_group_cs", g, r, C.byref(h))
            pix = np.empty(size * size, np.uint16)
            capi.call("slamhip_cs_holemap_download", h, pix.ctypes.data_as(C.POINTER(C.c_uint16)), pix.size)
            assert (pix == ref).all(), r
        # the whole scan in one call on both GPUs: same winner, pose normalised, both replicas updated from it
        fp = np.zeros(3, np.float32)
        dist, idx = C.c_int32(), C.c_int32()
        capi.call("slamhip_group_search_and_update", g, capi.fptr(base), C.c_float(0.6), 50, 60, capi.fptr(fp), C.byref(dist), C.byref(idx))
        wp = np.array([rpose[0], rpose[1], oc.normalize_angle(float(rpose[2]))], np.float32)
        assert idx.value == rbi and dist.value == rbd and (fp == wp).all()
        oc.update_holemap(ref, size, scale, xy, wp, 0.6, 50)
        capi.call("slamhip_group_replicas_equal", g, C.byref(eq))
        assert eq.value == 1
        for r in range(2):
            h = C.c_void_p()
            capi.call("slamhip_group_cs", g, r, C.byref(h))
            pix = np.empty(size * size, np.uint16)
            capi.call("slamhip_cs_holemap_download", h, pix.ctypes.data_as(C.POINTER(C.c_uint16)), pix.size)
            assert (pix == ref).all(), r
    finally:
        capi.call("slamhip_group_destroy", g)


def test_bench_two_gpus_rccl():
    """bench.py --gpus 2 under torch.distributed.run with RCCL (skipped on a one-GPU box): the library's communicator must be the
    one that ran (`collective` names libslamhip, two ranks seen by RCCL), and the winner over 2 x 4096 candidates equals that of
    one rank searching all 8192."""
    if _device_count() < 2:
        pytest.skip("needs two GPUs")
    import json
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    common = ["--steps", "20", "--warmup", "3", "--size", "1024", "--map-updates", "8", "--no-cpu-baseline"]
    r2 = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                         "--master-port", str(port), os.path.join(root, "bench.py"), "--gpus", "2", "--cands", "4096"] + common,
                        env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    assert r2.returncode == 0, r2.stderr.decode(errors="replace")[-3000:]
    j2 = json.loads([l for l in r2.stdout.decode().splitlines() if l.startswith("{")][0])
    r1 = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--cands", "8192", "--no-extras"] + common,
                        stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert r1.returncode == 0, r1.stderr.decode(errors="replace")[-3000:]
    j1 = json.loads([l for l in r1.stdout.decode().splitlines() if l.startswith("{")][0])
    assert j2["n_gpus"] == 2 and "libslamhip" in j2["config"]["collective"], j2["config"]["collective"]
    assert j2["config"]["collective_ranks"] == 2 and j2["multi_gpu"]["collective_ranks"] == 2
    assert (j2["config"]["best_index"], j2["config"]["best_distance"]) == (j1["config"]["best_index"], j1["config"]["best_distance"])
    assert j2["multi_gpu"]["allreduce_us"] > 0 and j2["multi_gpu"]["overlapped_evals_per_s"] > 0 and j2["value"] > 0
    assert j2["multi_gpu"]["replicas_equal"] is True


def test_lib_comm_blocking_step_single_rank(cs_mod, ctx, sim):
    """slamhip_cs_search_allreduce (the per-scan form: K1, the collective and the key's hand-over on one stream) on a one-rank
    communicator, mixed with the asynchronous batched form and a changed batch size: always the key of the blocking search."""
    import slam.net_amd.distributed as D
    size, R, K = 512, 360, 5000
    dev = make_dev(cs_mod, ctx, size, 64)
    segs = sim.default_field()
    rng = sim.PCG32(5)
    for p in sim.trajectory(6):
        _, xy = sim.make_scan(segs, p, R, rng)
        dev.set_scan(xy)
        dev.update_holemap(p)
    base = sim.trajectory(7)[-1]
    dev.set_offsets(sim.gaussian_offsets(K - 1))
    comm = D.LibComm(ctx, 0, 1)
    try:
        assert comm.info() == (0, 1)
        for first, count in ((0, K), (K // 3, K - K // 3), (K - 1, 1)):
            want = dev.search_shard(base, first, count)
            assert comm.search_allreduce(dev, base, first, count) == want
            step = comm.bind_step(dev, base, first, count)
            for batch in (1, 16, 5):
                comm.set_batch(batch)
                for _ in range(7):
                    step()
                assert comm.search_allreduce(dev, base, first, count) == want     # (waits for the asynchronous steps first)
                for _ in range(3):
                    step()
                assert comm.wait() == want
            bstep = comm.bind_search_allreduce(dev, base, first, count)
            assert [bstep() for _ in range(5)] == [want] * 5
        assert comm.search_allreduce(dev, base, 0, 0) == 2 ** 64 - 1           # a rank without candidates: the neutral key
        assert comm.allreduce_probe(20) > 0
    finally:
        comm.close()
        dev.close()


def test_lib_comm_fused_scan_single_rank(cs_mod, ctx, det, sim):
    """slamhip_cs_search_allreduce_and_update on a one-rank communicator: search over the rank's block, the collective, the winner
    decoded ON THE DEVICE from the reduced key, both map updates queued behind it -- winner, pose and both maps equal the oracle's
    scan by scan (as slamhip_cs_search_and_update does without a communicator), also with a sub-block of the list (the winner of
    the block) and for a rank without candidates (the neutral key: search pose, maps updated from it)."""
    import slam.net_amd.distributed as D
    oc = det
    size, osize, R, K = 512, 128, 540, 3000
    dev = make_dev(cs_mod, ctx, size, osize)
    segs = sim.default_field()
    rng = sim.PCG32(21)
    ref_h = np.full(size * size, 32750, np.uint16)
    ref_o = np.full(osize * osize, -5, np.int8)
    traj = sim.trajectory(12)
    for p in traj[:4]:
        _, xy = sim.make_scan(segs, p, R, rng)
        dev.set_scan(xy); dev.update_holemap(p); dev.update_obstaclemap(p)
        oc.update_holemap(ref_h, size, dev.hole_scale, xy, p)
        oc.update_obstaclemap(ref_o, osize, dev.obst_scale, xy, p)
    offs = sim.gaussian_offsets(K - 1, 0.05, math.radians(4.0), seed=9)
    dev.set_offsets(offs)
    comm = D.LibComm(ctx, 0, 1)
    try:
        for i, p in enumerate(traj[4:]):
            _, xy = sim.make_scan(segs, p, R, rng)
            base = (p + np.array([0.02, -0.01, 0.005], np.float32)).astype(np.float32)
            dev.set_scan(xy)
            first, count = ((0, K), (K // 4, K // 2), (0, K), (5, 0))[i % 4]
            pose, dist, idx = comm.search_allreduce_and_update(dev, base, first, count, 0.6, 50, 10)
            if count > 0:
                sub = offs[first - 1:first - 1 + count] if first > 0 else offs[:count - 1]
                rbi, rpose, rbd, _ = oc.search(ref_h, size, dev.hole_scale, xy, base, sub) if first == 0 else (None, None, None, None)
                if first > 0:                                      # a block that does not hold candidate 0: every candidate is base + offs
                    d_all = [oc.distance(ref_h, size, dev.hole_scale, xy, (base + o).astype(np.float32)) for o in sub]
                    j = int(np.argmin(np.asarray(d_all, np.int64)))
                    rbi, rbd, rpose = first + j, int(d_all[j]), (base + sub[j]).astype(np.float32)
                assert (idx, dist) == (int(rbi), int(rbd)), (i, idx, dist, rbi, rbd)
            else:
                assert (dist << 32 | (idx & 0xFFFFFFFF)) & 0xFFFFFFFFFFFFFFFF == 2 ** 64 - 1
                rpose = base.copy()
            rp = np.asarray(rpose, np.float32).copy()
            rp[2] = oc.normalize_angle(rp[2])
            assert (pose == rp).all(), (i, pose, rp)
            oc.update_holemap(ref_h, size, dev.hole_scale, xy, rp)
            oc.update_obstaclemap(ref_o, osize, dev.obst_scale, xy, rp)
            assert (dev.holemap_download() == ref_h).all(), i
            assert (dev.obstaclemap_download().reshape(-1) == ref_o).all(), i
    finally:
        comm.close()
        dev.close()


def test_group_worker_threads():
    """The group's per-GPU worker threads (slamhip_group_search / _update_maps / _set_scan run every rank's part on its own
    thread): forced on for the one-GPU group of test_group_single_gpu with SLAMHIP_GROUP_THREADS=1."""
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    env = dict(os.environ); env["SLAMHIP_GROUP_THREADS"] = "1"
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(here, "test_gpu_coreslam.py"), "-m", "gpu", "-x", "-q", "-k",
                        "test_group_single_gpu or test_maps_checksum_and_replica_checks"], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
    assert r.returncode == 0, r.stdout.decode(errors="replace")[-3000:]


def test_bench_library_collective_one_rank():
    """bench.py with SLAMHIP_BENCH_COLLECTIVE=lib1: the N > 1 section of the benchmark -- the library's own RCCL communicator, the
    blocking per-scan step, the overlapped form, the collective's probe, the replica check and the fused per-scan form
    (slamhip_cs_search_allreduce_and_update) with its self-checks -- on ONE rank, which is all a one-GPU box can run; the driver's
    multi-GPU run goes through the same code with N ranks."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, SLAMHIP_BENCH_COLLECTIVE="lib1")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "20", "--warmup", "3", "--size", "1024", "--map-updates", "8",
                        "--cands", "4096", "--no-cpu-baseline", "--no-extras"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert r.returncode == 0, r.stderr.decode(errors="replace")[-3000:]
    j = json.loads([l for l in r.stdout.decode().splitlines() if l.startswith("{")][0])
    assert "libslamhip" in j["config"]["collective"] and j["config"]["collective_ranks"] == 1
    m = j["multi_gpu"]
    assert m["replicas_equal"] is True and m["allreduce_us"] > 0
    assert m["us_per_step"] > 0 and m["per_scan_blocking_us_per_step"] > m["us_per_step"] * 0.5 and 0 < m["efficiency_same_form"] < 2.0
    assert "its key equals the timed region's: True" in j["config"]["per_scan_blocking_is"]
    f = m["fused_scan_allreduce_and_update"]
    assert f["first_scan_key_equals_search_key"] is True and f["replicas_equal_after"] is True and f["us_per_scan"] > 0, f
