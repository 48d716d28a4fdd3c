"""GPU parity tests for the CoreSLAM hot path: HIP kernels (through the C-ABI) vs the CPU oracle and the
committed golden fixtures.  Integer outputs (distances, arg-min, HoleMap / ObstacleMap cells) must be
bit-exact; poses must be identical floats (same IEEE adds on both sides)."""
import glob
import math
import os

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


def test_k1_tile_boxes_selfcheck():
    """Re-run the distance tests with SLAMHIP_K1_VERIFY=1 (every end point is checked against its LDS tile
    box, every staged pixel against the map), with SLAMHIP_K1_GLOBAL=1 (global-gather fallback kernels) and
    with tile budgets / layouts that force every step kind: all must stay bit-exact."""
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    sel = "test_distance_golden or test_distance_quirks or test_distance_64bit or test_search_full_size"
    for env_extra in ({"SLAMHIP_K1_VERIFY": "1", "SLAMHIP_K1_CPL": "1"}, {"SLAMHIP_K1_VERIFY": "1", "SLAMHIP_K1_CPL": "2"},
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
    """The fused call and the processor with the result block after the updates (SLAMHIP_FUSED_WAIT_UPDATES=1) and without
    the host mailbox (SLAMHIP_NO_HOSTWAIT=1: copy + synchronise): same results as the default (pose from K1's final arriver)."""
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    sel = "test_search_and_update_fused or test_processor_vs_oracle or test_fused_scans_back_to_back"
    for env_extra in ({"SLAMHIP_FUSED_WAIT_UPDATES": "1"}, {"SLAMHIP_NO_HOSTWAIT": "1"}, {"SLAMHIP_K1_GLOBAL": "1"}):
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
    assert j1["roofline"]["launches"] == 20


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
