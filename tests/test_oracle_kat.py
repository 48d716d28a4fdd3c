"""Known-answer tests that pin the CPU oracle (C restatement + independent NumPy restatement).

The reference ships no tests or golden vectors, so these are the hand-derived
answers of SURVEY.md sec.4 and the Appendix-A quirk list, each citing the reference lines.
CPU only (-m "not gpu").
"""
import math

import numpy as np
import pytest

INT_MAX = 2 ** 31 - 1


def uniform_map(size, v=32750):
    return np.full(size * size, v, np.uint16)


# ---- CoreSLAMProcessor.cs:169 / :431 --------------------------------------------------------
def test_reset_value_and_blend_kats(oc, npo):
    assert (0 + 65500) // 2 == 32750                                   # :169
    assert npo.blend(32750, 65500, 50) == 39146                        # :431
    assert npo.blend(39146, 65500, 50) == 44293
    assert npo.blend(44293, 65500, 50) == 48434
    assert npo.blend(32750, 0, 50) == 26353
    # order dependence
    assert npo.blend(npo.blend(32750, 65500, 50), 0, 50) == 31500
    assert npo.blend(npo.blend(32750, 0, 50), 65500, 50) == 33998
    # same through the C oracle: a 1-pixel-long "ray" cannot be drawn (derrorv==0), so draw a 3 px ray
    pix = uniform_map(16)
    n = oc.draw_ray_holemap(pix, 16, 2, 2, 12, 2, 8, 2, 0, 50)
    assert n == 11
    # far from the hole (x <= dx - 2*derrorv = 10 - 8 = 2): pixval stays 65500
    assert pix[2 * 16 + 2] == 39146 and pix[2 * 16 + 3] == 39146 and pix[2 * 16 + 4] == 39146


# ---- CoreSLAMProcessor.cs:226-259 -----------------------------------------------------------
@pytest.mark.parametrize("R", [8, 360, 1080])
def test_distance_uniform_map(oc, npo, R):
    size = 128
    scale = oc.map_scale(size, 40.0)
    pix = uniform_map(size)
    ang = np.arange(R) * (2 * math.pi / R)
    xy = np.stack([3.0 * np.cos(ang), 3.0 * np.sin(ang)], 1).astype(np.float32)
    assert oc.distance(pix, size, scale, xy, [20, 20, 0.1]) == 32750 * 1024 == 33536000
    # quirk 1 (:253): divides by ALL points -> exactly half in-bounds halves the distance
    xy_half = xy.copy()
    xy_half[: R // 2] += 1000.0
    assert oc.distance(pix, size, scale, xy_half, [20, 20, 0.1]) == 16768000
    # quirk 3 (:257): nothing in bounds -> int.MaxValue
    assert oc.distance(pix, size, scale, xy + 1000.0, [20, 20, 0.1]) == INT_MAX
    d = npo.distance_batch_pxcs(pix, size, xy_half, oc.pose_to_pxcs([20, 20, 0.1], scale))
    assert int(d[0]) == 16768000


def test_distance_truncation_quirk(oc):
    """Quirk 2 (:232-233,:240-244): +0.5 then truncate toward zero; (-1,0) lands on column 0."""
    size = 8
    pix = np.zeros(size * size, np.uint16)
    pix[0] = 1000                      # pixel (0,0)
    pix[size * 3 + 0] = 77             # pixel (x=0,y=3)
    # px = -0.75 -> (int)(-0.75) = 0 -> accepted
    pxcs = np.array([-0.75, 3.2, 1.0, 0.0], np.float32)      # c=1,s=0: x = px + X, y = py + Y
    xy = np.array([[0.0, 0.0]], np.float32)
    d, bi, bd = oc.distance_batch_pxcs(pix, size, xy, pxcs[None])
    assert d[0] == 77 * 1024
    pxcs2 = np.array([-1.0, 3.2, 1.0, 0.0], np.float32)      # exactly -1 -> x=-1 -> rejected
    d2, _, _ = oc.distance_batch_pxcs(pix, size, xy, pxcs2[None])
    assert d2[0] == INT_MAX
    # NaN coordinates behave like cvttss2si (INT_MIN -> rejected), not like "0"
    pxcs3 = np.array([np.nan, 3.2, 1.0, 0.0], np.float32)
    d3, _, _ = oc.distance_batch_pxcs(pix, size, xy, pxcs3[None])
    assert d3[0] == INT_MAX


def test_distance_64bit_intermediate(oc):
    """H3: sum*1024 exceeds 2^32 at R=1080 with bright pixels (:229,:253)."""
    size = 64
    pix = uniform_map(size, 65535)
    xy = np.zeros((1080, 2), np.float32)
    assert oc.distance(pix, size, oc.map_scale(size, 40.0), xy, [20, 20, 0]) == 65535 * 1024


def test_argmin_tiebreak(oc, npo):
    """Quirks 4,5 (:644,:700,:626-628): strict '<' keeps the earliest; base pose is the incumbent."""
    size = 32
    scale = oc.map_scale(size, 40.0)
    pix = uniform_map(size)
    xy = np.array([[1.0, 0.0], [0.0, 1.0]], np.float32)
    offs = np.zeros((5, 3), np.float32)
    offs[:, 0] = [0.1, 0.2, 0.3, 0.4, 0.5]
    bi, pose, bd, alld = oc.search(pix, size, scale, xy, [20, 20, 0], offs)
    assert (alld == 33536000).all() and bi == 0 and bd == 33536000
    assert tuple(pose) == (20.0, 20.0, 0.0)
    pix2 = pix.copy().reshape(size, size)
    pix2[:, :] = 40000
    pix2[16, 16:20] = 100      # make some jittered candidates strictly better
    bi2, _, _, all2 = oc.search(pix2.reshape(-1), size, scale, xy, [20, 20, 0], offs)
    assert bi2 == int(np.argmin(all2))
    b3, _, _, all3 = npo.search(pix2.reshape(-1), size, scale, xy, [20, 20, 0], offs, trig="libm")
    assert (all2 == all3).all() and b3 == bi2


# ---- NormalizeAngle MathEx.cs:116-138 -------------------------------------------------------
def test_normalize_angle(oc, npo):
    pi = np.float32(math.pi)
    for a in [0.0, 1.0, -1.0, 3.2, -3.2, 7.0, -7.0, 100.0, float(pi), float(-pi), 6.2831855]:
        r = oc.normalize_angle(a)
        assert -float(pi) - 1e-6 <= r <= float(pi)
        assert np.float32(r) == npo.normalize_angle(a)
        assert abs(math.remainder(r - a, 2 * math.pi)) < 1e-5


# ---- ClipRay :320-345 -----------------------------------------------------------------------
def test_clip_ray(oc, npo):
    assert oc.clip_ray(100, 50, 60, 10, 10) == (True, 50, 60)            # inside: untouched
    ok, x, y = oc.clip_ray(100, -50, 60, 10, 10)                          # :329 (60-10)*50/(-60) = -41
    assert (ok, x, y) == (True, 0, 60 - 41)
    ok, x, y = oc.clip_ray(100, 150, 60, 10, 10)                          # :340 (50)*(-51)/(140) = -18
    assert (ok, x, y) == (True, 99, 60 - 18)
    assert oc.clip_ray(100, -5, 60, -5, 10)[0] is False                   # :324 degenerate
    rng = np.random.default_rng(1)
    for _ in range(2000):
        a = [int(v) for v in rng.integers(-5000, 5000, 4)]
        assert oc.clip_ray(400, *a) == npo.clip_ray(400, *a)
    # wrapping products (:329 unchecked int): far-outside endpoints
    for a in [(-2000000000, 1500000000, 100, 200), (2000000000, -1500000000, 100, 200)]:
        assert oc.clip_ray(2048, *a) == npo.clip_ray(2048, *a)


# ---- DrawLaserRayOnHoleMap :359-443 ---------------------------------------------------------
def test_holemap_ray_profile(oc, npo):
    """Quirks 8-13: V profile indexed by unclipped dx, no correction on the way down, overshoot."""
    size = 256
    frags = npo.ray_fragments(size, 10, 100, 210, 100, 195, 100)         # dx=200, derrorv=15
    assert len(frags) == 201
    vals = [v for _, v in frags]
    assert all(v == 65500 for v in vals[:171])                           # x <= dx-2d = 170
    assert vals[185] == 65500 - 15 * (65500 // 15)                       # bottom of the V at x = dx-d
    assert vals[185] == 65500 % 15 == 10
    assert vals[200] == 65506                                            # overshoot (65506 for derrorv=15)
    assert min(vals) == vals[185]
    # derrorv == 0 -> skipped (:389-392)
    assert npo.ray_fragments(size, 10, 100, 210, 100, 210, 100) is None
    pix = uniform_map(size)
    assert oc.draw_ray_holemap(pix, size, 10, 100, 210, 100, 210, 100, 0, 50) == -1
    # clipped walk: dxc < dx but profile still uses dx (:368-371,:404-408)
    fr = npo.ray_fragments(size, 200, 100, 400, 100, 385, 100)           # ends outside the map
    assert len(fr) == 56 and all(v == 65500 for _, v in fr)
    # ties go to the y-major branch (:377 strict dx > dy)
    fr = npo.ray_fragments(size, 10, 10, 60, 60, 55, 55)
    assert [p for p, _ in fr[:3]] == [10 * size + 10, 11 * size + 11, 12 * size + 12]


def test_holemap_update_c_vs_numpy(oc, npo, sim):
    segs = sim.default_field()
    for size, R, hw, pose in [(64, 90, 0.6, (20, 20, 0.3)), (128, 360, 2.0, (12.5, 30.2, -2.0)),
                              (200, 180, 0.6, (38.9, 20.0, 1.0)), (96, 120, 5.0, (6.0, 6.0, 0.77))]:
        rays, xy = sim.make_scan(segs, pose, R, sim.PCG32(7))
        scale = oc.map_scale(size, 40.0)
        a = uniform_map(size); b = a.copy()
        for it in range(3):
            p = (pose[0] + 0.1 * it, pose[1] - 0.05 * it, pose[2] + 0.02 * it)
            pxcs = oc.pose_to_pxcs(p, scale)
            n1 = oc.update_holemap_pxcs(a, size, scale, xy, pxcs, hw, 50)
            n2 = npo.update_holemap_pxcs(b, size, scale, xy, pxcs, hw, 50)
            assert n1 == n2 and n1 > 0
            assert (a == b).all()


def test_holemap_hostile_points_c_vs_numpy(oc, npo, sim):
    """The inputs the randomised soak leans on: an end point tens of kilometres out (ClipRay's int32 products wrap,
    CoreSLAMProcessor.cs:329,:340 -- the clipped ray can come back with dyc > dxc), duplicates, a point at the robot,
    NaN, weights 1 and 255.  The two independently written restatements must agree pixel for pixel."""
    segs = sim.default_field()
    for size, q, hw, pose in [(64, 1, 2.0, (25.2, 16.0, -2.55)), (128, 255, 0.1, (20.0, 20.0, 0.4)), (96, 128, 5.0, (30.1, 9.7, 3.3))]:
        rays, xy = sim.make_scan(segs, pose, 60, sim.PCG32(11))
        xy = xy.copy()
        xy[0] = [3.0e4, -2.0e4]; xy[1] = xy[2]; xy[3] = [0.0, 0.0]; xy[4] = [np.nan, 1.0]; xy[5] = [-7.0e5, 9.0e5]
        scale = oc.map_scale(size, 40.0)
        a = uniform_map(size); b = a.copy()
        for it in range(2):
            p = (pose[0] + 0.1 * it, pose[1] - 0.05 * it, pose[2] + 0.02 * it)
            pxcs = oc.pose_to_pxcs(p, scale)
            n1 = oc.update_holemap_pxcs(a, size, scale, xy, pxcs, hw, q)
            n2 = npo.update_holemap_pxcs(b, size, scale, xy, pxcs, hw, q)
            assert n1 == n2 and n1 > 0
            assert (a == b).all()
        oa = np.full(32 * 32, -5, np.int8); ob = np.full((32, 32), -5, np.int8)
        pxo = oc.pose_to_pxcs(pose, oc.map_scale(32, 40.0))
        oc.update_obstaclemap_pxcs(oa, 32, xy, pxo, 10)
        npo.update_obstaclemap_pxcs(ob, 32, xy, pxo, 10)
        assert (oa.reshape(32, 32) == ob).all()


def test_holemap_robot_outside_map(oc):
    """Quirk 11 (:509-512)."""
    pix = uniform_map(64)
    xy = np.array([[1.0, 0.0]], np.float32)
    assert oc.update_holemap(pix, 64, oc.map_scale(64, 40.0), xy, [-1.0, 20.0, 0.0]) == 0
    assert (pix == 32750).all()


def test_holemap_zero_range_point_is_skipped(oc, npo):
    """Deviation D1: dist = 0 -> add = inf -> NaN (:524-530) is defined as 'skip the ray'."""
    pix = uniform_map(64); b = pix.copy()
    xy = np.array([[0.0, 0.0], [2.0, 1.0]], np.float32)
    scale = oc.map_scale(64, 40.0)
    pxcs = oc.pose_to_pxcs([20, 20, 0], scale)
    n = oc.update_holemap_pxcs(pix, 64, scale, xy, pxcs)
    assert n == npo.update_holemap_pxcs(b, 64, scale, xy, pxcs) and n > 0
    assert (pix == b).all()


# ---- ObstacleMap :456-490, :540-593 ---------------------------------------------------------
def test_obstaclemap_kats(oc, npo, sim):
    size = 32
    scale = oc.map_scale(size, 40.0)
    o = np.full((size, size), -5, np.int8)
    xy = np.array([[5.0, 0.0]], np.float32)                 # one ray along +x: 4 px
    oc.update_obstaclemap(o, size, scale, xy, [20.0, 20.0, 0.0])
    # origin pixel (16,16) .. (19,16) traversed -> -4 ; endpoint (20,16) hit -> -4 ; rest -5
    assert list(o[16, 16:22]) == [-4, -4, -4, -4, -4, -5]
    # saturate at MaxObstacleHits (:474), decay of positive cells that another ray traverses (:586-589)
    o2 = np.full((size, size), 9, np.int8)
    xy2 = np.array([[5.0, 0.0], [5.0, 0.0], [10.0, 0.0]], np.float32)
    oc.update_obstaclemap(o2, size, scale, xy2, [20.0, 20.0, 0.0], 10)
    assert o2[16, 20] == 9          # 9 -> 10 (first hit), second hit blocked by max, then decayed by ray 3
    assert o2[16, 24] == 10 and o2[16, 19] == 8
    segs = sim.default_field()
    rays, xyf = sim.make_scan(segs, (20, 20, 0.3), 360)
    for size in (64, 100):
        a = np.full((size, size), -5, np.int8); b = a.copy()
        sc = oc.map_scale(size, 40.0)
        for it in range(8):
            pxcs = oc.pose_to_pxcs((20 + 0.2 * it, 20, 0.3 + 0.1 * it), sc)
            oc.update_obstaclemap_pxcs(a, size, xyf, pxcs, 10)
            npo.update_obstaclemap_pxcs(b, size, xyf, pxcs, 10)
            assert (a == b).all()
        assert a.max() > 0 and a.min() < 0


def test_pack_holemap(oc):
    """HoleMap.cs:44-55"""
    pix = np.array([0x1234, 0xF000, 0x0FFF, 0xFFFF], np.uint16)
    assert list(oc.pack_holemap(pix)) == [0x1F, 0x0F]


# ---- Hector KATs ----------------------------------------------------------------------------
def test_hector_logodds_and_prob(oc, npo):
    g = oc.Grid(0.1, 32, 32)
    lf, lo = g.logodds
    assert np.float32(lo) == np.float32(2.1972244)                    # OccGridMap.cs:25,47,86-90
    assert np.float32(lf) == np.float32(-0.40546516)                  # :24,46
    assert g.prob(5) == 0.5                                           # :101-102 untouched cell
    ng = npo.NpGrid(0.1, 32, 32)
    assert ng.lo_free == np.float32(lf) and ng.lo_occ == np.float32(lo)
    assert (g.cells["update_index"] == -1).all() and (g.cells["value"] == 0).all()   # LogOddsCell.cs:38-42


def test_hector_reset_cache_aliasing_d5(oc):
    """Deviation D5 (OccGridMap.cs:16-19,38-42,97-107,147,244-252), worked by hand from the C# text.

    Epoch bookkeeping of the reference: cacheArray[i].Index = -1 only in the constructor (:38-42); currCacheIndex starts at 0
    (:19), UpdateByScan ends with currCacheIndex++ (:147), Reset sets currCacheIndex = 0 (:248) and leaves cacheArray alone.

      scan A (ray to +5 cells): cell (15,10) occupied -> Value = logOddsOccupied = 2.1972244; currCacheIndex 0 -> 1
      GetCachedProbability(cell): Index -1 != 1 -> odds = e^2.1972244 = 9, 9/10 = 0.9 cached with Index 1            (:99-104)
      Reset(): Value 0, currCacheIndex = 0; the cache entry keeps (0.9, Index 1)
      GetCachedProbability(cell) now: Index 1 != 0 -> recomputed, e^0 / (e^0 + 1) = 0.5, cached with Index 0          (fresh)
      scan B (ray to +8 cells): cell (15,10) is crossed as FREE -> Value = logOddsFree = -0.40546516; currCacheIndex 0 -> 1
      ... the first query after the Reset was in epoch 0, so the entry holds Index 0 != 1: fresh again (0.4).
    The aliasing needs the SAME epoch number on both sides of the Reset with no query in between:
      new map, scan A (epoch 1), query (cached 0.9 @ Index 1), Reset (epoch 0), scan B (epoch 1), query:
      Index 1 == currCacheIndex 1 -> the reference returns the PRE-RESET 0.9 for a cell whose value says 0.4.
    The oracle's literal restatement shows exactly that; oracle_grid_prob -- what the library implements -- returns 0.4."""
    cell = 10 * 32 + 15
    scan_a = np.array([[5.0, 0.0]], np.float32)
    scan_b = np.array([[8.0, 0.0]], np.float32)
    pose = [10.0, 10.0, 0.0]
    p_occ, p_free = np.float32(0.9), np.float32(0.4)

    g = oc.Grid(1.0, 32, 32)
    g.update_by_scan(scan_a, pose)
    assert abs(np.float32(g.prob_literal(cell)) - p_occ) < 1e-6 and g.prob_literal(cell) == g.prob(cell)
    g.reset()
    g.update_by_scan(scan_b, pose)
    assert g.cells["value"][cell] == np.float32(g.logodds[0])                   # the map says "free"
    assert abs(np.float32(g.prob(cell)) - p_free) < 1e-6                        # chosen behaviour: the current value's probability
    assert abs(np.float32(g.prob_literal(cell)) - p_occ) < 1e-6                 # the C# cache: the map that was reset away
    # one more scan moves the epoch on and the literal cache recovers (B again: Value < 0 so a second "free" is added)
    g.update_by_scan(scan_b, pose)
    assert g.prob_literal(cell) == g.prob(cell)

    # a query between Reset and the next scan re-tags the entry with epoch 0: no aliasing (the working above)
    g = oc.Grid(1.0, 32, 32)
    g.update_by_scan(scan_a, pose)
    g.prob_literal(cell)
    g.reset()
    assert g.prob_literal(cell) == 0.5
    g.update_by_scan(scan_b, pose)
    assert g.prob_literal(cell) == g.prob(cell)
    # without a Reset the cache is value-transparent: every cell, after every scan
    g = oc.Grid(1.0, 32, 32)
    for s in (scan_a, scan_b, scan_a):
        g.update_by_scan(s, pose)
        for c in range(10 * 32 + 8, 10 * 32 + 20):
            assert g.prob_literal(c) == g.prob(c)


def test_hector_update_semantics(oc):
    """Quirks 21,22 (OccGridMap.cs:137,158-161,192-239)."""
    g = oc.Grid(1.0, 32, 32)
    lf, lo = [np.float32(v) for v in g.logodds]
    xy = np.array([[5.0, 0.0]], np.float32)
    g.update_by_scan(xy, [10.0, 10.0, 0.0])
    v = g.cells["value"].reshape(32, 32)
    assert list(v[10, 10:15]) == [lf] * 5 and v[10, 15] == lo and v[10, 16] == 0      # endpoint excluded from free
    assert g.cells["update_index"].reshape(32, 32)[10, 15] == 2 and g.cells["update_index"].reshape(32, 32)[10, 12] == 1
    # free-then-occupied within one scan: ((v + lf) - lf) + lo   (:206-214)
    g2 = oc.Grid(1.0, 32, 32)
    xy2 = np.array([[8.0, 0.0], [5.0, 0.0]], np.float32)
    g2.update_by_scan(xy2, [10.0, 10.0, 0.0])
    v2 = g2.cells["value"].reshape(32, 32)
    assert v2[10, 15] == np.float32(np.float32(np.float32(0) + lf) - lf) + lo
    assert v2[10, 18] == lo and v2[10, 16] == lf
    # begin == end is skipped (:137); end outside map is skipped entirely (:158-161)
    g3 = oc.Grid(1.0, 32, 32)
    g3.update_by_scan(np.array([[0.2, 0.1], [100.0, 0.0]], np.float32), [10.0, 10.0, 0.0])
    assert (g3.cells["value"] == 0).all()
    # occupied cap: Value < 50 (:211)
    g4 = oc.Grid(1.0, 32, 32)
    g4.cells["value"][10 * 32 + 15] = 50.0
    g4.update_by_scan(xy, [10.0, 10.0, 0.0])
    assert g4.cells["value"][10 * 32 + 15] == 50.0


def test_hector_round_half_even(oc):
    """VectorEx.cs:183-186 MathF.Round = banker's rounding on endpoints."""
    g = oc.Grid(1.0, 32, 32)
    g.update_by_scan(np.array([[2.5, 0.0]], np.float32), [10.0, 10.0, 0.0])      # 12.5 -> 12
    v = g.cells["value"].reshape(32, 32)
    assert v[10, 12] > 0 and v[10, 13] == 0
    g = oc.Grid(1.0, 32, 32)
    g.update_by_scan(np.array([[3.5, 0.0]], np.float32), [10.0, 10.0, 0.0])      # 13.5 -> 14
    v = g.cells["value"].reshape(32, 32)
    assert v[10, 14] > 0 and v[10, 13] < 0


def test_hector_interp_bounds_and_gradient(oc, npo):
    """Quirks 16,17 (MapProperties.cs:42,83-87; ScanMatcher.cs:235-248)."""
    g = oc.Grid(0.5, 16, 16)
    c = g.cells
    rng = np.random.default_rng(3)
    c["value"][:] = rng.normal(0, 2, 256).astype(np.float32)
    assert tuple(g.interp(-0.01, 3.0)) == (0, 0, 0)
    assert tuple(g.interp(14.01, 3.0)) == (0, 0, 0)          # Limits = Dim - 2
    assert g.interp(14.0, 3.0)[0] != 0
    assert tuple(g.interp(float("nan"), 3.0)) == (0, 0, 0)
    ng = npo.NpGrid(0.5, 16, 16)
    ng.value[:] = c["value"]
    for cx, cy in rng.uniform(0, 14, (50, 2)).astype(np.float32):
        a = g.interp(cx, cy)
        P, gx, gy = ng.interp(np.float32(cx), np.float32(cy))
        assert np.allclose(a, [P, gx, gy], rtol=0, atol=2e-7)
    # the gradient is the reference's (x-fraction for d/dx), not the textbook cross form
    cx, cy = np.float32(3.25), np.float32(4.75)
    p = [g.prob(4 * 16 + 3), g.prob(4 * 16 + 4), g.prob(5 * 16 + 3), g.prob(5 * 16 + 4)]
    fx, fy = np.float32(0.25), np.float32(0.75)
    exp_dx = -((np.float32(p[0]) - np.float32(p[1])) * (1 - fx) + (np.float32(p[2]) - np.float32(p[3])) * fx)
    assert abs(g.interp(cx, cy)[1] - exp_dx) < 1e-7


def test_hector_map_extends(oc):
    """GridMap.GetMapExtends (GridMap.cs:147-207): empty map, a populated rectangle, and the quirk that the minima start
    at 10000 (:150) so a region entirely beyond column 10000 reads as "nothing found"."""
    g = oc.Grid(1.0, 40, 24)
    assert g.map_extends() == (False, 0, 0, 0, 0)
    c = g.cells
    c["value"][5 * 40 + 7] = 0.4
    c["value"][19 * 40 + 31] = -0.4
    c["value"][11 * 40 + 2] = np.nan                                       # NaN != 0 counts
    assert g.map_extends() == (True, 31, 19, 2, 5)
    wide = oc.Grid(1.0, 10300, 4)
    wide.cells["value"][2 * 10300 + 10100] = 1.0
    assert wide.map_extends() == (False, 0, 0, 0, 0)                       # xMin never left 10000
    wide.cells["value"][1 * 10300 + 10000] = 1.0
    assert wide.map_extends() == (False, 0, 0, 0, 0)                       # x == 10000 is not < 10000 either
    wide.cells["value"][3 * 10300 + 9] = 1.0
    assert wide.map_extends() == (True, 10100, 3, 9, 1)


def test_hector_pyramid_shape(oc):
    """Quirk 23 (MapRepMultiMap.cs:49-57)."""
    lv = oc.make_pyramid(40.0 / 2048, 2048, 2048, 3)
    assert [(g.w, g.h) for g in lv] == [(2048, 2048), (1024, 1024), (512, 512)]
    assert lv[1].cell_len == float(np.float32(40.0 / 2048) * 2)
    lv = oc.make_pyramid(0.1, 401, 401, 3)
    assert [g.w for g in lv] == [401, 200, 100]


def test_hector_match_recovers_pose(oc, npo, sim):
    """End-to-end sanity: map 10 scans at the true pose, then match from a perturbed hint."""
    segs = sim.default_field()
    lv = oc.make_pyramid(0.1, 400, 400, 3)
    rays, xy = sim.make_scan(segs, (20, 20, 0.0), 360, sim.PCG32(5))
    for _ in range(10):
        for g in lv:
            g.update_by_scan(xy, [20.0, 20.0, 0.0])
    out = oc.match_pyramid(lv, xy, [20.15, 19.9, 0.03], [5, 4, 4], n_threads=4)
    assert abs(out[0] - 20) < 0.05 and abs(out[1] - 20) < 0.05 and abs(out[2]) < 0.01
    # chunked thread-order summation (quirk 18): T=1 and T=4 differ only by fp32 rounding
    H1, d1 = lv[0].hessian(xy, lv[0].map_pose([20.05, 20.0, 0.01]), 1)
    H4, d4 = lv[0].hessian(xy, lv[0].map_pose([20.05, 20.0, 0.01]), 4)
    assert np.allclose(H1, H4, rtol=1e-4) and np.allclose(d1, d4, rtol=1e-3, atol=1e-3)
    # NumPy restatement agrees bit-for-bit on H/dTr for both chunkings when both use det trig
    oc.set_trig_mode(oc.TRIG_DET)
    try:
        ng = npo.NpGrid(0.1, 400, 400)
        ng.value[:] = lv[0].cells["value"]
        for T in (1, 4):
            Hc, dc = lv[0].hessian(xy, lv[0].map_pose([20.05, 20.0, 0.01]), T)
            Hn, dn = ng.hessian(xy, lv[0].map_pose([20.05, 20.0, 0.01]), T)
            assert np.allclose(Hc, Hn, rtol=2e-5, atol=1e-5) and np.allclose(dc, dn, rtol=2e-5, atol=1e-4)
    finally:
        oc.set_trig_mode(oc.TRIG_LIBM)


def test_hector_grid_update_c_vs_numpy(oc, npo, sim):
    segs = sim.default_field()
    oc.set_trig_mode(oc.TRIG_DET)
    try:
        g = oc.Grid(0.2, 200, 200)
        ng = npo.NpGrid(0.2, 200, 200)
        for it in range(4):
            pose = (20 + 0.3 * it, 20 - 0.1 * it, 0.2 * it)
            rays, xy = sim.make_scan(segs, pose, 180, sim.PCG32(it))
            g.update_by_scan(xy, pose)
            ng.update_by_scan(xy, pose)
            assert (g.cells["update_index"] == ng.upd).all()
            assert (g.cells["value"] == ng.value).all()
    finally:
        oc.set_trig_mode(oc.TRIG_LIBM)


def test_hector_grid_hostile_points_c_vs_numpy(oc, npo, sim):
    """Far, duplicate, NaN and at-the-origin points, a scan origin off the robot, a rectangular grid: the two
    restatements of OccGridMap.UpdateByScan agree cell for cell."""
    segs = sim.default_field()
    oc.set_trig_mode(oc.TRIG_DET)
    try:
        g = oc.Grid(0.25, 120, 77)
        ng = npo.NpGrid(0.25, 120, 77)
        for it in range(3):
            pose = (14 + 0.3 * it, 9 - 0.1 * it, 0.2 * it)
            rays, xy = sim.make_scan(segs, pose, 90, sim.PCG32(it))
            xy = xy.copy()
            xy[0] = [3.0e4, -2.0e4]; xy[1] = xy[2]; xy[3] = [0.0, 0.0]; xy[4] = [np.nan, 1.0]
            g.update_by_scan(xy, pose, origin=(0.3, -0.2))
            ng.update_by_scan(xy, pose, origin=(0.3, -0.2))
            assert (g.cells["update_index"] == ng.upd).all()
            assert (g.cells["value"] == ng.value).all()
        assert (g.cells["value"] != 0).sum() > 50
    finally:
        oc.set_trig_mode(oc.TRIG_LIBM)



# ---- the device candidate generator's specification (what stands where FillRandomQueues stood, CoreSLAMProcessor.cs:599-612) -------
def test_philox4x32_10_known_answers(npo):
    """The NumPy restatement of the generator's integer stream against Random123's published kat_vectors (philox4x32, 10 rounds):
    all-zero, all-ones and the digits-of-pi counter / key."""
    for ctr, key, want in npo.PHILOX4X32_10_KAT:
        got = npo.philox4x32_10(np.array(ctr, np.uint64), key)
        assert tuple(int(x) for x in got) == want
    # vectorised over counters == one at a time
    ctrs = np.array([[i, 0, 5, 0] for i in range(7)], np.uint64)
    many = npo.philox4x32_10(ctrs, (42, 0))
    for i in range(7):
        assert (many[i] == npo.philox4x32_10(ctrs[i], (42, 0))).all()


def test_philox_jitters_structure(npo):
    """philox_jitters (the binary64 restatement of k_jitter): jitter i depends on (seed, stream, i) only; dtheta is the i-th of n strata of
    N(0, sigma_theta) -- ascending, each inside its stratum's quantile range; dx, dy are Box-Muller pairs of the block's first two words."""
    from scipy.special import ndtr
    n, sxy, sth = 4001, 0.1, math.radians(10.0)
    j = npo.philox_jitters(n, sxy, sth, seed=42, stream=3)
    assert j.shape == (n, 3) and np.isfinite(j).all()
    assert (np.diff(j[:, 2]) > 0).all()
    q = ndtr(j[:, 2] / np.float64(np.float32(sth)))
    i = np.arange(n)
    assert (q >= i / n - 1e-6).all() and (q <= (i + 1) / n + 1e-6).all()
    w, u = npo.philox_jitter_words(n, 42, 3)
    assert (w[5] == npo.philox4x32_10(np.array([5, 0, 3, 0], np.uint64), (42, 0))).all()
    r2 = (j[:, 0] ** 2 + j[:, 1] ** 2) / np.float64(np.float32(sxy)) ** 2
    assert np.allclose(r2, -2.0 * np.log(u[:, 0]), rtol=1e-6)
    assert abs(j[:, 0].std() - sxy) < 0.005 and abs(j[:, 1].std() - sxy) < 0.005
    # another stream / seed: another list
    assert not np.allclose(j, npo.philox_jitters(n, sxy, sth, seed=42, stream=4))
    assert not np.allclose(j, npo.philox_jitters(n, sxy, sth, seed=43, stream=3))
