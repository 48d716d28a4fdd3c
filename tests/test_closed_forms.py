"""Closed forms used by the fragment-parallel raster kernels vs the literal recurrences of the reference (CPU only).

K2 (csrc/holemap.hip): after i iterations of the error recurrence of DrawLaserRayOnHoleMap
(CoreSLAM/CoreSLAMProcessor.cs:394-396,:433-441) the walk has taken
    m(i) = min(i, max(0, ceil((2*dyc*i - dxc) / (2*dxc))))   minor steps.
K5 (csrc/hector.hip): after i steps of Bresenham2D (HectorSLAM/Map/OccGridMap.cs:220-239) the walk has taken
    (e0 + i*db) // da   minor steps, e0 = da // 2, db <= da.
"""


def k2_literal(dxc, dyc):
    error = 2 * dyc - dxc; horiz = 2 * dyc; diago = 2 * (dyc - dxc)
    m, out = 0, []
    for _ in range(dxc + 1):
        out.append(m)
        if error > 0:
            m += 1; error += diago
        else:
            error += horiz
    return out


def k2_closed(dxc, dyc, i):
    if dxc <= 0:
        return 0
    num = 2 * dyc * i - dxc
    if num <= 0:
        return 0
    den = 2 * dxc
    return min(i, (num + den - 1) // den)


def test_k2_minor_steps_closed_form():
    for dxc in range(0, 90):
        for dyc in range(0, dxc + 4):          # dyc > dxc can only arise from clipping round-off; covered too
            lit = k2_literal(dxc, dyc)
            assert lit == [k2_closed(dxc, dyc, i) for i in range(dxc + 1)], (dxc, dyc)
    for dxc, dyc in ((2047, 1), (2047, 2046), (2047, 2047), (4095, 1234), (1448, 1447), (32767, 32766)):
        lit = k2_literal(dxc, dyc)
        assert lit == [k2_closed(dxc, dyc, i) for i in range(dxc + 1)]


def k5_literal(da, db):
    err = da // 2
    m, out = 0, [0]
    for _ in range(da - 1):
        err += db
        if err >= da:
            m += 1; err -= da
        out.append(m)
    return out


def test_k5_minor_steps_closed_form():
    for da in range(1, 120):
        for db in range(0, da + 1):
            lit = k5_literal(da, db)
            assert lit == [(da // 2 + i * db) // da for i in range(da)], (da, db)
    for da, db in ((2047, 2047), (2047, 1), (4095, 4094), (1023, 511)):
        assert k5_literal(da, db) == [(da // 2 + i * db) // da for i in range(da)]


# ---- K2 V-profile (pixval) ------------------------------------------------------------------------------------
TS_NO_OBSTACLE, TS_OBSTACLE = 65500, 0


def _cs_div(a, b):
    """C# integer division: truncation toward zero."""
    q = abs(a) // abs(b)
    return q if (a < 0) == (b < 0) else -q


def k2_pixval_literal(dx, derrorv, x):
    """csrc/holemap.hip k2_pixval == the pixval recurrence of DrawLaserRayOnHoleMap (:397-428) up to step x."""
    incv = _cs_div(TS_OBSTACLE - TS_NO_OBSTACLE, derrorv)
    incerrorv = (TS_OBSTACLE - TS_NO_OBSTACLE) - derrorv * incv
    sincv = -1
    lim2, lim1 = dx - 2 * derrorv, dx - derrorv
    pixval, errorv = TS_NO_OBSTACLE, _cs_div(derrorv, 2)
    if x <= lim2:
        return pixval
    xs = 0 if lim2 < 0 else lim2 + 1
    for xi in range(xs, x + 1):
        if xi <= lim1:
            pixval += incv; errorv += incerrorv
            if errorv > derrorv:
                pixval += sincv; errorv -= derrorv
        else:
            pixval -= incv; errorv -= incerrorv
            if errorv < 0:
                pixval -= sincv; errorv += derrorv
    return pixval


def k2_pixval_closed(dx, derrorv, x):
    """Closed form used by the pixel kernels (valid because incerrorv <= 0 for TS_OBSTACLE < TS_NO_OBSTACLE): the
    descending half never carries; on the ascending half the carries fire on the first J steps only."""
    d = derrorv
    incv = _cs_div(TS_OBSTACLE - TS_NO_OBSTACLE, d)
    incerrorv = (TS_OBSTACLE - TS_NO_OBSTACLE) - d * incv
    sincv = -1
    assert incerrorv <= 0
    lim2, lim1 = dx - 2 * d, dx - d
    if x <= lim2:
        return TS_NO_OBSTACLE
    xs = 0 if lim2 < 0 else lim2 + 1
    n1 = max(0, min(x, lim1) - xs + 1)
    j = (x - xs + 1) - n1
    u0 = _cs_div(d, 2) + n1 * incerrorv
    g = -incerrorv
    J = 0
    if d - u0 > 0:
        J = (d - u0 + (g + d) - 1) // (g + d) - 1
    f = min(j, max(J, 0))
    return TS_NO_OBSTACLE + n1 * incv - j * incv - f * sincv


def test_k2_pixval_closed_form():
    for derrorv in list(range(1, 140)) + [255, 256, 257, 1000, 4097, 65499, 65500, 65501, 70000]:
        for dx in sorted(set([0, 1, 2, derrorv - 1, derrorv, derrorv + 1, 2 * derrorv - 1, 2 * derrorv, 2 * derrorv + 1,
                              3 * derrorv + 7, 5, 17, 100, 333])):
            if dx < 0 or dx > 1500:
                continue
            for x in range(0, dx + 1):
                assert k2_pixval_literal(dx, derrorv, x) == k2_pixval_closed(dx, derrorv, x), (dx, derrorv, x)


# ---- K3 ObstacleMap line (DrawLaserRayOnObstacleMap, CoreSLAMProcessor.cs:456-490) -----------------------------------
def k3_literal(dx, dy):
    """(x steps, y steps) taken before each iteration of the Rosetta-style walk, up to and including the end point."""
    err = _cs_div(dx if dx > dy else -dy, 2)
    ax = ay = 0
    out = []
    for _ in range(max(dx, dy) + 1):
        out.append((ax, ay))
        if ax == dx and ay == dy:
            break
        e2 = err
        if e2 > -dx:
            err -= dy; ax += 1
        if e2 < dy:
            err += dx; ay += 1
    return out


def _ceil_div(a, b):
    return -((-a) // b)


def k3_closed(dx, dy, i):
    """csrc/obstacle.hip: after i iterations the walk has taken i steps along the major axis and
    max(0, ceil((i*minor - e0) / major)) along the minor one, e0 = major / 2 (C# division: floor for major >= 0)."""
    if dx > dy:
        return (i, max(0, _ceil_div(i * dy - dx // 2, dx)))
    if dy == 0:
        return (0, 0)
    return (max(0, _ceil_div(i * dx - dy // 2, dy)), i)


def test_k3_walk_closed_form():
    for dx in range(0, 70):
        for dy in range(0, 70):
            lit = k3_literal(dx, dy)
            assert len(lit) == max(dx, dy) + 1 and lit[-1] == (dx, dy), (dx, dy)
            assert lit == [k3_closed(dx, dy, i) for i in range(max(dx, dy) + 1)], (dx, dy)
    for dx, dy in ((511, 1), (511, 510), (511, 511), (1023, 777), (2, 1023), (1023, 1022), (8191, 4097)):
        assert k3_literal(dx, dy) == [k3_closed(dx, dy, i) for i in range(max(dx, dy) + 1)]
