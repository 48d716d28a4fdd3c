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
