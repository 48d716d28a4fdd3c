"""Synthetic world + lidar scan generator (headless stand-in for the reference's WPF simulator).

Reproduces the *inputs* the reference simulator feeds the processors, not its code:
  - field: the two closed polygons of Simulation/Field.cs:45-59 and :63-69 with scale 30 m and
    offset (5, 5) in a 40 m world (Simulation/MainWindow.xaml.cs:97);
  - ray i has angle i*2*pi/R by INTEGER index (the reference's float-accumulated loop,
    MainWindow.xaml.cs:391, yields 361 rays for R=360), cast from the true pose out to
    maxScanDist = 40 m (:37); misses are dropped (:395);
  - range noise: uniform integer in [-100, 99] / 100 * 0.02 m (:38, :397) from a seeded PCG32
    (the reference uses an unseeded System.Random, so no stream can be matched);
  - cloud points in the robot frame: (r*cos(a), r*sin(a)) in float32 (:170-174).
Box2D ray casting (un-vendored submodule) is replaced by an analytic ray/segment intersector.
"""
import math

import numpy as np

_OUTER = [(0.00, 0.0), (1.00, 0.0), (1.00, 0.2), (0.80, 0.3), (0.80, 0.5), (1.00, 0.4),
          (1.00, 1.0), (0.6, 1.0), (0.6, 0.8), (0.5, 0.8), (0.5, 1.0), (0.0, 1.0)]
_INNER = [(0.2, 0.3), (0.3, 0.3), (0.4, 0.7), (0.3, 0.7)]

MAX_SCAN_DIST = 40.0
MEASURE_ERROR = 0.02


class PCG32:
    """Minimal PCG-XSH-RR 64/32 so fixtures do not depend on NumPy's Generator internals."""

    def __init__(self, seed=1234, seq=54):
        self.state = 0
        self.inc = ((seq << 1) | 1) & 0xFFFFFFFFFFFFFFFF
        self.next_u32()
        self.state = (self.state + seed) & 0xFFFFFFFFFFFFFFFF
        self.next_u32()

    def next_u32(self):
        old = self.state
        self.state = (old * 6364136223846793005 + self.inc) & 0xFFFFFFFFFFFFFFFF
        xorshifted = (((old >> 18) ^ old) >> 27) & 0xFFFFFFFF
        rot = old >> 59
        return ((xorshifted >> rot) | (xorshifted << ((-rot) & 31))) & 0xFFFFFFFF

    def randint(self, lo, hi):
        """Uniform integer in [lo, hi)."""
        return lo + self.next_u32() % (hi - lo)

    def uniform(self):
        return self.next_u32() / 4294967296.0

    def normal_pair(self):
        u1 = (self.next_u32() + 1.0) / 4294967297.0
        u2 = self.next_u32() / 4294967296.0
        r = math.sqrt(-2.0 * math.log(u1))
        return r * math.cos(2 * math.pi * u2), r * math.sin(2 * math.pi * u2)


def default_field(scale=30.0, offset=(5.0, 5.0)):
    """Wall segments (N,4) float64: x0,y0,x1,y1."""
    segs = []
    for poly in (_OUTER, _INNER):
        pts = [(offset[0] + x * scale, offset[1] + y * scale) for x, y in poly]
        for i in range(len(pts)):
            a, b = pts[i], pts[(i + 1) % len(pts)]
            segs.append((a[0], a[1], b[0], b[1]))
    return np.array(segs, np.float64)


def raycast(segs, pos, angles, max_dist=MAX_SCAN_DIST):
    """Nearest hit distance per angle (inf when nothing within max_dist)."""
    ox, oy = float(pos[0]), float(pos[1])
    dx = np.cos(angles)[:, None]; dy = np.sin(angles)[:, None]
    x0, y0, x1, y1 = segs[:, 0][None, :], segs[:, 1][None, :], segs[:, 2][None, :], segs[:, 3][None, :]
    ex, ey = x1 - x0, y1 - y0
    den = dx * ey - dy * ex
    with np.errstate(divide="ignore", invalid="ignore"):
        t = ((x0 - ox) * ey - (y0 - oy) * ex) / den      # along the ray
        u = ((x0 - ox) * dy - (y0 - oy) * dx) / den      # along the segment
    ok = (np.abs(den) > 1e-12) & (t >= 0) & (t <= max_dist) & (u >= 0) & (u <= 1)
    t = np.where(ok, t, np.inf)
    return t.min(axis=1)


def make_scan(segs, true_pose, n_rays, rng=None, noise=True):
    """Returns (rays (M,2) float32 [angle, radius], xy (M,2) float32 robot-frame points)."""
    idx = np.arange(n_rays)
    angles = (idx * (2.0 * math.pi / n_rays)).astype(np.float32)       # ray.Angle (lidar frame)
    hit = raycast(segs, true_pose[:2], angles.astype(np.float64) + float(true_pose[2]))
    keep = np.isfinite(hit)
    r = hit.copy()
    if noise:
        rng = rng or PCG32()
        for i in range(n_rays):
            e = rng.randint(-100, 100) / 100.0 * MEASURE_ERROR
            if keep[i]:
                r[i] += e
    angles = angles[keep]; r = r[keep].astype(np.float32)
    rays = np.stack([angles, r], axis=1).astype(np.float32)
    xy = np.stack([r * np.cos(angles).astype(np.float32), r * np.sin(angles).astype(np.float32)], axis=1)
    return rays, xy.astype(np.float32)


def trajectory(n, start=(20.0, 20.0, 0.0), step=(0.05, 0.02, math.radians(0.5))):
    """Fixed mapping trajectory of SURVEY.md sec.8d."""
    p = np.zeros((n, 3), np.float64)
    for i in range(n):
        p[i] = (start[0] + i * step[0], start[1] + i * step[1], start[2] + i * step[2])
    return p.astype(np.float32)


def lap_trajectory(n=None, step=0.1):
    """True poses on an ellipse around the inner obstacle (x 11..17 m, y 14..26 m), heading along the tangent; consecutive
    poses are `step` metres apart; n = None -> one full lap."""
    cx, cy, ax, ay = 14.0, 20.0, 6.5, 9.0
    # arc-length parametrisation by fine sampling
    t = np.linspace(0.0, 2.0 * math.pi, 200001)
    x, y = cx + ax * np.cos(t), cy + ay * np.sin(t)
    s = np.concatenate([[0.0], np.cumsum(np.hypot(np.diff(x), np.diff(y)))])
    total = s[-1]
    if n is None:
        n = int(total / step)
    want = (np.arange(n) * step) % total
    ti = np.interp(want, s, t)
    px, py = cx + ax * np.cos(ti), cy + ay * np.sin(ti)
    th = np.arctan2(ay * np.cos(ti), -ax * np.sin(ti))
    return np.stack([px, py, th], 1).astype(np.float32), total


def gaussian_offsets(n, sigma_xy=0.1, sigma_theta=math.radians(10.0), seed=42):
    """(n,3) float32 jitter list: N(0,sigma_xy) for x,y and N(0,sigma_theta) for theta
    (Simulation/MainWindow.xaml.cs:69 values), drawn in the reference's X,Y,theta order
    (CoreSLAMProcessor.cs:635-637) from a seeded generator of our own (Redzen is entropy-seeded)."""
    g = np.random.Generator(np.random.Philox(seed))
    o = g.standard_normal((n, 3))
    o[:, 0:2] *= sigma_xy
    o[:, 2] *= sigma_theta
    return o.astype(np.float32)
