"""Independent NumPy/Python restatement of the slam.net hot path (TEST INFRASTRUCTURE ONLY).

Written separately from the C oracle (different structure: vectorised float32 NumPy for the
float parts, Python ints with explicit 32-bit wrapping for the integer parts) so that the two
restatements can pin each other: they must agree bit-for-bit on every integer output and on the
float32 intermediates.  Used only in tests/ and by oracle/gen_golden.py, which emits the small
fixtures under tests/golden/.  PARITY UNPINNED vs the C# reference itself (oracle/oracle.h).

Reference lines are cited per function (paths relative to /root/reference).
"""
import math

import numpy as np

F = np.float32
TS_NO_OBSTACLE = 65500      # CoreSLAM/CoreSLAMProcessor.cs:21
TS_OBSTACLE = 0             # :22
INT_MIN = -(2 ** 31)
INT_MAX = 2 ** 31 - 1


# ---- integer helpers (C# unchecked int) ----------------------------------------------------
def wrap32(v):
    v &= 0xFFFFFFFF
    return v - (1 << 32) if v & 0x80000000 else v


def cdiv(a, b):
    """C#/C truncating integer division."""
    q = abs(a) // abs(b)
    return q if (a >= 0) == (b >= 0) else -q


def sign(a):
    return (a > 0) - (a < 0)


def f2i(f):
    """(int)float on x64: truncate toward zero; NaN / out of range -> int.MinValue."""
    f = float(f)
    if math.isnan(f) or f >= 2147483648.0 or f <= -2147483904.0:
        return INT_MIN
    return int(f)


def f2i_array(a):
    a = np.asarray(a, dtype=np.float32)
    bad = ~np.isfinite(a) | (a >= F(2147483648.0)) | (a <= F(-2147483904.0))
    out = np.where(bad, 0, np.trunc(a)).astype(np.int64)
    out[bad] = INT_MIN
    return out


# ---- deterministic trig (same algorithm as oracle/det_trig.c, in float64 NumPy) ---------------
_TWO_OVER_PI = np.float64(6.36619772367581382433e-01)
_P1 = np.float64(1.57079632673412561417e+00)
_P2 = np.float64(6.07710050630396597660e-11)
_P2T = np.float64(2.02226624879595063154e-21)
_S = [np.float64(v) for v in (-1.66666666666666324348e-01, 8.33333333332248946124e-03,
                              -1.98412698298579493134e-04, 2.75573137070700676789e-06,
                              -2.50507602534068634195e-08, 1.58969099521155010221e-10)]
_C = [np.float64(v) for v in (4.16666666666666019037e-02, -1.38888888888741095749e-03,
                              2.48015872894767294178e-05, -2.75573143513906633035e-07,
                              2.08757232129817482790e-09, -1.13596475577881948265e-11)]


def det_sincos(a):
    """Vectorised deterministic sin/cos of float32 angles (|a| <= 65536), rounded once to f32."""
    a = np.asarray(a, dtype=np.float32)
    x = a.astype(np.float64)
    k = np.rint(x * _TWO_OVER_PI)
    r = ((x - k * _P1) - k * _P2) - k * _P2T
    z = r * r
    ps = _S[1] + z * (_S[2] + z * (_S[3] + z * (_S[4] + z * _S[5])))
    sn = r + (z * r) * (_S[0] + z * ps)
    pc = z * (_C[0] + z * (_C[1] + z * (_C[2] + z * (_C[3] + z * (_C[4] + z * _C[5])))))
    cs = 1.0 - (0.5 * z - z * pc)
    q = k.astype(np.int64) & 3
    so = np.choose(q, [sn, cs, -sn, -cs])
    co = np.choose(q, [cs, -sn, -cs, sn])
    return so.astype(np.float32), co.astype(np.float32)


# ---- BaseSLAM/MathEx.cs:116-138 ------------------------------------------------------------
def normalize_angle(angle):
    pi = F(math.pi)
    pi2 = F(pi * F(2.0))
    a = F(np.fmod(F(np.fmod(F(angle), pi2) + pi2), pi2))
    if a > pi:
        a = F(a - F(F(2.0) * pi))
    return a


# ---- CoreSLAM ------------------------------------------------------------------------------
def map_scale(size_px, size_m):                      # HoleMap.cs:20
    return F(F(size_px) / F(size_m))


def poses_to_pxcs(poses, scale, trig="det"):
    """CoreSLAMProcessor.cs:232-235 for a (K,3) float32 pose array."""
    poses = np.asarray(poses, np.float32).reshape(-1, 3)
    scale = F(scale)
    if trig == "det":
        s, c = det_sincos(poses[:, 2])
    else:
        s, c = np.sin(poses[:, 2]).astype(np.float32), np.cos(poses[:, 2]).astype(np.float32)
    out = np.empty((poses.shape[0], 4), np.float32)
    out[:, 0] = poses[:, 0] * scale + F(0.5)
    out[:, 1] = poses[:, 1] * scale + F(0.5)
    out[:, 2] = c * scale
    out[:, 3] = s * scale
    return out


def distance_batch_pxcs(pixels, size, xy, pxcs):
    """CoreSLAMProcessor.cs:226-259 for K candidates at once; returns int32[K]."""
    xy = np.asarray(xy, np.float32).reshape(-1, 2)
    pxcs = np.asarray(pxcs, np.float32).reshape(-1, 4)
    n = xy.shape[0]
    X = xy[None, :, 0]; Y = xy[None, :, 1]
    px = pxcs[:, 0:1]; py = pxcs[:, 1:2]; c = pxcs[:, 2:3]; s = pxcs[:, 3:4]
    fx = (px + c * X) - s * Y                                        # :240
    fy = (py + s * X) + c * Y                                        # :241
    ix = f2i_array(fx); iy = f2i_array(fy)
    ok = (ix >= 0) & (ix < size) & (iy >= 0) & (iy < size)           # :244
    idx = np.where(ok, iy * size + ix, 0)
    vals = np.where(ok, pixels.reshape(-1)[idx].astype(np.int64), 0)
    sums = vals.sum(axis=1)                                          # :246
    cnt = ok.sum(axis=1)
    # C# long division truncates toward zero; operands are non-negative here
    d = (sums * 1024) // max(n, 1)                                   # :253
    return np.where(cnt > 0, d, INT_MAX).astype(np.int32)            # :257


def argmin_first(dist):
    """First strictly smaller wins (CoreSLAMProcessor.cs:644, :700)."""
    return int(np.argmin(dist))      # np.argmin returns the first occurrence of the minimum


def search(pixels, size, scale, xy, search_pose, offs, trig="det"):
    """CoreSLAMProcessor.cs:624-653 / :695-705 with flat candidates (index 0 = base pose)."""
    sp = np.asarray(search_pose, np.float32)
    offs = np.asarray(offs, np.float32).reshape(-1, 3)
    poses = np.vstack([sp[None, :], sp[None, :] + offs]).astype(np.float32)   # :635-637
    d = distance_batch_pxcs(pixels, size, xy, poses_to_pxcs(poses, scale, trig))
    bi = argmin_first(d)
    return bi, poses[bi], int(d[bi]), d


def clip_ray(size, xyc, yxc, xy, yx):
    """CoreSLAMProcessor.cs:320-345.  Returns (ok, xyc, yxc)."""
    if xyc < 0:
        if xyc == xy:
            return False, xyc, yxc
        num = wrap32(wrap32(yxc - yx) * wrap32(-xyc)); den = wrap32(xyc - xy)
        if den == -1 and num == INT_MIN:
            return False, xyc, yxc
        yxc = wrap32(yxc + cdiv(num, den)); xyc = 0
    if xyc >= size:
        if xyc == xy:
            return False, xyc, yxc
        num = wrap32(wrap32(yxc - yx) * wrap32(size - 1 - xyc)); den = wrap32(xyc - xy)
        if den == -1 and num == INT_MIN:
            return False, xyc, yxc
        yxc = wrap32(yxc + cdiv(num, den)); xyc = size - 1
    return True, xyc, yxc


def ray_fragments(size, x1, y1, x2, y2, xp, yp, value=TS_OBSTACLE):
    """CoreSLAMProcessor.cs:359-443 without the blend: list of (ptr, pixval) in walk order,
    or None if the ray is skipped."""
    ok, x2c, y2c = clip_ray(size, x2, y2, x1, y1)                    # :365
    if not ok:
        return None
    ok, y2c, x2c = clip_ray(size, y2c, x2c, y1, x1)                  # :366
    if not ok:
        return None
    if not (0 <= x2c < size and 0 <= y2c < size):                    # deviation D4 (oracle.h / coreslam_oracle.c)
        return None
    ddx, ddy, ddxc, ddyc = wrap32(x2 - x1), wrap32(y2 - y1), wrap32(x2c - x1), wrap32(y2c - y1)
    if INT_MIN in (ddx, ddy, ddxc, ddyc):
        return None
    dx, dy, dxc, dyc = abs(ddx), abs(ddy), abs(ddxc), abs(ddyc)      # :368-371
    incx, incy = sign(ddx), wrap32(sign(ddy) * size)                 # :372-373
    sincv = sign(value - TS_NO_OBSTACLE)                             # :374
    if dx > dy:                                                      # :377
        t = wrap32(xp - x2)
    else:
        dx = dy; dxc, dyc = dyc, dxc; incx, incy = incy, incx        # :383-385
        t = wrap32(yp - y2)
    if t == INT_MIN:
        return None
    derrorv = abs(t)                                                 # :379 / :386
    if derrorv == 0:                                                 # :389
        return None
    error = wrap32(2 * dyc - dxc); horiz = wrap32(2 * dyc); diago = wrap32(2 * (dyc - dxc))   # :394-396
    errorv = derrorv // 2                                            # :397
    incv = cdiv(value - TS_NO_OBSTACLE, derrorv)                     # :398
    incerrorv = wrap32(value - TS_NO_OBSTACLE - derrorv * incv)      # :399
    ptr = wrap32(y1 * size + x1); pixval = TS_NO_OBSTACLE            # :401-402
    lim2, lim1 = wrap32(dx - 2 * derrorv), wrap32(dx - derrorv)
    out = []
    for x in range(0, dxc + 1):                                      # :404
        if x > lim2:                                                 # :406
            if x <= lim1:                                            # :408
                pixval = wrap32(pixval + incv); errorv = wrap32(errorv + incerrorv)
                if errorv > derrorv:
                    pixval = wrap32(pixval + sincv); errorv = wrap32(errorv - derrorv)
            else:
                pixval = wrap32(pixval - incv); errorv = wrap32(errorv - incerrorv)
                if errorv < 0:
                    pixval = wrap32(pixval - sincv); errorv = wrap32(errorv + derrorv)
        out.append((ptr, pixval))
        if error > 0:                                                # :433
            ptr = wrap32(ptr + incy); error = wrap32(error + diago)
        else:
            error = wrap32(error + horiz)
        ptr = wrap32(ptr + incx)
    return out


def blend(pix, pixval, alpha):
    """CoreSLAMProcessor.cs:431"""
    return (wrap32(wrap32((256 - alpha) * pix) + wrap32(alpha * pixval)) >> 8) & 0xFFFF


def holemap_rays(size, scale, xy, pxcs, hole_width):
    """CoreSLAMProcessor.cs:496-530: per-ray integer endpoints (x1,y1,x2,y2,xp,yp) or None."""
    xy = np.asarray(xy, np.float32).reshape(-1, 2)
    px, py, c, s = [F(v) for v in pxcs]
    scale = F(scale); hw = F(hole_width)
    x1, y1 = f2i(px), f2i(py)                                        # :505-506
    if x1 < 0 or x1 >= size or y1 < 0 or y1 >= size:                 # :509
        return None
    X = xy[:, 0]; Y = xy[:, 1]
    with np.errstate(all="ignore"):
        x2p = c * X - s * Y                                          # :519
        y2p = s * X + c * Y                                          # :520
        xp = f2i_array(px + x2p); yp = f2i_array(py + y2p)           # :521-522
        dist = np.sqrt(x2p * x2p + y2p * y2p)                        # :524
        add = hw * scale / F(2.0) / dist                             # :525
        x2 = f2i_array(px + x2p * (F(1.0) + add))                    # :527,:529
        y2 = f2i_array(py + y2p * (F(1.0) + add))                    # :528,:530
    rays = []
    for i in range(xy.shape[0]):
        if INT_MIN in (int(xp[i]), int(yp[i]), int(x2[i]), int(y2[i])):
            rays.append(None)                                        # deviation D1 (oracle.h)
        else:
            rays.append((x1, y1, int(x2[i]), int(y2[i]), int(xp[i]), int(yp[i])))
    return rays


def update_holemap_pxcs(pixels, size, scale, xy, pxcs, hole_width=0.6, quality=50):
    """CoreSLAMProcessor.cs:496-534 in place on a flat uint16 array; returns blended pixel count."""
    rays = holemap_rays(size, scale, xy, pxcs, hole_width)
    if rays is None:
        return 0
    total = 0
    npix = size * size
    for r in rays:
        if r is None:
            continue
        frags = ray_fragments(size, *r)
        if frags is None:
            continue
        for ptr, pixval in frags:
            if 0 <= ptr < npix:
                pixels[ptr] = blend(int(pixels[ptr]), pixval, quality)
                total += 1
    return total


def update_obstaclemap_pxcs(pixels, size, xy, pxcs, max_hits=10):
    """CoreSLAMProcessor.cs:540-593 (+ :456-490) in place on an int8 [size,size] array."""
    xy = np.asarray(xy, np.float32).reshape(-1, 2)
    px, py, c, s = [F(v) for v in pxcs]
    nohit = np.zeros((size, size), bool)                             # :542
    x0, y0 = f2i(px), f2i(py)                                        # :553-554
    if x0 < 0 or x0 >= size or y0 < 0 or y0 >= size:                 # :557
        return
    with np.errstate(all="ignore"):
        ex = f2i_array((px + c * xy[:, 0]) - s * xy[:, 1])           # :566
        ey = f2i_array((py + s * xy[:, 0]) + c * xy[:, 1])           # :567
    for i in range(xy.shape[0]):
        x1, y1, x2, y2 = x0, y0, int(ex[i]), int(ey[i])
        ddx, ddy = wrap32(x2 - x1), wrap32(y2 - y1)
        if ddx == INT_MIN or ddy == INT_MIN:
            continue
        dx, sx, dy, sy = abs(ddx), sign(ddx), abs(ddy), sign(ddy)    # :458-459
        err = cdiv(dx if dx > dy else -dy, 2)                        # :460
        while True:
            if x1 < 0 or x1 >= size or y1 < 0 or y1 >= size:         # :465
                break
            if x1 == x2 and y1 == y2:                                # :471
                if pixels[y1, x1] < max_hits:
                    pixels[y1, x1] += 1                              # :474-477
                break
            nohit[y1, x1] = True                                     # :483
            e2 = err
            if e2 > -dx:
                err = wrap32(err - dy); x1 = wrap32(x1 + sx)         # :487
            if e2 < dy:
                err = wrap32(err + dx); y1 = wrap32(y1 + sy)         # :488
    neg = nohit & (pixels < 0)                                       # :576-592
    pos = nohit & (pixels > 0)
    pixels[neg] += 1
    pixels[pos] -= 1


# ---- Hector --------------------------------------------------------------------------------
class M32:
    """System.Numerics.Matrix3x2 restated in float32 (row-vector convention)."""

    def __init__(self, m11, m12, m21, m22, m31, m32):
        self.m = [F(m11), F(m12), F(m21), F(m22), F(m31), F(m32)]

    @staticmethod
    def rotation(radians, trig="det"):
        pi = F(math.pi)
        radians = F(math.remainder(float(F(radians)), float(F(pi * F(2)))))   # IEEERemainder (exact op)
        eps = F(F(0.001) * pi / F(180.0))
        if -eps < radians < eps:
            c, s = F(1), F(0)
        elif F(pi / F(2) - eps) < radians < F(pi / F(2) + eps):
            c, s = F(0), F(1)
        elif radians < F(-pi + eps) or radians > F(pi - eps):
            c, s = F(-1), F(0)
        elif F(-pi / F(2) - eps) < radians < F(-pi / F(2) + eps):
            c, s = F(0), F(-1)
        elif trig == "det":
            s, c = det_sincos(np.array([radians], np.float32)); s, c = F(s[0]), F(c[0])
        else:
            s, c = F(math.sin(radians)), F(math.cos(radians))
        return M32(c, s, -s, c, 0, 0)

    @staticmethod
    def translation(x, y):
        return M32(1, 0, 0, 1, x, y)

    @staticmethod
    def scale(s):
        return M32(s, 0, 0, s, 0, 0)

    def __mul__(a, b):
        a11, a12, a21, a22, a31, a32 = a.m
        b11, b12, b21, b22, b31, b32 = b.m
        return M32(a11 * b11 + a12 * b21, a11 * b12 + a12 * b22,
                   a21 * b11 + a22 * b21, a21 * b12 + a22 * b22,
                   a31 * b11 + a32 * b21 + b31, a31 * b12 + a32 * b22 + b32)

    def invert(self):
        m11, m12, m21, m22, m31, m32 = self.m
        det = F(m11 * m22) - F(m21 * m12)
        inv = F(1.0) / det
        return M32(m22 * inv, -m12 * inv, -m21 * inv, m11 * inv,
                   (m21 * m32 - m31 * m22) * inv, (m31 * m12 - m11 * m32) * inv)

    def transform(self, x, y):
        m11, m12, m21, m22, m31, m32 = self.m
        x = np.asarray(x, np.float32); y = np.asarray(y, np.float32)
        return x * m11 + y * m21 + m31, x * m12 + y * m22 + m32


class NpGrid:
    """OccGridMap (HectorSLAM/Map/OccGridMap.cs + GridMap.cs) with value / update_index arrays."""

    def __init__(self, cell_len, w, h, trig="det"):
        self.w, self.h, self.cell = w, h, F(cell_len)
        self.stm = F(F(1.0) / self.cell)                               # MapProperties.cs:32
        self.value = np.zeros(w * h, np.float32)
        self.upd = np.full(w * h, -1, np.int32)
        self.cur = 0
        self.trig = trig
        # MathF.Log / MathF.Exp: evaluated in float64 and rounded once (NumPy's float32 SIMD log/exp
        # are not correctly rounded; glibc's logf/expf are, to within double-rounding cases)
        self.lo_free = F(math.log(float(F(F(0.4) / F(F(1.0) - F(0.4))))))   # OccGridMap.cs:46,86-90
        self.lo_occ = F(math.log(float(F(F(0.9) / F(F(1.0) - F(0.9))))))    # :47
        self.map_t_world = M32.scale(self.stm) * M32.translation(0, 0)  # GridMap.cs:46
        self.world_t_map = self.map_t_world.invert()

    def prob(self, idx):                                                # OccGridMap.cs:97-107
        odds = np.exp(self.value[idx].astype(np.float64)).astype(np.float32)
        return (odds / (odds + F(1.0))).astype(np.float32)

    def update_by_scan(self, xy, pose, origin=(0.0, 0.0)):              # OccGridMap.cs:114-148
        xy = np.asarray(xy, np.float32).reshape(-1, 2)
        mark_free, mark_occ = self.cur + 1, self.cur + 2
        t = M32.rotation(pose[2], self.trig) * M32.translation(pose[0], pose[1]) * M32.scale(self.stm)
        bxf, byf = t.transform(F(origin[0]), F(origin[1]))
        bx, by = f2i(np.rint(bxf)), f2i(np.rint(byf))                   # ToRoundPoint (banker's)
        exf, eyf = t.transform(xy[:, 0], xy[:, 1])
        ex = f2i_array(np.rint(exf)); ey = f2i_array(np.rint(eyf))
        W = self.w
        for i in range(xy.shape[0]):
            x2, y2 = int(ex[i]), int(ey[i])
            if (bx, by) == (x2, y2):                                    # :137
                continue
            if not (0 <= bx < W and 0 <= by < self.h and 0 <= x2 < W and 0 <= y2 < self.h):   # :158
                continue
            dx, dy = x2 - bx, y2 - by
            adx, ady = abs(dx), abs(dy)
            odx, ody = sign(dx), sign(dy) * W
            off = by * W + bx
            if adx >= ady:
                da, db, err, oa, ob = adx, ady, adx // 2, odx, ody      # :175-179
            else:
                da, db, err, oa, ob = ady, adx, ady // 2, ody, odx      # :180-185
            cells = [off]
            for _ in range(da - 1):                                     # :226
                off += oa; err += db
                if err >= da:
                    off += ob; err -= da
                cells.append(off)
            for cidx in cells:                                          # BresenhamCellFree :192-199
                if self.upd[cidx] < mark_free:
                    self.value[cidx] = self.value[cidx] + self.lo_free
                    self.upd[cidx] = mark_free
            e = y2 * W + x2                                             # BresenhamCellOcc :201-218
            if self.upd[e] < mark_occ:
                if self.upd[e] == mark_free:
                    self.value[e] = self.value[e] - self.lo_free
                if self.value[e] < F(50.0):
                    self.value[e] = self.value[e] + self.lo_occ
                self.upd[e] = mark_occ
        self.cur += 3

    def interp(self, cx, cy):                                           # ScanMatcher.cs:211-249
        cx = np.asarray(cx, np.float32); cy = np.asarray(cy, np.float32)
        with np.errstate(invalid="ignore"):
            oob = np.isnan(cx) | np.isnan(cy) | (cx < 0) | (cx > F(self.w - 2.0)) | (cy < 0) | (cy > F(self.h - 2.0))
        cxs = np.where(oob, F(0), cx); cys = np.where(oob, F(0), cy)
        ix = np.floor(cxs).astype(np.int64); iy = np.floor(cys).astype(np.int64)
        fx = cxs - ix.astype(np.float32); fy = cys - iy.astype(np.float32)
        idx = iy * self.w + ix
        i0, i1, i2, i3 = self.prob(idx), self.prob(idx + 1), self.prob(idx + self.w), self.prob(idx + self.w + 1)
        xi, yi = F(1.0) - fx, F(1.0) - fy
        P = ((i0 * xi + i1 * fx) * yi) + ((i2 * xi + i3 * fx) * fy)
        gx = -(((i0 - i1) * xi) + ((i2 - i3) * fx))
        gy = -(((i0 - i2) * yi) + ((i1 - i3) * fy))
        z = F(0)
        return np.where(oob, z, P), np.where(oob, z, gx), np.where(oob, z, gy)

    def hessian(self, xy, pose, n_threads=1):                           # ScanMatcher.cs:135-204
        xy = np.asarray(xy, np.float32).reshape(-1, 2)
        t = M32.rotation(pose[2], self.trig) * M32.translation(F(pose[0]) * self.cell, F(pose[1]) * self.cell) \
            * M32.scale(self.stm)
        if self.trig == "det":
            s, c = det_sincos(np.array([pose[2]], np.float32)); s, c = F(s[0]), F(c[0])
        else:
            s, c = F(math.sin(F(pose[2]))), F(math.cos(F(pose[2])))
        sinRot, cosRot = s * self.stm, c * self.stm
        X, Y = xy[:, 0], xy[:, 1]
        mx, my = t.transform(X, Y)
        P, gx, gy = self.interp(mx, my)
        fun = F(1.0) - P
        rot = ((-sinRot * X - cosRot * Y) * gx + (cosRot * X - sinRot * Y) * gy)
        terms = np.stack([gx * fun, gy * fun, rot * fun, gx * gx, gy * gy, rot * rot, gx * gy, gx * rot, gy * rot])
        n = xy.shape[0]
        chunk = (n + n_threads - 1) // n_threads
        tot = np.zeros(9, np.float32)
        for th in range(n_threads):
            loc = np.zeros(9, np.float32)
            for i in range(th * chunk, min(n, (th + 1) * chunk)):       # sequential fp32 accumulation
                loc = loc + terms[:, i]
            tot = tot + loc
        dTr = tot[0:3]
        H = np.array([[tot[3], tot[6], tot[7]], [tot[6], tot[4], tot[8]], [tot[7], tot[8], tot[5]]], np.float32)
        return H, dTr


# ---- the device candidate generator (what replaces FillRandomQueues, CoreSLAMProcessor.cs:599-612) -------------------------------
# Philox4x32-10 as published (Salmon, Moraes, Dror, Shaw: "Parallel random numbers: as easy as 1, 2, 3", SC'11; Random123 1.x
# philox.h): multipliers 0xD2511F53 / 0xCD9E8D57, Weyl key increments 0x9E3779B9 / 0xBB67AE85, ten rounds.  The reference's
# generator (Redzen's ziggurat, entropy-seeded) is not reproducible, so THIS is the specification of the library's own stream:
# libslamhip's k_jitter (csrc/coreslam.hip) must produce the same integers, and -- through float logf / sqrtf / sincosf /
# normcdfinvf, which are the device library's and a few ulp from exact -- floats within a few ulp of philox_jitters().
PHILOX_M0, PHILOX_M1 = 0xD2511F53, 0xCD9E8D57
PHILOX_W0, PHILOX_W1 = 0x9E3779B9, 0xBB67AE85
# known answers: Random123's kat_vectors for philox4x32, 10 rounds (counter[4], key[2] -> output[4])
PHILOX4X32_10_KAT = (
    ((0x00000000, 0x00000000, 0x00000000, 0x00000000), (0x00000000, 0x00000000), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
    ((0xffffffff, 0xffffffff, 0xffffffff, 0xffffffff), (0xffffffff, 0xffffffff), (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
    ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0), (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1)),
)


def philox4x32_10(ctr, key):
    """ctr: (..., 4) uint32 counters, key: (k0, k1) -> (..., 4) uint32.  Vectorised over leading dimensions."""
    c = np.asarray(ctr, dtype=np.uint64) & np.uint64(0xFFFFFFFF)
    c0, c1, c2, c3 = c[..., 0], c[..., 1], c[..., 2], c[..., 3]
    k0, k1 = int(key[0]) & 0xFFFFFFFF, int(key[1]) & 0xFFFFFFFF
    mask = np.uint64(0xFFFFFFFF)
    for _ in range(10):
        p0 = np.uint64(PHILOX_M0) * c0
        p1 = np.uint64(PHILOX_M1) * c2
        n0 = (p1 >> np.uint64(32)) ^ c1 ^ np.uint64(k0)
        n1 = p1 & mask
        n2 = (p0 >> np.uint64(32)) ^ c3 ^ np.uint64(k1)
        n3 = p0 & mask
        c0, c1, c2, c3 = n0, n1, n2, n3
        k0 = (k0 + PHILOX_W0) & 0xFFFFFFFF
        k1 = (k1 + PHILOX_W1) & 0xFFFFFFFF
    return np.stack([c0, c1, c2, c3], axis=-1).astype(np.uint32)


def philox_jitter_words(n, seed, stream):
    """The three uniforms of jitter i = 0 .. n-1 as the kernel forms them: counter (i, 0, stream_lo, stream_hi), key (seed_lo, seed_hi);
    u = ((float)(word >> 8) + 0.5f) * 2^-24 of words 0, 1, 2 IN BINARY32 -- from 2^23 on the sum is not representable and rounds to
    even (so u can be exactly 1.0: its logarithm is 0, the jitter (0, 0)); the scaling is exact."""
    i = np.arange(n, dtype=np.uint64)
    ctr = np.stack([i, np.zeros(n, np.uint64), np.full(n, stream & 0xFFFFFFFF, np.uint64), np.full(n, (stream >> 32) & 0xFFFFFFFF, np.uint64)], axis=-1)
    w = philox4x32_10(ctr, (seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF))
    u = (((w[:, :3] >> np.uint32(8)).astype(np.float32) + np.float32(0.5)) * np.float32(1.0 / 16777216.0)).astype(np.float64)
    return w, u


def philox_jitters(n, sigma_xy, sigma_theta, seed=0, stream=0):
    """k_jitter restated in binary64 (csrc/coreslam.hip): dx, dy ~ N(0, sigma_xy) by Box-Muller from u1, u2; dtheta = the i-th of n
    equal-probability strata of N(0, sigma_theta), at position u3 inside the stratum (quantile capped below 1).  Returns
    float64 (n, 3): the exact values the device's float results are a few ulp from."""
    from scipy.special import ndtri
    _, u = philox_jitter_words(n, seed, stream)
    u1, u2, u3 = u[:, 0], u[:, 1], u[:, 2]
    with np.errstate(divide="ignore"):
        rad = np.sqrt(np.float32(-2.0) * np.log(u1))
    ang = (np.float32(6.28318530718) * u2.astype(np.float32)).astype(np.float64)      # (one binary32 rounding, as the kernel: sincosf's argument)
    i = np.arange(n, dtype=np.float64)
    qf = ((np.float32(1) * i.astype(np.float32) + u3.astype(np.float32)) / np.float32(n)).astype(np.float32)   # binary32, as the kernel
    q = np.minimum(qf, np.float32(0.99999994)).astype(np.float64)
    sxy, sth = np.float64(np.float32(sigma_xy)), np.float64(np.float32(sigma_theta))
    return np.stack([sxy * rad * np.cos(ang), sxy * rad * np.sin(ang), sth * ndtri(q)], axis=-1)
