"""ctypes loader for the C parity oracle (oracle/_build/liboracle.so).

TEST INFRASTRUCTURE ONLY -- imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  The product package (slam.net_amd) never imports this module.
PARITY UNPINNED: see oracle/oracle.h.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "liboracle.so")

TRIG_LIBM, TRIG_DET = 0, 1


def build(force=False):
    srcs = ["det_trig.c", "coreslam_oracle.c", "hector_oracle.c", "cpu_baseline.c", "oracle.h", "Makefile"]
    stale = force or not os.path.exists(_SO) or any(
        os.path.getmtime(os.path.join(_HERE, s)) > os.path.getmtime(_SO) for s in srcs)
    if stale:
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        try:
            build()
        except Exception:
            if not os.path.exists(_SO):
                raise
        _lib = C.CDLL(_SO)
        _declare(_lib)
    return _lib


def _p(a, ct):
    return a.ctypes.data_as(C.POINTER(ct))


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _declare(L):
    f, i32, i64, vp = C.c_float, C.c_int32, C.c_int64, C.c_void_p
    fp, ip, u16p, i8p, u8p = (C.POINTER(C.c_float), C.POINTER(C.c_int32), C.POINTER(C.c_uint16),
                              C.POINTER(C.c_int8), C.POINTER(C.c_uint8))
    L.oracle_set_trig_mode.argtypes = [C.c_int]
    L.oracle_get_trig_mode.restype = C.c_int
    L.oracle_cosf.argtypes = [f]; L.oracle_cosf.restype = f
    L.oracle_sinf.argtypes = [f]; L.oracle_sinf.restype = f
    L.oracle_det_sincosf.argtypes = [f, fp, fp]
    L.oracle_normalize_angle.argtypes = [f]; L.oracle_normalize_angle.restype = f
    L.oracle_deg_diff.argtypes = [f, f]; L.oracle_deg_diff.restype = f
    L.oracle_map_scale.argtypes = [C.c_int, f]; L.oracle_map_scale.restype = f
    L.oracle_cs_pose_to_pxcs.argtypes = [fp, f, fp]
    L.oracle_cs_distance_pxcs.argtypes = [u16p, C.c_int, fp, C.c_int, fp]; L.oracle_cs_distance_pxcs.restype = i32
    L.oracle_cs_distance.argtypes = [u16p, C.c_int, f, fp, C.c_int, fp]; L.oracle_cs_distance.restype = i32
    L.oracle_cs_distance_batch_pxcs.argtypes = [u16p, C.c_int, fp, C.c_int, fp, C.c_int, ip, ip]
    L.oracle_cs_distance_batch_pxcs.restype = i32
    L.oracle_cs_search.argtypes = [u16p, C.c_int, f, fp, C.c_int, fp, fp, C.c_int, fp, ip, ip]
    L.oracle_cs_search.restype = i32
    L.oracle_cs_clip_ray.argtypes = [C.c_int, ip, ip, C.c_int, C.c_int]; L.oracle_cs_clip_ray.restype = C.c_int
    L.oracle_cs_draw_ray_holemap.argtypes = [u16p] + [C.c_int] * 9; L.oracle_cs_draw_ray_holemap.restype = C.c_int
    L.oracle_cs_update_holemap.argtypes = [u16p, C.c_int, f, fp, C.c_int, fp, f, C.c_int]
    L.oracle_cs_update_holemap.restype = i64
    L.oracle_cs_update_holemap_pxcs.argtypes = [u16p, C.c_int, f, fp, C.c_int, fp, f, C.c_int]
    L.oracle_cs_update_holemap_pxcs.restype = i64
    L.oracle_cs_update_obstaclemap.argtypes = [i8p, u8p, C.c_int, f, fp, C.c_int, fp, C.c_int]
    L.oracle_cs_update_obstaclemap_pxcs.argtypes = [i8p, u8p, C.c_int, fp, C.c_int, fp, C.c_int]
    L.oracle_cs_segments_to_cloud.argtypes = [fp, ip, C.c_int, fp, fp, fp]
    L.oracle_cs_pack_holemap.argtypes = [u16p, C.c_int, u8p]
    L.oracle_csproc_create.argtypes = [f, C.c_int, C.c_int, fp]; L.oracle_csproc_create.restype = vp
    L.oracle_csproc_destroy.argtypes = [vp]
    L.oracle_csproc_reset.argtypes = [vp]
    L.oracle_csproc_set_params.argtypes = [vp, C.c_int, f, C.c_int, C.c_int, C.c_int]
    L.oracle_csproc_update.argtypes = [vp, fp, ip, C.c_int, fp, fp, C.c_int]
    L.oracle_csproc_get_pose.argtypes = [vp, fp]
    L.oracle_csproc_holemap.argtypes = [vp]; L.oracle_csproc_holemap.restype = u16p
    L.oracle_csproc_obstaclemap.argtypes = [vp]; L.oracle_csproc_obstaclemap.restype = i8p
    L.oracle_grid_create.argtypes = [f, C.c_int, C.c_int, f, f]; L.oracle_grid_create.restype = vp
    L.oracle_grid_destroy.argtypes = [vp]
    L.oracle_grid_reset.argtypes = [vp]
    L.oracle_grid_set_factors.argtypes = [vp, f, f]
    L.oracle_grid_get_logodds.argtypes = [vp, fp, fp]
    L.oracle_grid_cells.argtypes = [vp]; L.oracle_grid_cells.restype = vp
    L.oracle_grid_w.argtypes = [vp]; L.oracle_grid_w.restype = C.c_int
    L.oracle_grid_h.argtypes = [vp]; L.oracle_grid_h.restype = C.c_int
    L.oracle_grid_prob.argtypes = [vp, C.c_int]; L.oracle_grid_prob.restype = f
    L.oracle_grid_prob_literal.argtypes = [vp, C.c_int]; L.oracle_grid_prob_literal.restype = f
    L.oracle_grid_map_pose.argtypes = [vp, fp, fp]
    L.oracle_grid_world_pose.argtypes = [vp, fp, fp]
    L.oracle_grid_update_by_scan.argtypes = [vp, fp, C.c_int, fp, fp]
    L.oracle_grid_bitmap.argtypes = [vp, u8p]
    L.oracle_grid_map_extends.argtypes = [vp, C.POINTER(C.c_int)]; L.oracle_grid_map_extends.restype = C.c_int
    L.oracle_hs_interp.argtypes = [vp, f, f, fp]
    L.oracle_hs_hessian.argtypes = [vp, fp, C.c_int, fp, C.c_int, fp, fp]
    L.oracle_hs_estimate_step.argtypes = [vp, fp, C.c_int, fp, C.c_int]; L.oracle_hs_estimate_step.restype = C.c_int
    L.oracle_hs_match_grid.argtypes = [vp, fp, C.c_int, fp, C.c_int, C.c_int, fp]
    L.oracle_hs_match_pyramid.argtypes = [C.POINTER(vp), C.c_int, fp, C.c_int, fp, ip, C.c_int, fp]
    L.oracle_cpu_baseline_search.argtypes = [u16p, C.c_int, f, fp, C.c_int, fp, fp, C.c_int, C.c_int, C.c_int,
                                             C.POINTER(i64), ip, ip]
    L.oracle_cpu_baseline_search.restype = C.c_double
    L.oracle_cpu_baseline_search_timed.argtypes = [u16p, C.c_int, f, fp, C.c_int, fp, fp, C.c_int, C.c_int, C.c_int,
                                                   C.POINTER(i64), ip, ip, C.POINTER(C.c_double)]
    L.oracle_cpu_baseline_search_timed.restype = C.c_double


CELL_DTYPE = np.dtype([("update_index", np.int32), ("value", np.float32)])


# ---- trig / math ----------------------------------------------------------------------------
def set_trig_mode(mode):
    lib().oracle_set_trig_mode(int(mode))


def det_sincos(a):
    s, c = C.c_float(), C.c_float()
    lib().oracle_det_sincosf(C.c_float(a), C.byref(s), C.byref(c))
    return s.value, c.value


def det_sincos_array(a):
    a = _f32(a).ravel()
    s = np.empty_like(a); c = np.empty_like(a)
    L = lib()
    sv, cv = C.c_float(), C.c_float()
    for i, v in enumerate(a):
        L.oracle_det_sincosf(C.c_float(v), C.byref(sv), C.byref(cv))
        s[i] = sv.value; c[i] = cv.value
    return s, c


def normalize_angle(a):
    return lib().oracle_normalize_angle(C.c_float(a))


def deg_diff(a, b):
    return lib().oracle_deg_diff(C.c_float(a), C.c_float(b))


def map_scale(size_px, size_m):
    return lib().oracle_map_scale(int(size_px), C.c_float(size_m))


# ---- CoreSLAM -------------------------------------------------------------------------------
def pose_to_pxcs(pose, scale):
    pose = _f32(pose); out = np.empty(4, np.float32)
    lib().oracle_cs_pose_to_pxcs(_p(pose, C.c_float), C.c_float(scale), _p(out, C.c_float))
    return out


def poses_to_pxcs(poses, scale):
    poses = _f32(poses).reshape(-1, 3)
    out = np.empty((poses.shape[0], 4), np.float32)
    L = lib()
    for k in range(poses.shape[0]):
        L.oracle_cs_pose_to_pxcs(_p(poses[k], C.c_float), C.c_float(scale), _p(out[k], C.c_float))
    return out


def distance(pixels, size, scale, xy, pose):
    xy = _f32(xy); pose = _f32(pose)
    return lib().oracle_cs_distance(_p(pixels, C.c_uint16), size, C.c_float(scale), _p(xy, C.c_float),
                                    xy.shape[0], _p(pose, C.c_float))


def distance_batch_pxcs(pixels, size, xy, pxcs):
    xy = _f32(xy); pxcs = _f32(pxcs).reshape(-1, 4)
    K = pxcs.shape[0]
    out = np.empty(K, np.int32); bd = C.c_int32()
    bi = lib().oracle_cs_distance_batch_pxcs(_p(pixels, C.c_uint16), size, _p(xy, C.c_float), xy.shape[0],
                                             _p(pxcs, C.c_float), K, _p(out, C.c_int32), C.byref(bd))
    return out, bi, bd.value


def search(pixels, size, scale, xy, search_pose, offs):
    """Returns (best_index, best_pose, best_dist, all_dist[K]) with K = len(offs)+1, index 0 = base."""
    xy = _f32(xy); sp = _f32(search_pose); offs = _f32(offs).reshape(-1, 3)
    n = offs.shape[0]
    all_d = np.empty(n + 1, np.int32); pose = np.empty(3, np.float32); bd = C.c_int32()
    bi = lib().oracle_cs_search(_p(pixels, C.c_uint16), size, C.c_float(scale), _p(xy, C.c_float), xy.shape[0],
                                _p(sp, C.c_float), _p(offs, C.c_float), n, _p(pose, C.c_float), C.byref(bd),
                                _p(all_d, C.c_int32))
    return bi, pose, bd.value, all_d


def clip_ray(size, xyc, yxc, xy, yx):
    a, b = C.c_int32(xyc), C.c_int32(yxc)
    ok = lib().oracle_cs_clip_ray(size, C.byref(a), C.byref(b), xy, yx)
    return bool(ok), a.value, b.value


def draw_ray_holemap(pixels, size, x1, y1, x2, y2, xp, yp, value, alpha):
    return lib().oracle_cs_draw_ray_holemap(_p(pixels, C.c_uint16), size, x1, y1, x2, y2, xp, yp, value, alpha)


def update_holemap(pixels, size, scale, xy, pose, hole_width=0.6, quality=50):
    xy = _f32(xy); pose = _f32(pose)
    return lib().oracle_cs_update_holemap(_p(pixels, C.c_uint16), size, C.c_float(scale), _p(xy, C.c_float),
                                          xy.shape[0], _p(pose, C.c_float), C.c_float(hole_width), int(quality))


def update_holemap_pxcs(pixels, size, scale, xy, pxcs, hole_width=0.6, quality=50):
    xy = _f32(xy); pxcs = _f32(pxcs)
    return lib().oracle_cs_update_holemap_pxcs(_p(pixels, C.c_uint16), size, C.c_float(scale), _p(xy, C.c_float),
                                               xy.shape[0], _p(pxcs, C.c_float), C.c_float(hole_width), int(quality))


def update_obstaclemap(pixels, size, scale, xy, pose, max_hits=10):
    xy = _f32(xy); pose = _f32(pose)
    nohit = np.zeros(size * size, np.uint8)
    lib().oracle_cs_update_obstaclemap(_p(pixels, C.c_int8), _p(nohit, C.c_uint8), size, C.c_float(scale),
                                       _p(xy, C.c_float), xy.shape[0], _p(pose, C.c_float), int(max_hits))


def update_obstaclemap_pxcs(pixels, size, xy, pxcs, max_hits=10):
    xy = _f32(xy); pxcs = _f32(pxcs)
    nohit = np.zeros(size * size, np.uint8)
    lib().oracle_cs_update_obstaclemap_pxcs(_p(pixels, C.c_int8), _p(nohit, C.c_uint8), size,
                                            _p(xy, C.c_float), xy.shape[0], _p(pxcs, C.c_float), int(max_hits))


def segments_to_cloud(seg_poses, seg_start, rays, odo_pose):
    seg_poses = _f32(seg_poses).reshape(-1, 3); rays = _f32(rays).reshape(-1, 2)
    seg_start = np.ascontiguousarray(seg_start, np.int32); odo = _f32(odo_pose)
    out = np.empty((rays.shape[0], 2), np.float32)
    lib().oracle_cs_segments_to_cloud(_p(seg_poses, C.c_float), _p(seg_start, C.c_int32), seg_poses.shape[0],
                                      _p(rays, C.c_float), _p(odo, C.c_float), _p(out, C.c_float))
    return out


def pack_holemap(pixels):
    out = np.empty(pixels.size // 2, np.uint8)
    lib().oracle_cs_pack_holemap(_p(pixels, C.c_uint16), pixels.size, _p(out, C.c_uint8))
    return out


class CSProc:
    """oracle_csproc: CoreSLAMProcessor ctor/Reset/Update with explicit offsets."""

    def __init__(self, physical, hole_size, obst_size, start_pose):
        sp = _f32(start_pose)
        self.hole_size, self.obst_size = hole_size, obst_size
        self._h = lib().oracle_csproc_create(C.c_float(physical), hole_size, obst_size, _p(sp, C.c_float))

    def close(self):
        if self._h:
            lib().oracle_csproc_destroy(self._h); self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def reset(self):
        lib().oracle_csproc_reset(self._h)

    def set_params(self, quality=50, hole_width=0.6, search_beginning=5, unmapped_hits=-5, max_hits=10):
        lib().oracle_csproc_set_params(self._h, quality, C.c_float(hole_width), search_beginning, unmapped_hits, max_hits)

    def update(self, seg_poses, seg_start, rays, offs=None):
        seg_poses = _f32(seg_poses).reshape(-1, 3); rays = _f32(rays).reshape(-1, 2)
        seg_start = np.ascontiguousarray(seg_start, np.int32)
        if offs is None or len(offs) == 0:
            op, n = None, 0
        else:
            offs = _f32(offs).reshape(-1, 3); op, n = _p(offs, C.c_float), offs.shape[0]
        lib().oracle_csproc_update(self._h, _p(seg_poses, C.c_float), _p(seg_start, C.c_int32), seg_poses.shape[0],
                                   _p(rays, C.c_float), op, n)

    @property
    def pose(self):
        out = np.empty(3, np.float32)
        lib().oracle_csproc_get_pose(self._h, _p(out, C.c_float))
        return out

    @property
    def holemap(self):
        ptr = lib().oracle_csproc_holemap(self._h)
        return np.ctypeslib.as_array(ptr, shape=(self.hole_size * self.hole_size,))

    @property
    def obstaclemap(self):
        ptr = lib().oracle_csproc_obstaclemap(self._h)
        return np.ctypeslib.as_array(ptr, shape=(self.obst_size, self.obst_size))


# ---- Hector ---------------------------------------------------------------------------------
class Grid:
    def __init__(self, cell_len, w, h, off=(0.0, 0.0)):
        self.w, self.h, self.cell_len = w, h, float(np.float32(cell_len))
        self._h = lib().oracle_grid_create(C.c_float(cell_len), w, h, C.c_float(off[0]), C.c_float(off[1]))
        if not self._h:
            raise RuntimeError("Map to world matrix is not invertible")

    def close(self):
        if self._h:
            lib().oracle_grid_destroy(self._h); self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def reset(self):
        lib().oracle_grid_reset(self._h)

    def set_factors(self, free_f, occ_f):
        lib().oracle_grid_set_factors(self._h, C.c_float(free_f), C.c_float(occ_f))

    @property
    def logodds(self):
        a, b = C.c_float(), C.c_float()
        lib().oracle_grid_get_logodds(self._h, C.byref(a), C.byref(b))
        return a.value, b.value

    @property
    def cells(self):
        ptr = lib().oracle_grid_cells(self._h)
        buf = (C.c_char * (self.w * self.h * 8)).from_address(ptr)
        return np.frombuffer(buf, dtype=CELL_DTYPE)

    def prob(self, index):
        return lib().oracle_grid_prob(self._h, int(index))

    def prob_literal(self, index):
        """OccGridMap.cs:97-107 with the C# cache restated literally (deviation D5: stale across Reset)"""
        return lib().oracle_grid_prob_literal(self._h, int(index))

    def map_pose(self, world):
        w = _f32(world); o = np.empty(3, np.float32)
        lib().oracle_grid_map_pose(self._h, _p(w, C.c_float), _p(o, C.c_float)); return o

    def world_pose(self, mp):
        m = _f32(mp); o = np.empty(3, np.float32)
        lib().oracle_grid_world_pose(self._h, _p(m, C.c_float), _p(o, C.c_float)); return o

    def update_by_scan(self, xy, pose, origin=(0.0, 0.0)):
        xy = _f32(xy); pose = _f32(pose); org = _f32(origin)
        lib().oracle_grid_update_by_scan(self._h, _p(xy, C.c_float), xy.shape[0], _p(org, C.c_float), _p(pose, C.c_float))

    def bitmap(self):
        out = np.empty(self.w * self.h, np.uint8)
        lib().oracle_grid_bitmap(self._h, _p(out, C.c_uint8)); return out

    def map_extends(self):
        """(found, xMax, yMax, xMin, yMin) -- GridMap.cs:147-207"""
        o = (C.c_int * 4)()
        f = lib().oracle_grid_map_extends(self._h, o)
        return (bool(f), o[0], o[1], o[2], o[3])

    def interp(self, cx, cy):
        o = np.empty(3, np.float32)
        lib().oracle_hs_interp(self._h, C.c_float(cx), C.c_float(cy), _p(o, C.c_float)); return o

    def hessian(self, xy, pose_map, n_threads=1):
        xy = _f32(xy); p = _f32(pose_map); H = np.empty(9, np.float32); d = np.empty(3, np.float32)
        lib().oracle_hs_hessian(self._h, _p(xy, C.c_float), xy.shape[0], _p(p, C.c_float), n_threads,
                                _p(H, C.c_float), _p(d, C.c_float))
        return H.reshape(3, 3), d

    def estimate_step(self, xy, estimate, n_threads=1):
        xy = _f32(xy); e = _f32(estimate).copy()
        ok = lib().oracle_hs_estimate_step(self._h, _p(xy, C.c_float), xy.shape[0], _p(e, C.c_float), n_threads)
        return bool(ok), e

    def match(self, xy, hint_world, iterations=3, n_threads=1):
        xy = _f32(xy); h = _f32(hint_world); o = np.empty(3, np.float32)
        lib().oracle_hs_match_grid(self._h, _p(xy, C.c_float), xy.shape[0], _p(h, C.c_float), iterations, n_threads,
                                   _p(o, C.c_float))
        return o


def match_pyramid(levels, xy, hint_world, iterations, n_threads=1):
    xy = _f32(xy); h = _f32(hint_world); o = np.empty(3, np.float32)
    arr = (C.c_void_p * len(levels))(*[g._h for g in levels])
    its = np.ascontiguousarray(iterations, np.int32)
    lib().oracle_hs_match_pyramid(arr, len(levels), _p(xy, C.c_float), xy.shape[0], _p(h, C.c_float),
                                  _p(its, C.c_int32), n_threads, _p(o, C.c_float))
    return o


def make_pyramid(cell_len, w, h, levels):
    """MapRepMultiMap.cs:40-58: level i has size/2^i (integer division) and resolution*2^i."""
    out = []
    res = np.float32(cell_len)
    for _ in range(levels):
        out.append(Grid(float(res), w, h))
        w //= 2; h //= 2
        res = np.float32(res * np.float32(2.0))
    return out


# ---- CPU baseline ---------------------------------------------------------------------------
def cpu_baseline_search(pixels, size, scale, xy, search_pose, offs, n_threads, iters, n_scans):
    xy = _f32(xy); sp = _f32(search_pose); offs = _f32(offs).reshape(-1, 3)
    assert offs.shape[0] >= n_threads * iters
    ev, bi, bd = C.c_int64(), C.c_int32(), C.c_int32()
    secs = lib().oracle_cpu_baseline_search(_p(pixels, C.c_uint16), size, C.c_float(scale), _p(xy, C.c_float),
                                            xy.shape[0], _p(sp, C.c_float), _p(offs, C.c_float),
                                            n_threads, iters, n_scans, C.byref(ev), C.byref(bi), C.byref(bd))
    return secs, ev.value, bi.value, bd.value


def cpu_baseline_search_timed(pixels, size, scale, xy, search_pose, offs, n_threads, iters, n_scans):
    """As cpu_baseline_search, plus the wall time of every scan (float64[n_scans])."""
    xy = _f32(xy); sp = _f32(search_pose); offs = _f32(offs).reshape(-1, 3)
    assert offs.shape[0] >= n_threads * iters
    ev, bi, bd = C.c_int64(), C.c_int32(), C.c_int32()
    per = np.zeros(n_scans, np.float64)
    secs = lib().oracle_cpu_baseline_search_timed(_p(pixels, C.c_uint16), size, C.c_float(scale), _p(xy, C.c_float),
                                                  xy.shape[0], _p(sp, C.c_float), _p(offs, C.c_float),
                                                  n_threads, iters, n_scans, C.byref(ev), C.byref(bi), C.byref(bd),
                                                  per.ctypes.data_as(C.POINTER(C.c_double)))
    return secs, ev.value, bi.value, bd.value, per
