"""Host mirror of the reference's HectorSLAM public API on top of the C-ABI (include/slamhip.h).

Mirrors HectorSLAM/Main/MapRepMultiMap.cs, Map/OccGridMap.cs, Matcher/ScanMatcher.cs and
Main/HectorSLAMProcessor.cs: same names and argument meaning; all compute is in libslamhip.so.
"""
import ctypes as C

import numpy as np

from . import capi
from .coreslam import Context


class ScanCloud:
    """BaseSLAM/ScanCloud.cs:10-21"""

    def __init__(self, points, pose=(0.0, 0.0, 0.0)):
        self.Points = capi.f32(points, (-1, 2))
        self.Pose = np.asarray(pose, np.float32)


class OccGridMap:
    """One pyramid level (HectorSLAM/Map/OccGridMap.cs + GridMap.cs), a view onto the device pyramid."""

    def __init__(self, rep, level):
        self._rep, self.level = rep, level
        w, h, c = C.c_int32(), C.c_int32(), C.c_float()
        capi.call("slamhip_hs_level_info", rep._h, level, C.byref(w), C.byref(h), C.byref(c))
        self.Dimensions = (w.value, h.value)
        self.CellLength = c.value
        self._iters = 3

    @property
    def EstimateIterations(self):
        return self._iters

    @EstimateIterations.setter
    def EstimateIterations(self, v):
        self._iters = int(v)
        self._rep._push_iterations()

    def GetCells(self):
        n = self.Dimensions[0] * self.Dimensions[1]
        out = np.empty(n, capi.CELL_DTYPE)
        capi.call("slamhip_hs_cells_download", self._rep._h, self.level, out.ctypes.data_as(C.c_void_p), n)
        return out

    def SetCells(self, cells):
        cells = np.ascontiguousarray(cells, capi.CELL_DTYPE)
        capi.call("slamhip_hs_cells_upload", self._rep._h, self.level, cells.ctypes.data_as(C.c_void_p), cells.size)

    def GetBitmapData(self):
        n = self.Dimensions[0] * self.Dimensions[1]
        out = np.empty(n, np.uint8)
        capi.call("slamhip_hs_bitmap_download", self._rep._h, self.level, out.ctypes.data_as(C.POINTER(C.c_uint8)), n)
        return out

    def GetMapExtends(self):
        """GridMap.GetMapExtends (GridMap.cs:147-207): (found, xMax, yMax, xMin, yMin), reduced on the device."""
        e = (C.c_int32 * 4)()
        f = C.c_int32()
        capi.call("slamhip_hs_map_extends", self._rep._h, self.level, e, C.byref(f))
        return (bool(f.value), e[0], e[1], e[2], e[3])

    def checksum(self):
        """(log-odds, update indices) replica-check words of this level (slamhip_hs_checksum)."""
        out = (C.c_uint64 * 2)()
        capi.call("slamhip_hs_checksum", self._rep._h, self.level, out)
        return int(out[0]), int(out[1])

    def GetCell(self, *a):
        """GridMap.GetCell(x, y) / GetCell(index) (GridMap.cs:70-96): one LogOddsCell read back from the device."""
        idx = a[1] * self.Dimensions[0] + a[0] if len(a) == 2 else int(a[0])
        return self.GetCells()[idx]

    def GetCachedProbability(self, indices):
        idx = np.ascontiguousarray(np.atleast_1d(indices), np.int32)
        out = np.empty(idx.size, np.float32)
        capi.call("slamhip_hs_probability", self._rep._h, self.level, capi.iptr(idx), idx.size, capi.fptr(out))
        return out

    def Hessian(self, pose_map):
        p = capi.f32(pose_map); H = np.empty(9, np.float32); d = np.empty(3, np.float32)
        capi.call("slamhip_hs_hessian", self._rep._h, self.level, capi.fptr(p), capi.fptr(H), capi.fptr(d))
        return H.reshape(3, 3), d


class MapRepMultiMap:
    """HectorSLAM/Main/MapRepMultiMap.cs:19-96"""

    def __init__(self, mapResolution, mapSize, numDepth, startCoords=(0.0, 0.0), ctx=None, _handle=None):
        self.ctx = ctx or Context(0)
        self._owned = _handle is None
        self._h = C.c_void_p() if _handle is None else _handle
        if _handle is None:
            capi.call("slamhip_hs_create", self.ctx._h, C.c_float(mapResolution), int(mapSize[0]), int(mapSize[1]),
                      int(numDepth), C.byref(self._h))
        self.Maps = [OccGridMap(self, l) for l in range(numDepth)]
        self._scan_set = None

    @property
    def NumLevels(self):
        return len(self.Maps)

    def _push_iterations(self):
        it = np.array([m._iters for m in self.Maps], np.int32)
        capi.call("slamhip_hs_set_iterations", self._h, capi.iptr(it))

    def Reset(self):
        capi.call("slamhip_hs_reset", self._h)

    def SetUpdateFactorFree(self, factor):
        self._free = float(factor)
        capi.call("slamhip_hs_set_factors", self._h, C.c_float(factor), C.c_float(getattr(self, "_occ", 0.9)))

    def SetUpdateFactorOccupied(self, factor):
        self._occ = float(factor)
        capi.call("slamhip_hs_set_factors", self._h, C.c_float(getattr(self, "_free", 0.4)), C.c_float(factor))

    def set_scan(self, scan):
        org = capi.f32(scan.Pose[:2])
        capi.call("slamhip_hs_set_scan", self._h, capi.fptr(scan.Points), scan.Points.shape[0], capi.fptr(org))

    def UpdateByScan(self, scan, pose):
        self.set_scan(scan)
        p = capi.f32(pose)
        capi.call("slamhip_hs_update_by_scan", self._h, capi.fptr(p))

    def close(self):
        if self._h and self._owned:
            capi.lib().slamhip_hs_destroy(self._h)
        self._h = C.c_void_p()


class ScanMatcher:
    """HectorSLAM/Matcher/ScanMatcher.cs:18-271.  numThreads is accepted for source compatibility; the
    point-chunk fan-out it controlled (:149-185) is the workgroup reduction of kernel K4."""

    def __init__(self, numThreads=1, logger=None):
        self.numThreads = numThreads

    def MatchData(self, target, scan, hintPose):
        hint = capi.f32(hintPose); out = np.empty(3, np.float32)
        if isinstance(target, MapRepMultiMap):                      # :41
            target.set_scan(scan)
            capi.call("slamhip_hs_match", target._h, capi.fptr(hint), capi.fptr(out))
        else:                                                       # :64 MatchData(OccGridMap, ...)
            target._rep.set_scan(scan)
            capi.call("slamhip_hs_match_level", target._rep._h, target.level, capi.fptr(hint),
                      target.EstimateIterations, capi.fptr(out))
        return out

    def MatchDataBatch(self, rep, scan, hintPoses):
        hints = capi.f32(hintPoses, (-1, 3)); out = np.empty_like(hints)
        rep.set_scan(scan)
        capi.call("slamhip_hs_match_batch", rep._h, capi.fptr(hints), hints.shape[0], capi.fptr(out))
        return out

    def Dispose(self):
        pass


class HectorSLAMProcessor:
    """HectorSLAM/Main/HectorSLAMProcessor.cs:17-160"""

    def __init__(self, mapResolution, mapSize, startPose, numDepth, numThreads=1, logger=None, ctx=None):
        self._own_ctx = ctx is None
        self.ctx = ctx or Context(0)
        sp = capi.f32(startPose)
        self._h = C.c_void_p()
        capi.call("slamhip_hsproc_create", self.ctx._h, C.c_float(mapResolution), int(mapSize[0]), int(mapSize[1]),
                  capi.fptr(sp), int(numDepth), C.byref(self._h))
        hsh = C.c_void_p()
        capi.call("slamhip_hsproc_hs", self._h, C.byref(hsh))
        self.MapRep = MapRepMultiMap(mapResolution, mapSize, numDepth, ctx=self.ctx, _handle=hsh)
        self._min_dist, self._min_angle = 0.3, 0.13

    def _get(self):
        m = np.empty(3, np.float32); l = np.empty(3, np.float32); mt, ut = C.c_float(), C.c_float()
        capi.call("slamhip_hsproc_get", self._h, capi.fptr(m), capi.fptr(l), C.byref(mt), C.byref(ut))
        return m, l, mt.value, ut.value

    MatchPose = property(lambda self: self._get()[0])
    LastMapUpdatePose = property(lambda self: self._get()[1])
    MatchTiming = property(lambda self: self._get()[2])
    UpdateTiming = property(lambda self: self._get()[3])

    @property
    def MinDistanceDiffForMapUpdate(self):
        return self._min_dist

    @MinDistanceDiffForMapUpdate.setter
    def MinDistanceDiffForMapUpdate(self, v):
        self._min_dist = float(v)
        capi.call("slamhip_hsproc_set_thresholds", self._h, C.c_float(self._min_dist), C.c_float(self._min_angle))

    @property
    def MinAngleDiffForMapUpdate(self):
        return self._min_angle

    @MinAngleDiffForMapUpdate.setter
    def MinAngleDiffForMapUpdate(self, v):
        self._min_angle = float(v)
        capi.call("slamhip_hsproc_set_thresholds", self._h, C.c_float(self._min_dist), C.c_float(self._min_angle))

    def Update(self, scan, poseHintWorld, mapWithoutMatching=False):
        hint = capi.f32(poseHintWorld); org = capi.f32(scan.Pose[:2]); upd = C.c_int32()
        capi.call("slamhip_hsproc_update", self._h, capi.fptr(scan.Points), scan.Points.shape[0], capi.fptr(org),
                  capi.fptr(hint), 1 if mapWithoutMatching else 0, C.byref(upd))
        return bool(upd.value)

    def Reset(self):
        capi.call("slamhip_hsproc_reset", self._h)

    def Dispose(self):
        if self._h:
            capi.lib().slamhip_hsproc_destroy(self._h)
            self._h = C.c_void_p()
        if self._own_ctx:
            self.ctx.close()
