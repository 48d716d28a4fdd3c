"""Host mirror of the reference's CoreSLAM public API on top of the C-ABI (include/slamhip.h).

Same names, argument meaning and behaviour as CoreSLAM/CoreSLAMProcessor.cs, HoleMap.cs and
ObstacleMap.cs, so parity tests read like tests of the reference; every call goes through
libslamhip.so (HIP kernels) -- nothing is computed in Python.
"""
import ctypes as C

import numpy as np

from . import capi


class Context:
    """One GPU + one HIP stream (slamhip_ctx); stands where `new ParallelWorker(n)` stood."""

    def __init__(self, device=0):
        self._h = C.c_void_p()
        capi.call("slamhip_ctx_create", int(device), C.byref(self._h))

    def close(self):
        if self._h:
            capi.lib().slamhip_ctx_destroy(self._h)
            self._h = C.c_void_p()

    @property
    def stream(self):
        return capi.lib().slamhip_ctx_stream(self._h)

    def synchronize(self):
        capi.call("slamhip_ctx_synchronize", self._h)

    def set_wait_timeout(self, timeout_ms):
        """Bound on every blocking wait of the context (slamhip_ctx_set_wait_timeout; <= 0: none)."""
        capi.call("slamhip_ctx_set_wait_timeout", self._h, int(timeout_ms))

    @property
    def poisoned(self):
        v = C.c_int32()
        capi.call("slamhip_ctx_poisoned", self._h, C.byref(v))
        return bool(v.value)

    def philox4x32_10(self, counter, key):
        """One Philox4x32-10 block on the device (slamhip_ctx_philox4x32_10): the candidate generator's integer stream, for known-answer tests."""
        c = (C.c_uint32 * 4)(*[int(x) & 0xFFFFFFFF for x in counter])
        k = (C.c_uint32 * 2)(*[int(x) & 0xFFFFFFFF for x in key])
        o = (C.c_uint32 * 4)()
        capi.call("slamhip_ctx_philox4x32_10", self._h, c, k, o)
        return tuple(int(x) for x in o)

    def timing_enable(self, mask=-1):
        """mask: bit k enables kernel class k (capi.K_*); 0 = off; -1 = all."""
        capi.call("slamhip_ctx_timing_enable", self._h, int(mask))

    def timing_reset(self):
        capi.call("slamhip_ctx_timing_reset", self._h)

    def timing_get(self, which):
        ms, n = C.c_double(), C.c_int64()
        capi.call("slamhip_ctx_timing_get", self._h, int(which), C.byref(ms), C.byref(n))
        return ms.value, n.value


class Ray:
    """BaseSLAM/Ray.cs:10-32"""

    def __init__(self, angle, radius):
        self.Angle = float(angle)
        self.Radius = float(radius)


class ScanSegment:
    """BaseSLAM/ScanSegment.cs:13-29"""

    def __init__(self, rays, pose, is_last=True):
        self.Rays = rays            # list of Ray, or (n,2) float32 array of (angle, radius)
        self.Pose = np.asarray(pose, np.float32)
        self.IsLast = is_last


def _segments_to_arrays(segments):
    poses, start, rays = [], [0], []
    for s in segments:
        poses.append(np.asarray(s.Pose, np.float32))
        r = s.Rays
        if not isinstance(r, np.ndarray):
            r = np.array([[x.Angle, x.Radius] for x in r], np.float32).reshape(-1, 2)
        rays.append(np.asarray(r, np.float32).reshape(-1, 2))
        start.append(start[-1] + rays[-1].shape[0])
    return (np.ascontiguousarray(np.stack(poses), np.float32), np.ascontiguousarray(start, np.int32),
            np.ascontiguousarray(np.concatenate(rays) if rays else np.zeros((0, 2)), np.float32))


class CoreSlamDevice:
    """Operator-level object (slamhip_cs): device HoleMap + ObstacleMap + scan + candidate list."""

    def __init__(self, ctx, physical_map_size, hole_map_size, obstacle_map_size, _handle=None):
        self.ctx = ctx
        self._owned = _handle is None
        self._h = C.c_void_p() if _handle is None else _handle
        if _handle is None:
            capi.call("slamhip_cs_create", ctx._h, C.c_float(physical_map_size), int(hole_map_size),
                      int(obstacle_map_size), C.byref(self._h))
        hs, hsc, os_, osc = C.c_int32(), C.c_float(), C.c_int32(), C.c_float()
        capi.call("slamhip_cs_info", self._h, C.byref(hs), C.byref(hsc), C.byref(os_), C.byref(osc))
        self.hole_size, self.hole_scale, self.obst_size, self.obst_scale = hs.value, hsc.value, os_.value, osc.value
        self.n_offsets = 0

    def close(self):
        if self._h and self._owned:
            capi.lib().slamhip_cs_destroy(self._h)
        self._h = C.c_void_p()

    # -- maps
    def reset(self, unmapped_obstacle_hits=-5):
        capi.call("slamhip_cs_reset", self._h, int(unmapped_obstacle_hits))

    def holemap_upload(self, pixels):
        p = np.ascontiguousarray(pixels, np.uint16).reshape(-1)
        capi.call("slamhip_cs_holemap_upload", self._h, p.ctypes.data_as(C.POINTER(C.c_uint16)), p.size)

    def holemap_download(self):
        out = np.empty(self.hole_size * self.hole_size, np.uint16)
        capi.call("slamhip_cs_holemap_download", self._h, out.ctypes.data_as(C.POINTER(C.c_uint16)), out.size)
        return out

    def holemap_download_packed(self):
        out = np.empty(self.hole_size * self.hole_size // 2, np.uint8)
        capi.call("slamhip_cs_holemap_download_packed", self._h, out.ctypes.data_as(C.POINTER(C.c_uint8)), out.size)
        return out

    def holemap_mirror(self, pixels):
        """Brings `pixels` (the uint16[Size*Size] array the previous mirror call filled) up to date in place by copying only
        the rectangle the updates since then touched; returns (x0, y0, x1, y1) inclusive, or (0, 0, -1, -1)."""
        assert pixels.dtype == np.uint16 and pixels.size == self.hole_size * self.hole_size and pixels.flags.c_contiguous
        rect = np.zeros(4, np.int32)
        capi.call("slamhip_cs_holemap_mirror", self._h, pixels.ctypes.data_as(C.POINTER(C.c_uint16)), pixels.size, capi.iptr(rect))
        return tuple(int(v) for v in rect)

    def holemap_mirror_async(self, pixels):
        """Asynchronous, span-exact refresh of the host mirror `pixels` (slamhip_cs_holemap_mirror_async): returns at once; the
        array is page-locked on first use, must stay alive until holemap_mirror_release / close, and must not be read before
        holemap_mirror_wait."""
        assert pixels.dtype == np.uint16 and pixels.size == self.hole_size * self.hole_size and pixels.flags.c_contiguous
        self._mirror_keepalive = pixels
        capi.call("slamhip_cs_holemap_mirror_async", self._h, pixels.ctypes.data_as(C.POINTER(C.c_uint16)), pixels.size)

    def holemap_mirror_wait(self):
        """Waits for the last asynchronous mirror push; returns ((x0, y0, x1, y1), pixels pushed)."""
        rect = np.zeros(4, np.int32); px = C.c_int64()
        capi.call("slamhip_cs_holemap_mirror_wait", self._h, capi.iptr(rect), C.byref(px))
        return tuple(int(v) for v in rect), int(px.value)

    def holemap_mirror_release(self):
        capi.call("slamhip_cs_holemap_mirror_release", self._h)
        self._mirror_keepalive = None

    def obstaclemap_upload(self, pixels):
        p = np.ascontiguousarray(pixels, np.int8).reshape(-1)
        capi.call("slamhip_cs_obstaclemap_upload", self._h, p.ctypes.data_as(C.POINTER(C.c_int8)), p.size)

    def obstaclemap_download(self):
        out = np.empty((self.obst_size, self.obst_size), np.int8)
        capi.call("slamhip_cs_obstaclemap_download", self._h, out.ctypes.data_as(C.POINTER(C.c_int8)), out.size)
        return out

    # -- scan / distance / search
    def set_scan(self, xy):
        xy = capi.f32(xy, (-1, 2))
        capi.call("slamhip_cs_set_scan", self._h, capi.fptr(xy), xy.shape[0])

    def _distance(self, fn, arr, width, want_all):
        arr = capi.f32(arr, (-1, width))
        K = arr.shape[0]
        out = np.empty(K, np.int32) if want_all else None
        bi, bd = C.c_int32(), C.c_int32()
        capi.call(fn, self._h, capi.fptr(arr), K, capi.iptr(out) if want_all else None, C.byref(bi), C.byref(bd))
        return out, bi.value, bd.value

    def distance_pxcs(self, pxcs, want_all=True):
        return self._distance("slamhip_cs_distance_pxcs", pxcs, 4, want_all)

    def distance_poses(self, poses, want_all=True):
        return self._distance("slamhip_cs_distance_poses", poses, 3, want_all)

    def set_offsets(self, offs):
        offs = capi.f32(offs, (-1, 3))
        capi.call("slamhip_cs_set_offsets", self._h, capi.fptr(offs), offs.shape[0])
        self.n_offsets = offs.shape[0]

    def generate_offsets(self, n, sigma_xy, sigma_theta, seed=0, stream=0, lattice=False):
        """Device-generated jitters; lattice=True: the heading lattice (slamhip_cs_generate_offsets_lattice)."""
        capi.call("slamhip_cs_generate_offsets_lattice" if lattice else "slamhip_cs_generate_offsets", self._h, int(n),
                  C.c_float(sigma_xy), C.c_float(sigma_theta), C.c_uint64(seed), C.c_uint64(stream))
        self.n_offsets = int(n)

    def offsets_download(self, n=None):
        n = self.n_offsets if n is None else int(n)
        out = np.empty((n, 3), np.float32)
        capi.call("slamhip_cs_offsets_download", self._h, capi.fptr(out), n)
        return out

    def search(self, search_pose):
        sp = capi.f32(search_pose)
        pose = np.empty(3, np.float32); d, i = C.c_int32(), C.c_int32()
        capi.call("slamhip_cs_search", self._h, capi.fptr(sp), capi.fptr(pose), C.byref(d), C.byref(i))
        return pose, d.value, i.value

    def search_shard(self, search_pose, first, count):
        sp = capi.f32(search_pose); key = C.c_uint64()
        capi.call("slamhip_cs_search_shard", self._h, capi.fptr(sp), int(first), int(count), C.byref(key))
        return key.value

    def search_shard_async(self, search_pose, first, count, device_ptr):
        sp = capi.f32(search_pose)
        capi.call("slamhip_cs_search_shard_async", self._h, capi.fptr(sp), int(first), int(count), C.c_void_p(device_ptr))

    def search_shard_enqueue(self, search_pose, first, count):
        """Enqueue-only search into the handle's result ring; returns the device address of the key (slamhip_cs_search_shard_enqueue)."""
        sp = capi.f32(search_pose)
        d = C.c_void_p()
        capi.call("slamhip_cs_search_shard_enqueue", self._h, capi.fptr(sp), int(first), int(count), C.byref(d))
        return d.value

    def key_read(self, device_ptr):
        k = C.c_uint64()
        capi.call("slamhip_cs_key_read", self._h, C.c_void_p(device_ptr), C.byref(k))
        return k.value

    def pose_from_key(self, search_pose, key):
        sp = capi.f32(search_pose)
        pose = np.empty(3, np.float32); d, i = C.c_int32(), C.c_int32()
        capi.call("slamhip_cs_pose_from_key", self._h, capi.fptr(sp), C.c_uint64(key), capi.fptr(pose), C.byref(d), C.byref(i))
        return pose, d.value, i.value

    # -- map updates
    def update_holemap(self, pose, hole_width=0.6, quality=50):
        p = capi.f32(pose)
        capi.call("slamhip_cs_update_holemap", self._h, capi.fptr(p), C.c_float(hole_width), int(quality))

    def update_holemap_pxcs(self, pxcs, hole_width=0.6, quality=50):
        p = capi.f32(pxcs)
        capi.call("slamhip_cs_update_holemap_pxcs", self._h, capi.fptr(p), C.c_float(hole_width), int(quality))

    def update_obstaclemap(self, pose, max_hits=10):
        p = capi.f32(pose)
        capi.call("slamhip_cs_update_obstaclemap", self._h, capi.fptr(p), int(max_hits))

    def update_obstaclemap_pxcs(self, pxcs, max_hits=10):
        p = capi.f32(pxcs)
        capi.call("slamhip_cs_update_obstaclemap_pxcs", self._h, capi.fptr(p), int(max_hits))

    @property
    def last_holemap_pixels(self):
        v = C.c_int64()
        capi.call("slamhip_cs_last_holemap_pixels", self._h, C.byref(v))
        return v.value

    def maps_checksum(self):
        """(HoleMap, ObstacleMap) replica-check words (slamhip_cs_maps_checksum; definition in include/slamhip.h)."""
        out = (C.c_uint64 * 2)()
        capi.call("slamhip_cs_maps_checksum", self._h, out)
        return int(out[0]), int(out[1])

    def prepared_lists(self):
        """(served, prepared): candidate lists the per-scan flow found prepared ahead / prepared in all (slamhip_cs_prepared_lists)."""
        a, b = C.c_uint64(), C.c_uint64()
        capi.call("slamhip_cs_prepared_lists", self._h, C.byref(a), C.byref(b))
        return a.value, b.value

    @property
    def prelaunch_stats(self):
        """(searched ahead of the scan's tables, abandoned, layout remade first, refused): slamhip_cs_prelaunch_stats"""
        v = (C.c_uint64 * 4)()
        capi.call("slamhip_cs_prelaunch_stats", self._h, v)
        return tuple(int(x) for x in v)

    @property
    def plan_stats(self):
        """(searches launched with a plan, without, host waits for a plan slot, plans skipped: inputs in flight): slamhip_cs_plan_stats"""
        v = (C.c_uint64 * 4)()
        capi.call("slamhip_cs_plan_stats", self._h, v)
        return tuple(int(x) for x in v)

    @property
    def selfcheck_failures(self):
        v = C.c_uint32()
        capi.call("slamhip_cs_selfcheck_failures", self._h, C.byref(v))
        return v.value

    def search_and_update(self, search_pose, hole_width=0.6, quality=50, max_hits=10):
        sp = capi.f32(search_pose)
        pose = np.empty(3, np.float32); d, i = C.c_int32(), C.c_int32()
        capi.call("slamhip_cs_search_and_update", self._h, capi.fptr(sp), C.c_float(hole_width), int(quality),
                  int(max_hits), capi.fptr(pose), C.byref(d), C.byref(i))
        return pose, d.value, i.value


    def scan_search_and_update(self, xy, search_pose, hole_width=0.6, quality=50, max_hits=10):
        """set_scan + search_and_update in one call (slamhip_cs_scan_search_and_update): the search launch may precede the scan's tables"""
        pts = np.ascontiguousarray(xy, np.float32).reshape(-1, 2)
        sp = capi.f32(search_pose)
        pose = np.empty(3, np.float32); d, i = C.c_int32(), C.c_int32()
        capi.call("slamhip_cs_scan_search_and_update", self._h, capi.fptr(pts), int(pts.shape[0]), capi.fptr(sp), C.c_float(hole_width), int(quality),
                  int(max_hits), capi.fptr(pose), C.byref(d), C.byref(i))
        return pose, d.value, i.value

    def search_and_update_pxcs(self, pxcs_search, pxcs_update_hole, pxcs_update_obst=None, hole_width=0.6, quality=50, max_hits=10):
        """The fused scan with the caller's own (px, py, c, s) (slamhip_cs_search_and_update_pxcs): the candidates for the search,
        and their rows -- of the normalised pose -- at both map scales for the updates.  Returns (index, distance) of the first strict
        minimum; both maps are updated from row `index` of the update arrays."""
        ps = np.ascontiguousarray(pxcs_search, np.float32).reshape(-1, 4)
        ph = np.ascontiguousarray(pxcs_update_hole, np.float32).reshape(-1, 4)
        po = None if pxcs_update_obst is None else np.ascontiguousarray(pxcs_update_obst, np.float32).reshape(-1, 4)
        assert ph.shape == ps.shape and (po is None or po.shape == ps.shape)
        d, i = C.c_int32(), C.c_int32()
        capi.call("slamhip_cs_search_and_update_pxcs", self._h, capi.fptr(ps), capi.fptr(ph), capi.fptr(po) if po is not None else None, int(ps.shape[0]),
                  C.c_float(hole_width), int(quality), int(max_hits), C.byref(i), C.byref(d))
        return i.value, d.value

    def update_maps_pxcs(self, pxcs_hole, pxcs_obst=None, hole_width=0.6, quality=50, max_hits=10):
        ph = capi.f32(pxcs_hole); po = None if pxcs_obst is None else capi.f32(pxcs_obst)
        capi.call("slamhip_cs_update_maps_pxcs", self._h, capi.fptr(ph), capi.fptr(po) if po is not None else None, C.c_float(hole_width), int(quality), int(max_hits))


class HoleMap:
    """CoreSLAM/HoleMap.cs: Pixels / Size / Scale / GetPackedPixels(), backed by the device map."""

    def __init__(self, dev):
        self._dev = dev
        self.Size = dev.hole_size
        self.Scale = dev.hole_scale
        self._pixels = None

    @property
    def Pixels(self):
        """ushort[Size*Size]; refreshed from the device (the managed mirror of SURVEY.md sec.8b)."""
        self._pixels = self._dev.holemap_download()
        return self._pixels

    def GetPackedPixels(self):
        return self._dev.holemap_download_packed()


class ObstacleMap:
    """CoreSLAM/ObstacleMap.cs: Pixels[y, x] / Size / Scale."""

    def __init__(self, dev):
        self._dev = dev
        self.Size = dev.obst_size
        self.Scale = dev.obst_scale

    @property
    def Pixels(self):
        return self._dev.obstaclemap_download()


class CoreSLAMProcessor:
    """CoreSLAM/CoreSLAMProcessor.cs public surface: ctor :119-120, Update :717, Reset :167, Pose :106,
    HoleMap :45, ObstacleMap :50, Quality :80, HoleWidth :85, PositionSearchBeginning :90,
    UnmappedObstacleHits :96, MaxObstacleHits :101, Dispose :757."""

    def __init__(self, physicalMapSize, holeMapSize, obstacleMapSize, startPose, sigmaXY, sigmaTheta,
                 iterationsPerThread, numSearchThreads, ctx=None):
        self._own_ctx = ctx is None
        self.ctx = ctx or Context(0)
        sp = capi.f32(startPose)
        self._h = C.c_void_p()
        capi.call("slamhip_csproc_create", self.ctx._h, C.c_float(physicalMapSize), int(holeMapSize),
                  int(obstacleMapSize), capi.fptr(sp), C.c_float(sigmaXY), C.c_float(sigmaTheta),
                  int(iterationsPerThread), int(numSearchThreads), C.byref(self._h))
        csh = C.c_void_p()
        capi.call("slamhip_csproc_cs", self._h, C.byref(csh))
        self.device = CoreSlamDevice(self.ctx, physicalMapSize, holeMapSize, obstacleMapSize, _handle=csh)
        self.PhysicalMapSize = float(physicalMapSize)
        self.SigmaXY, self.SigmaTheta = float(sigmaXY), float(sigmaTheta)
        self.SearchIterationsPerThread, self.NumSearchThreads = int(iterationsPerThread), int(numSearchThreads)
        self.HoleMap = HoleMap(self.device)
        self.ObstacleMap = ObstacleMap(self.device)
        self._params = dict(Quality=50, HoleWidth=0.6, PositionSearchBeginning=5, UnmappedObstacleHits=-5,
                            MaxObstacleHits=10)

    def _push(self):
        p = self._params
        capi.call("slamhip_csproc_set_params", self._h, int(p["Quality"]), C.c_float(p["HoleWidth"]),
                  int(p["PositionSearchBeginning"]), int(p["UnmappedObstacleHits"]), int(p["MaxObstacleHits"]))

    def __getattr__(self, name):
        if name in ("Quality", "HoleWidth", "PositionSearchBeginning", "UnmappedObstacleHits", "MaxObstacleHits"):
            return self.__dict__["_params"][name]
        raise AttributeError(name)

    def __setattr__(self, name, value):
        if name in ("Quality", "HoleWidth", "PositionSearchBeginning", "UnmappedObstacleHits", "MaxObstacleHits"):
            self.__dict__["_params"][name] = value
            self._push()
        else:
            object.__setattr__(self, name, value)

    @property
    def Pose(self):
        out = np.empty(3, np.float32)
        capi.call("slamhip_csproc_get_pose", self._h, capi.fptr(out))
        return out

    def Reset(self):
        capi.call("slamhip_csproc_reset", self._h)

    def Update(self, segments):
        poses, start, rays = _segments_to_arrays(segments)
        capi.call("slamhip_csproc_update", self._h, capi.fptr(poses), capi.iptr(start), poses.shape[0], capi.fptr(rays))

    # extensions used by parity tests (the reference's sampler is entropy-seeded)
    def SetSeed(self, seed):
        capi.call("slamhip_csproc_set_seed", self._h, C.c_uint64(seed))

    def SetLattice(self, on):
        """Opt-in: the per-scan candidates as a heading lattice (slamhip_csproc_set_lattice)."""
        capi.call("slamhip_csproc_set_lattice", self._h, 1 if on else 0)

    def SetOffsets(self, offs):
        offs = capi.f32(offs, (-1, 3))
        capi.call("slamhip_csproc_set_offsets", self._h, capi.fptr(offs), offs.shape[0])
        self.device.n_offsets = offs.shape[0]

    def Dispose(self):
        if self._h:
            capi.lib().slamhip_csproc_destroy(self._h)
            self._h = C.c_void_p()
        if self._own_ctx:
            self.ctx.close()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.Dispose()
