"""Multi-GPU plumbing for the sharded Monte-Carlo search: one process per GPU over torch.distributed
(backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in CPU tests).

The reference's only cross-worker step is the serial arg-min over the ParallelWorker threads' results with a
strict '<' (CoreSLAM/CoreSLAMProcessor.cs:695-705).  Here every rank evaluates a contiguous block of the flat
candidate list and contributes ONE packed 64-bit key (distance << 32 | flat index); min over the keys
reproduces the reference winner and its tie-break (lowest flat index = earliest thread / earliest draw).
Distances are non-negative int32, so the key fits a signed int64 and MIN on int64 tensors is exact.
"""
import ctypes as C

import numpy as np
import torch
import torch.distributed as dist

from . import capi

INT32_MAX = 2 ** 31 - 1


def shard_range(rank, world, k_total):
    """Contiguous block [first, first+count) of the flat candidate list owned by `rank`."""
    first = k_total * rank // world
    return first, k_total * (rank + 1) // world - first


def pack_key(distance, index):
    return (int(distance) << 32) | int(index)


def unpack_key(key):
    key = int(key)
    return key >> 32, key & 0xFFFFFFFF


def allreduce_min_key(key_tensor):
    """In-place min all-reduce of a 1-element int64 key tensor (8-byte message: latency bound)."""
    assert key_tensor.dtype == torch.int64 and key_tensor.numel() == 1
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(key_tensor, op=dist.ReduceOp.MIN)
    return key_tensor


def replicas_equal(checksums):
    """Replica check over torch.distributed (the map updates run as replicas on every rank): `checksums` =
    this rank's words (e.g. CoreSlamDevice.maps_checksum()); True when every rank holds the same ones.  uint64 words travel
    as two int32 halves each (gloo / RCCL reduce signed types), compared through a MIN and a MAX all-reduce.
    A rank that could not produce its words passes None: it still joins both collectives (the others are in them) with words
    that cannot match, and every rank gets False."""
    failed = checksums is None
    w = np.asarray([0, 2 ** 64 - 1] if failed else list(checksums), dtype=np.uint64)
    halves = torch.from_numpy(w.view(np.int32).astype(np.int64))
    if failed:
        halves = torch.tensor([-2 ** 31, 2 ** 31 - 1, -2 ** 31, 2 ** 31 - 1], dtype=torch.int64)[:halves.numel()] if halves.numel() <= 4 else halves
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")
        lo, hi = halves.to(dev), halves.to(dev).clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        return bool((lo == hi).all().item()) and not failed
    return not failed


class LibComm:
    """This rank's end of the library's own RCCL communicator (slamhip_comm_*): torch.distributed only carries the
    128-byte RCCL id from rank 0 to the others; every search step is then ONE C call that enqueues K1 on the operator's
    stream and the 8-byte min all-reduce on the communicator's stream (the next search does not wait for it)."""

    def __init__(self, ctx, rank, world):
        # ncclCommInitRank is collective: a rank that cannot get there (no librccl, no id) would leave the others blocked in
        # it for good, so every rank reports first whether it can, and nobody goes on unless all can
        uid = np.zeros(128, np.uint8)
        ok = 1
        try:
            capi.call("slamhip_comm_probe")
            if rank == 0:
                capi.call("slamhip_comm_unique_id", uid.ctypes.data_as(C.POINTER(C.c_uint8)))
        except Exception:                                          # noqa: BLE001
            ok = 0
        if world > 1:
            okt = torch.tensor([ok], dtype=torch.int32)
            if dist.get_backend() == "nccl":
                okt = okt.cuda()
            dist.all_reduce(okt, op=dist.ReduceOp.MIN)
            ok = int(okt.item())
        if not ok:
            raise RuntimeError("librccl cannot be resolved on every rank")
        if world > 1:
            t = torch.from_numpy(uid).cuda() if dist.get_backend() == "nccl" else torch.from_numpy(uid)
            dist.broadcast(t, 0)
            uid = np.ascontiguousarray(t.cpu().numpy())
        self._h = C.c_void_p()
        capi.call("slamhip_comm_create", ctx._h, uid.ctypes.data_as(C.POINTER(C.c_uint8)), int(rank), int(world), C.byref(self._h))
        self._step = capi.lib().slamhip_cs_search_allreduce_async

    def bind_step(self, dev, pose, first, count):
        """The per-step call with its arguments bound once: returns a zero-argument function."""
        self._pose = np.ascontiguousarray(pose, np.float32)        # (the bound pointer refers to this array)
        args = (dev._h, self._h, capi.fptr(self._pose), int(first), int(count), None)
        fn = self._step

        def step():
            capi.check(fn(*args))
        return step

    def search_allreduce(self, dev, pose, first, count):
        """One sharded search step, blocking (the per-scan form): K1, the 8-byte min all-reduce and the hand-over of the reduced
        key on the operator's stream; returns the reduced key."""
        k = C.c_uint64()
        capi.call("slamhip_cs_search_allreduce", dev._h, self._h, capi.fptr(np.ascontiguousarray(pose, np.float32)), int(first), int(count), C.byref(k))
        return int(k.value)

    def bind_search_allreduce(self, dev, pose, first, count):
        """The blocking step with its arguments bound once: a zero-argument function returning the reduced key."""
        self._pose_b = np.ascontiguousarray(pose, np.float32)
        k = C.c_uint64()
        args = (dev._h, self._h, capi.fptr(self._pose_b), int(first), int(count), C.byref(k))
        fn = capi.lib().slamhip_cs_search_allreduce

        def step():
            capi.check(fn(*args))
            return int(k.value)
        return step

    def search_allreduce_and_update(self, dev, pose, first, count, hole_width=0.6, quality=50, max_hits=10):
        """One scan on every rank: sharded search, RCCL min, winner decoded on the device, both map updates queued behind it
        (slamhip_cs_search_allreduce_and_update); returns (pose with theta normalised, distance, flat index)."""
        out = np.empty(3, np.float32); d, i = C.c_int32(), C.c_int32()
        capi.call("slamhip_cs_search_allreduce_and_update", dev._h, self._h, capi.fptr(np.ascontiguousarray(pose, np.float32)), int(first), int(count),
                  C.c_float(hole_width), int(quality), int(max_hits), capi.fptr(out), C.byref(d), C.byref(i))
        return out, int(d.value), int(i.value)

    def set_batch(self, steps):
        capi.call("slamhip_comm_set_batch", self._h, int(steps))

    def allreduce_probe(self, iters=200):
        """Device microseconds per 8-byte min all-reduce (no search in front); every rank calls it."""
        us = C.c_float()
        capi.call("slamhip_comm_allreduce_probe", self._h, int(iters), C.byref(us))
        return float(us.value)

    def replicas_equal(self, dev):
        """slamhip_comm_replicas_equal: the ranks' HoleMap / ObstacleMap checksums agree (every rank calls it)."""
        eq = C.c_int32()
        capi.call("slamhip_comm_replicas_equal", dev._h, self._h, C.byref(eq))
        return bool(eq.value)

    def info(self):
        r, n = C.c_int32(), C.c_int32()
        capi.call("slamhip_comm_info", self._h, C.byref(r), C.byref(n))
        return int(r.value), int(n.value)

    def wait(self):
        """Waits for every step issued so far; returns the reduced key of the last one."""
        k = C.c_uint64()
        capi.call("slamhip_comm_wait", self._h, C.byref(k))
        return int(k.value)

    def synchronize(self):
        capi.call("slamhip_comm_wait", self._h, None)

    def close(self):
        if self._h:
            capi.lib().slamhip_comm_destroy(self._h)
            self._h = C.c_void_p()
