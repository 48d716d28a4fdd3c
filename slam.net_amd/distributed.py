"""Multi-GPU plumbing for the sharded Monte-Carlo search: one process per GPU over torch.distributed
(backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in CPU tests).

The reference's only cross-worker step is the serial arg-min over the ParallelWorker threads' results with a
strict '<' (CoreSLAM/CoreSLAMProcessor.cs:695-705).  Here every rank evaluates a contiguous block of the flat
candidate list and contributes ONE packed 64-bit key (distance << 32 | flat index); min over the keys
reproduces the reference winner and its tie-break (lowest flat index = earliest thread / earliest draw).
Distances are non-negative int32, so the key fits a signed int64 and MIN on int64 tensors is exact.
"""
import torch
import torch.distributed as dist

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
