"""world_size-2/3 gloo tests (CPU) of the sharded search: contiguous candidate shards + one min all-reduce of the
packed (distance << 32 | index) key must reproduce the full search, including the reference tie-break.
Distances come from the CPU oracle here (test infrastructure); on the GPU box the same plumbing carries the
keys produced by slamhip_cs_search_shard_async (tests/test_gpu_coreslam.py checks those against the oracle)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, K, tie, out_q):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle_c as oc
    import slam.net_amd.distributed as D
    import slam.net_amd.sim as sim
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    oc.set_trig_mode(oc.TRIG_DET)
    size = 128
    scale = oc.map_scale(size, 40.0)
    segs = sim.default_field()
    pix = np.full(size * size, 32750, np.uint16)
    rng = sim.PCG32(5)
    for p in sim.trajectory(4):
        _, xy = sim.make_scan(segs, p, 180, rng)
        oc.update_holemap(pix, size, scale, xy, p)
    _, xy = sim.make_scan(segs, sim.trajectory(5)[-1], 180, rng)
    if tie:
        pix[:] = 32750                                    # uniform map + a small scan: every candidate ties
        ang = np.arange(90) * (2 * np.pi / 90)
        xy = np.stack([2.0 * np.cos(ang), 2.0 * np.sin(ang)], 1).astype(np.float32)
    base = np.array([20.2, 20.1, 0.03], np.float32)
    offs = sim.gaussian_offsets(K - 1, seed=9)
    poses = np.vstack([base[None], base[None] + offs]).astype(np.float32)
    first, count = D.shard_range(rank, world, K)
    d, bi, bd = oc.distance_batch_pxcs(pix, size, xy, oc.poses_to_pxcs(poses[first:first + count], scale))
    key = torch.tensor([D.pack_key(bd, first + bi)], dtype=torch.int64)
    D.allreduce_min_key(key)
    full_d, full_bi, full_bd = oc.distance_batch_pxcs(pix, size, xy, oc.poses_to_pxcs(poses, scale))
    out_q.put((rank, first, count, int(key.item()), D.pack_key(full_bd, full_bi)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,K,tie", [(2, 1000, False), (2, 1001, True), (3, 257, False)])
def test_sharded_search_gloo(world, K, tie):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, K, tie, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    res.sort()
    covered = 0
    for rank, first, count, key, full in res:
        assert first == covered
        covered += count
        assert key == full, (rank, key, full)
    assert covered == K
    if tie:
        assert res[0][3] & 0xFFFFFFFF == 0                  # the un-jittered base pose (flat index 0) wins ties


def test_key_helpers():
    sys.path.insert(0, ROOT)
    import slam.net_amd.distributed as D
    assert D.unpack_key(D.pack_key(2 ** 31 - 1, 4294967295)) == (2 ** 31 - 1, 4294967295)
    assert D.pack_key(5, 7) < D.pack_key(5, 8) < D.pack_key(6, 0) < 2 ** 63
    tot = 0
    for r in range(8):
        f, c = D.shard_range(r, 8, 262144)
        assert f == tot and c == 32768
        tot += c
    assert [D.shard_range(r, 3, 10) for r in range(3)] == [(0, 3), (3, 3), (6, 4)]


def _replica_worker(rank, world, port, out_q):
    sys.path.insert(0, ROOT)
    import slam.net_amd.distributed as D
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    same = (0xFEDCBA9876543210, 0x8000000000000001)           # (words with the top bit set: they travel as int32 halves)
    a = D.replicas_equal(same)
    b = D.replicas_equal((same[0], same[1] + (1 if rank == world - 1 else 0)))     # one rank's ObstacleMap word is off by one
    c = D.replicas_equal((same[0] ^ ((1 << 63) if rank == 0 else 0), same[1]))     # rank 0 differs in the top bit only
    # a rank that could not make its words (None) still joins both collectives -- the others are in them -- and every rank gets False
    d = D.replicas_equal(None if rank == world - 1 else same)
    e = D.replicas_equal(same)                                                       # (and the next check is not disturbed)
    out_q.put((rank, a, b, c, d, e))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_replicas_equal_gloo(world):
    """The replica check over torch.distributed: equal words on every rank -> True on every rank, one
    deviating rank -> False on every rank."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_replica_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, a, b, c, d, e in res:
        assert a is True and b is False and c is False and d is False and e is True, (rank, a, b, c, d, e)
