"""Generate the committed golden fixtures under tests/golden/ (TEST INFRASTRUCTURE ONLY).

The reference has no golden vectors and cannot run here (PARITY UNPINNED, oracle/oracle.h), so
the fixtures are produced by the independent NumPy restatement (oracle/np_oracle.py) from
seeded synthetic inputs; tests/test_golden.py checks the C oracle against them on CPU and the
HIP path against them on the GPU.  Run:  python oracle/gen_golden.py
All trig in the fixtures is the deterministic float trig (det_sincos), and candidate (px,py,c,s)
tuples are stored explicitly, so no libm enters the expected values.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

import np_oracle as npo            # noqa: E402
import slam.net_amd.sim as sim     # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")


def build_map(size, n_rays, n_updates, hole_width, seed):
    """Reset to 32750 and apply n_updates mapping updates along the sec.8d trajectory."""
    segs = sim.default_field()
    scale = npo.map_scale(size, 40.0)
    pix = np.full(size * size, 32750, np.uint16)
    traj = sim.trajectory(n_updates)
    rng = sim.PCG32(seed)
    scans, pxcs_list = [], []
    for i in range(n_updates):
        rays, xy = sim.make_scan(segs, traj[i], n_rays, rng)
        pxcs = npo.poses_to_pxcs(traj[i][None], scale)[0]
        npo.update_holemap_pxcs(pix, size, scale, xy, pxcs, hole_width, 50)
        scans.append(xy); pxcs_list.append(pxcs)
    return pix, scale, scans, np.array(pxcs_list, np.float32), traj


def gen_distance(name, size, n_rays, n_updates, K, hole_width=0.6, seed=1234):
    pix, scale, scans, _, traj = build_map(size, n_rays, n_updates, hole_width, seed)
    segs = sim.default_field()
    true_pose = sim.trajectory(n_updates + 1)[-1]
    rays, xy = sim.make_scan(segs, true_pose, n_rays, sim.PCG32(seed + 1))
    base = (true_pose + np.array([0.03, -0.02, np.radians(1.0)], np.float32)).astype(np.float32)
    offs = sim.gaussian_offsets(K - 1)
    poses = np.vstack([base[None], base[None] + offs]).astype(np.float32)
    pxcs = npo.poses_to_pxcs(poses, scale)
    dist = npo.distance_batch_pxcs(pix, size, xy, pxcs)
    np.savez_compressed(os.path.join(OUT, name), size=size, scale=np.float32(scale), pixels=pix, xy=xy,
                        base=base, offs=offs, pxcs=pxcs, dist=dist, best=np.int32(npo.argmin_first(dist)))
    print(name, "K", K, "best", npo.argmin_first(dist), "min", dist.min(), "max", dist.max())


def gen_holemap(name, size, n_rays, n_updates, hole_width, seed=99):
    segs = sim.default_field()
    scale = npo.map_scale(size, 40.0)
    pix = np.full(size * size, 32750, np.uint16)
    traj = sim.trajectory(n_updates, start=(20.0, 20.0, 0.3), step=(0.11, -0.07, np.radians(2.0)))
    rng = sim.PCG32(seed)
    xys, pxcs_all, counts, after1 = [], [], [], None
    for i in range(n_updates):
        rays, xy = sim.make_scan(segs, traj[i], n_rays, rng)
        pxcs = npo.poses_to_pxcs(traj[i][None], scale)[0]
        counts.append(npo.update_holemap_pxcs(pix, size, scale, xy, pxcs, hole_width, 50))
        if i == 0:
            after1 = pix.copy()
        xys.append(xy); pxcs_all.append(pxcs)
    np.savez_compressed(os.path.join(OUT, name), size=size, scale=np.float32(scale),
                        hole_width=np.float32(hole_width), quality=50, xy=np.array(xys, np.float32),
                        pxcs=np.array(pxcs_all, np.float32), counts=np.array(counts, np.int64),
                        after1=after1, after_all=pix)
    print(name, "blended px per update", counts)


def gen_obstacle(name, size, n_rays, n_updates, seed=77):
    segs = sim.default_field()
    scale = npo.map_scale(size, 40.0)
    pix = np.full((size, size), -5, np.int8)
    traj = sim.trajectory(n_updates, start=(20.0, 20.0, -0.4), step=(0.15, 0.1, np.radians(3.0)))
    rng = sim.PCG32(seed)
    xys, pxcs_all, after1 = [], [], None
    for i in range(n_updates):
        rays, xy = sim.make_scan(segs, traj[i], n_rays, rng)
        pxcs = npo.poses_to_pxcs(traj[i][None], scale)[0]
        npo.update_obstaclemap_pxcs(pix, size, xy, pxcs, 10)
        if i == 0:
            after1 = pix.copy()
        xys.append(xy); pxcs_all.append(pxcs)
    np.savez_compressed(os.path.join(OUT, name), size=size, scale=np.float32(scale), max_hits=10,
                        xy=np.array(xys, np.float32), pxcs=np.array(pxcs_all, np.float32),
                        after1=after1, after_all=pix)
    print(name, "values", np.unique(pix))


def gen_hector(name, side, cell, n_rays, n_updates, seed=5):
    segs = sim.default_field()
    g = npo.NpGrid(cell, side, side)
    traj = sim.trajectory(n_updates, start=(20.0, 20.0, 0.1), step=(0.2, 0.1, np.radians(2.5)))
    rng = sim.PCG32(seed)
    xys = []
    for i in range(n_updates):
        rays, xy = sim.make_scan(segs, traj[i], n_rays, rng)
        g.update_by_scan(xy, traj[i])
        xys.append(xy)
    rays, xy = sim.make_scan(segs, traj[-1], n_rays, sim.PCG32(seed + 1))
    est_world = (traj[-1] + np.array([0.12, -0.08, 0.02], np.float32)).astype(np.float32)
    est_map = np.array([est_world[0] * g.stm, est_world[1] * g.stm, est_world[2]], np.float32)
    H1, d1 = g.hessian(xy, est_map, 1)
    H4, d4 = g.hessian(xy, est_map, 4)
    np.savez_compressed(os.path.join(OUT, name), side=side, cell=np.float32(cell), poses=traj,
                        xy=np.array(xys, np.float32), value=g.value, upd=g.upd,
                        match_xy=xy, est_map=est_map, H1=H1, d1=d1, H4=H4, d4=d4)
    print(name, "touched cells", int((g.upd >= 0).sum()), "H1", H1.ravel()[:3], "d1", d1)


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    gen_distance("cs_distance_64_r90_k256.npz", 64, 90, 6, 256)
    gen_distance("cs_distance_256_r360_k256.npz", 256, 360, 8, 256, hole_width=2.0)
    gen_distance("cs_distance_400_r1080_k64.npz", 400, 1080, 4, 64)
    gen_holemap("cs_holemap_64_r90.npz", 64, 90, 10, 0.6)
    gen_holemap("cs_holemap_256_r360_hw2.npz", 256, 360, 10, 2.0)
    gen_obstacle("cs_obstacle_64_r360.npz", 64, 360, 10)
    gen_hector("hs_grid_200_r180.npz", 200, 0.2, 180, 6)
