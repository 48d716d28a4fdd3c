"""PCIe-inclusive rate of the explicit-candidate entry point: slamhip_cs_distance_pxcs with 16 384 host
candidates per call (256 KB up, 8 bytes back), blocking; never the `value` of bench.py."""
import sys, os, time, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import slam.net_amd.coreslam as cs, slam.net_amd.sim as sim
K, size = 16384, 2048
ctx = cs.Context(0); dev = cs.CoreSlamDevice(ctx, 40.0, size, size // 4)
segs = sim.default_field(); rng = sim.PCG32(1234); traj = sim.trajectory(31)
for p in traj[:-1]:
    _, xy = sim.make_scan(segs, p, 1080, rng); dev.set_scan(xy); dev.update_holemap(p)
_, xy = sim.make_scan(segs, traj[-1], 1080, rng)
dev.set_scan(xy)
offs = sim.gaussian_offsets(K - 1, 0.1, math.radians(10.0))
poses = np.vstack([traj[-1][None], traj[-1][None] + offs]).astype(np.float32)
th = poses[:, 2].astype(np.float64)
pxcs = np.stack([poses[:, 0] * dev.hole_scale + 0.5, poses[:, 1] * dev.hole_scale + 0.5, np.cos(th) * dev.hole_scale, np.sin(th) * dev.hole_scale], 1).astype(np.float32)
for _ in range(10): dev.distance_pxcs(pxcs, want_all=False)
t0 = time.perf_counter()
for _ in range(200): dev.distance_pxcs(pxcs, want_all=False)
dt = (time.perf_counter() - t0) / 200
print("distance_pxcs, %d host candidates per call: %.1f us per call -> %.3g evals/s (PCIe and the unsorted-candidate path included)" % (K, dt * 1e6, K / dt))
