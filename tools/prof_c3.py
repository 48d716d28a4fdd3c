"""The fused scan (configuration C3) back to back, for a kernel trace at sustained clocks: 2048^2 map, 1080 rays, 16 384 candidates,
`calls` fused search + update calls on one scan (python3 tools/prof_c3.py [calls]); and `ups` stand-alone HoleMap updates back to
back behind them (no host work in between: the launches queue up)."""
import sys, os, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import slam.net_amd.coreslam as cs, slam.net_amd.sim as sim
calls = int(sys.argv[1]) if len(sys.argv) > 1 else 300
K, size, R = 16384, 2048, 1080
ctx = cs.Context(0); dev = cs.CoreSlamDevice(ctx, 40.0, size, size // 4)
segs = sim.default_field(); rng = sim.PCG32(1234); traj = sim.trajectory(31)
for p in traj[:-1]:
    _, xy = sim.make_scan(segs, p, R, rng); dev.set_scan(xy); dev.update_holemap(p)
_, xy = sim.make_scan(segs, traj[-1], R, rng)
base = (traj[-1] + np.array([0.03, -0.02, math.radians(1.0)], np.float32)).astype(np.float32)
dev.set_scan(xy); dev.set_offsets(sim.gaussian_offsets(K - 1, 0.1, math.radians(10.0)))
for _ in range(calls): dev.search_and_update(base)
ctx.synchronize()
