"""How long the device needs to reach its sustained clocks: the headline search (and the same at 262 144 candidates) timed over
`steps` launches behind `warm` untimed ones, after a 0.2 s pause each (DESIGN.md sec.5 "Clocks")."""
import sys, os, math, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import slam.net_amd.coreslam as cs, slam.net_amd.sim as sim
ctx = cs.Context(0); dev = cs.CoreSlamDevice(ctx, 40.0, 2048, 512)
segs = sim.default_field(); rng = sim.PCG32(1234); traj = sim.trajectory(31)
for p in traj[:-1]:
    _, xy = sim.make_scan(segs, p, 1080, rng); dev.set_scan(xy); dev.update_holemap(p, 0.6, 50)
_, xy = sim.make_scan(segs, traj[-1], 1080, rng)
base = (traj[-1] + np.array([0.03, -0.02, math.radians(1.0)], np.float32)).astype(np.float32)
dev.set_scan(xy)
for K in (16384, 262144):
    dev.set_offsets(sim.gaussian_offsets(K - 1, 0.1, math.radians(10.0), seed=42))
    for warm, steps in ((6, 50), (20, 200), (2000, 200), (20, 2000), (6, 50)):
        ctx.synchronize(); time.sleep(0.2)
        for _ in range(warm): dev.search_shard_enqueue(base, 0, K)
        ctx.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps): dev.search_shard_enqueue(base, 0, K)
        ctx.synchronize()
        print(K, "warm", warm, "steps", steps, "us/step %.2f" % ((time.perf_counter() - t0) / steps * 1e6))
