"""Host-side cost of the per-scan steps around the fused search + update (GPU only)."""
import sys, os, time, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import slam.net_amd.coreslam as cs, slam.net_amd.sim as sim
ctx = cs.Context(0); dev = cs.CoreSlamDevice(ctx, 40.0, 2048, 512)
segs = sim.default_field(); rng = sim.PCG32(5); traj = sim.trajectory(40)
scans = [sim.make_scan(segs, p, 1080, rng)[1] for p in traj]
for i in range(8):
    dev.set_scan(scans[i]); dev.update_holemap(traj[i]); dev.update_obstaclemap(traj[i])
def timeit(f, n=200):
    f(); t0 = time.perf_counter()
    for _ in range(n): f()
    return (time.perf_counter() - t0) / n * 1e6
print("set_scan            %.1f us" % timeit(lambda: dev.set_scan(scans[9])))
print("generate_offsets    %.1f us" % timeit(lambda: (dev.generate_offsets(16383, 0.1, math.radians(10.0), seed=1, stream=2), ctx.synchronize())))
dev.generate_offsets(16383, 0.1, math.radians(10.0), seed=1, stream=2)
print("search_and_update   %.1f us" % timeit(lambda: dev.search_and_update(traj[9])))
def both():
    dev.set_scan(scans[9]); dev.generate_offsets(16383, 0.1, math.radians(10.0), seed=1, stream=2); dev.search_and_update(traj[9])
print("all three           %.1f us" % timeit(both))
