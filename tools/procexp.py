"""End-to-end CoreSLAMProcessor.Update latency per scan (GPU only): host scan prep + search + both map updates."""
import sys, os, time, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import slam.net_amd.coreslam as cs, slam.net_amd.sim as sim
size = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
K = int(sys.argv[2]) if len(sys.argv) > 2 else 16384
ctx = cs.Context(0)
segs = sim.default_field(); rng = sim.PCG32(5); traj = sim.trajectory(80)
proc = cs.CoreSLAMProcessor(40.0, size, size // 4, traj[0], 0.1, math.radians(10.0), (K - 1) // 64, 64, ctx=ctx)
scans = [sim.make_scan(segs, p, 1080, rng) for p in traj]
def seg(i):
    rays, xy = scans[i]
    return [cs.ScanSegment(rays, np.zeros(3, np.float32))]
for i in range(10): proc.Update(seg(i))
ctx.synchronize()
t0 = time.perf_counter()
for i in range(10, 210): proc.Update(seg(10 + i % 60))
ctx.synchronize()            # (Update returns with the pose; the last scan's map updates belong to the figure)
dt = (time.perf_counter() - t0) / 200
print("CoreSLAMProcessor.Update (%d^2, %d candidates): %.1f us per scan" % (size, K, dt * 1e6))
