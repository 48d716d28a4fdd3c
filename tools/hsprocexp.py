"""HectorSLAMProcessor.Update per scan (GPU only), on its own context: set_scan + match + gated grid update."""
import sys, os, time, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import slam.net_amd.coreslam as cs, slam.net_amd.hector as hs, slam.net_amd.sim as sim
side = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
levels = int(sys.argv[2]) if len(sys.argv) > 2 else 3
rays = int(sys.argv[3]) if len(sys.argv) > 3 else 1080
every = float(sys.argv[4]) if len(sys.argv) > 4 else 0.0      # MinDistanceDiffForMapUpdate (0: every scan updates the map)
ctx = cs.Context(0)
segs = sim.default_field(); rng = sim.PCG32(5)
traj, _ = sim.lap_trajectory(260, 0.1)
proc = hs.HectorSLAMProcessor(40.0 / side, (side, side), traj[0].copy(), levels, 4, ctx=ctx)
proc.MinDistanceDiffForMapUpdate = every
proc.MinAngleDiffForMapUpdate = math.radians(8.0) if every > 0 else 0.0
scans = [hs.ScanCloud(sim.make_scan(segs, p, rays, rng)[1]) for p in traj]
for i in range(10): proc.Update(scans[0], proc.MatchPose, True)
for i in range(10, 40): proc.Update(scans[i - 9], proc.MatchPose, False)
ctx.synchronize()
t0 = time.perf_counter(); n_up = 0
for i in range(40, 240): n_up += 1 if proc.Update(scans[i - 9], proc.MatchPose, False) else 0
ctx.synchronize()
dt = (time.perf_counter() - t0) / 200
print("HectorSLAMProcessor.Update (%d^2 x %d, %d rays, %d of 200 scans update the map): %.1f us per scan" % (side, levels, rays, n_up, dt * 1e6))
