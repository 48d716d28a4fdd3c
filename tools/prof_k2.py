import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import slam.net_amd.coreslam as cs, slam.net_amd.sim as sim
size = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
ctx = cs.Context(0); dev = cs.CoreSlamDevice(ctx, 40.0, size, size // 4)
segs = sim.default_field(); rng = sim.PCG32(1234); traj = sim.trajectory(40)
scans = [sim.make_scan(segs, p, 1080, rng)[1] for p in traj]
for i in range(40):
    dev.set_scan(scans[i]); dev.update_holemap(traj[i])
ctx.synchronize()
