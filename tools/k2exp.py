"""K2 experiment driver: HoleMap update timing at one size (GPU only)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import slam.net_amd.capi as capi, slam.net_amd.coreslam as cs, slam.net_amd.sim as sim
size = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
ctx = cs.Context(0); dev = cs.CoreSlamDevice(ctx, 40.0, size, size // 4)
segs = sim.default_field(); rng = sim.PCG32(1234); traj = sim.trajectory(40)
scans = [sim.make_scan(segs, p, 1080, rng)[1] for p in traj]
for i in range(8):
    dev.set_scan(scans[i]); dev.update_holemap(traj[i]); dev.update_obstaclemap(traj[i])
ctx.timing_reset(); ctx.timing_enable(-1)
px = 0
for i in range(8, 40):
    dev.set_scan(scans[i]); dev.update_holemap(traj[i]); px += dev.last_holemap_pixels; dev.update_obstaclemap(traj[i])
ms2, n2 = ctx.timing_get(capi.K_CS_HOLEMAP); ms3, n3 = ctx.timing_get(capi.K_CS_OBSTACLE)
print("K2 %d: %.1f us/update (%.0f px) | K3 %d: %.1f us" % (size, ms2 / n2 * 1e3, px / n2, size // 4, ms3 / n3 * 1e3))
