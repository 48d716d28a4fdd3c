"""K1 experiment driver: per-kernel-class timings for one search configuration (GPU only)."""
import sys, os, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import slam.net_amd.coreslam as cs, slam.net_amd.sim as sim

K = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
sig_deg = float(sys.argv[2]) if len(sys.argv) > 2 else 10.0
size = int(sys.argv[3]) if len(sys.argv) > 3 else 2048
R = int(sys.argv[4]) if len(sys.argv) > 4 else 1080
ctx = cs.Context(0); dev = cs.CoreSlamDevice(ctx, 40.0, size, size // 4)
segs = sim.default_field(); rng = sim.PCG32(1234); traj = sim.trajectory(31)
for p in traj[:-1]:
    _, xy = sim.make_scan(segs, p, R, rng); dev.set_scan(xy); dev.update_holemap(p)
_, xy = sim.make_scan(segs, traj[-1], R, rng)
base = (traj[-1] + np.array([0.03, -0.02, math.radians(1.0)], np.float32)).astype(np.float32)
dev.set_scan(xy); dev.set_offsets(sim.gaussian_offsets(K - 1, 0.1, math.radians(sig_deg)))
for _ in range(5): dev.search_shard(base, 0, K)
ctx.timing_reset(); ctx.timing_enable(-1)
for _ in range(50): dev.search_shard(base, 0, K)
tot = 0
for nm, k in (("prep", 0), ("dist", 1), ("reduce", 2)):
    ms, n = ctx.timing_get(k); tot += ms / max(n, 1); print(nm, "%.2f us" % (ms / max(n, 1) * 1e3), end=" | ")
print("sum %.2f us -> %.3g evals/s (kernels only)" % (tot * 1e3, K / (tot * 1e-3)))
print("selfcheck", dev.selfcheck_failures)
