"""Blocking-call latencies of the operator entry points (one MI355X): what a per-scan caller waits for."""
import sys, os, math, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import slam.net_amd.coreslam as cs, slam.net_amd.sim as sim

def med(f, n=300, warm=30, idle=None):
    for _ in range(warm): f()
    ts = []
    for _ in range(n):
        if idle: idle()                  # (a call that returns before all its device work is done: start from an idle device)
        t0 = time.perf_counter(); f(); ts.append(time.perf_counter() - t0)
    ts.sort()
    return {"median_us": ts[len(ts) // 2] * 1e6, "p10_us": ts[len(ts) // 10] * 1e6, "p90_us": ts[len(ts) * 9 // 10] * 1e6}

K, size, R = 16384, 2048, 1080
ctx = cs.Context(0); dev = cs.CoreSlamDevice(ctx, 40.0, size, size // 4)
segs = sim.default_field(); rng = sim.PCG32(1234); traj = sim.trajectory(31)
for p in traj[:-1]:
    _, xy = sim.make_scan(segs, p, R, rng); dev.set_scan(xy); dev.update_holemap(p)
_, xy = sim.make_scan(segs, traj[-1], R, rng)
base = (traj[-1] + np.array([0.03, -0.02, math.radians(1.0)], np.float32)).astype(np.float32)
dev.set_scan(xy); dev.set_offsets(sim.gaussian_offsets(K - 1, 0.1, math.radians(10.0)))
out = {"search_shard_blocking": med(lambda: dev.search_shard(base, 0, K)),
       "search_and_update_to_pose": med(lambda: dev.search_and_update(base), idle=ctx.synchronize),       # (the map updates run on)
       "search_and_update_back_to_back": med(lambda: dev.search_and_update(base)),
       "update_holemap_blocking": med(lambda: dev.update_holemap(base)),
       "update_obstaclemap_blocking": med(lambda: dev.update_obstaclemap(base))}
print(json.dumps(out, indent=1))

# Hector: 3-level 2048^2 pyramid, 1080 rays
import slam.net_amd.hector as hs
rep = hs.MapRepMultiMap(40.0 / 2048, (2048, 2048), 3, ctx=ctx)
rng = sim.PCG32(3)
scans = []
for it in range(12):
    p = np.array([20 + 0.05 * it, 20 + 0.02 * it, 0.01 * it], np.float32)
    scans.append((sim.make_scan(segs, p, 1080, rng)[1], p))
for xy2, p in scans[:10]: rep.UpdateByScan(hs.ScanCloud(xy2), p)
m = hs.ScanMatcher(4)
xy2, p = scans[-1]; scan = hs.ScanCloud(xy2); hint = p + np.array([0.1, -0.08, 0.03], np.float32)
out2 = {"hector_match_blocking": med(lambda: m.MatchData(rep, scan, hint)),
        "hector_update_blocking": med(lambda: rep.UpdateByScan(scan, p), n=100, warm=10)}
print(json.dumps(out2, indent=1))
