"""Hector matches for a rocprofv3 pass (python3 tools/prof_k4.py [single|batch]): a 3-level 2048^2 pyramid built from twelve scans,
then 60 single matches (one persistent workgroup each: the latency form) or 6 batches of 4096 hints (the throughput form)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import slam.net_amd.coreslam as cs, slam.net_amd.hector as hs, slam.net_amd.sim as sim
mode = sys.argv[1] if len(sys.argv) > 1 else "single"
ctx = cs.Context(0)
segs = sim.default_field()
rep = hs.MapRepMultiMap(40.0 / 2048, (2048, 2048), 3, ctx=ctx)
rng = sim.PCG32(3)
scans = []
for it in range(12):
    p = np.array([20 + 0.05 * it, 20 + 0.02 * it, 0.01 * it], np.float32)
    scans.append((sim.make_scan(segs, p, 1080, rng)[1], p))
for xy, p in scans: rep.UpdateByScan(hs.ScanCloud(xy), p)
m = hs.ScanMatcher(4)
xy, p = scans[-1]; scan = hs.ScanCloud(xy); hint = p + np.array([0.1, -0.08, 0.03], np.float32)
if mode == "single":
    for _ in range(60): m.MatchData(rep, scan, hint)
elif mode == "fresh":       # every match on a freshly set scan (the per-scan flow's form: the match pulls the scan from the staging block), scans and hints alternating
    for k in range(60):
        xy2, p2 = scans[-1 - (k % 4)]
        m.MatchData(rep, hs.ScanCloud(xy2.copy()), p2 + np.array([0.1, -0.08, 0.03], np.float32))
else:
    B = 4096
    hints = np.tile(hint, (B, 1)) + np.random.default_rng(0).normal(0, 0.05, (B, 3)).astype(np.float32) * np.array([1, 1, 0.2], np.float32)
    for _ in range(6): m.MatchDataBatch(rep, scan, hints)
ctx.synchronize()
