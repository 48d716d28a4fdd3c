"""K5/K4 experiment driver: Hector grid update + match timing (GPU only)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import slam.net_amd.capi as capi, slam.net_amd.coreslam as cs, slam.net_amd.hector as hs, slam.net_amd.sim as sim
ctx = cs.Context(0)
segs = sim.default_field()
rep = hs.MapRepMultiMap(40.0 / 2048, (2048, 2048), 3, ctx=ctx)
rng = sim.PCG32(3)
scans = []
for it in range(30):
    p = np.array([20 + 0.05 * it, 20 + 0.02 * it, 0.01 * it], np.float32)
    xy = sim.make_scan(segs, p, 1080, rng)[1]; scans.append((xy, p))
for xy, p in scans[:10]: rep.UpdateByScan(hs.ScanCloud(xy), p)
ctx.timing_reset(); ctx.timing_enable(-1)
for xy, p in scans[10:]: rep.UpdateByScan(hs.ScanCloud(xy), p)
ms5, n5 = ctx.timing_get(capi.K_HS_UPDATE)
m = hs.ScanMatcher(4)
xy, p = scans[-1]; scan = hs.ScanCloud(xy); hint = p + np.array([0.1, -0.08, 0.03], np.float32)
for _ in range(3): m.MatchData(rep, scan, hint)
ctx.timing_reset()
for _ in range(30): m.MatchData(rep, scan, hint)
ms4, n4 = ctx.timing_get(capi.K_HS_MATCH)
print("K5 update: %.1f us | K4 match: %.1f us" % (ms5 / n5 * 1e3, ms4 / n4 * 1e3))
