import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import slam.net_amd.coreslam as cs, slam.net_amd.hector as hs, slam.net_amd.sim as sim
ctx = cs.Context(0)
segs = sim.default_field()
rep = hs.MapRepMultiMap(40.0 / 2048, (2048, 2048), 3, ctx=ctx)
rng = sim.PCG32(3)
for it in range(30):
    p = np.array([20 + 0.05 * it, 20 + 0.02 * it, 0.01 * it], np.float32)
    xy = sim.make_scan(segs, p, 1080, rng)[1]
    rep.UpdateByScan(hs.ScanCloud(xy), p)
ctx.synchronize()
