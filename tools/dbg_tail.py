import sys, os, math
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo")); sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "oracle"))
import numpy as np
import slam.net_amd.coreslam as cs, slam.net_amd.sim as sim, oracle_c as oc
oc.set_trig_mode(oc.TRIG_DET)
ctx = cs.Context(0); size=int(os.environ.get('DBG_SIZE','400')); dev = cs.CoreSlamDevice(ctx, 40.0, size, 100); NR=int(os.environ.get('DBG_R','360'))
segs = sim.default_field(); rng = sim.PCG32(1234)
for p in sim.trajectory(8):
    _, xy = sim.make_scan(segs, p, NR, rng); dev.set_scan(xy); dev.update_holemap(p)
pix = dev.holemap_download()
_, xy = sim.make_scan(segs, sim.trajectory(9)[-1], NR, sim.PCG32(99))
base = (sim.trajectory(9)[-1] + np.array([0.03,-0.02,math.radians(1.0)],np.float32)).astype(np.float32)
dev.set_scan(xy)
for K in (500, 2000, 4001):
    offs = sim.gaussian_offsets(K-1); poses = np.vstack([base[None], base[None]+offs]).astype(np.float32)
    order = np.argsort(poses[:,2], kind="stable")
    d, bi, bd = dev.distance_poses(poses[order])
    ref, rbi, rbd = oc.distance_batch_pxcs(pix, size, xy, oc.poses_to_pxcs(poses[order], dev.hole_scale))
    bad = np.flatnonzero(d != ref)
    print("K", K, "mismatch", bad.size, "first", bad[:10], d[bad[:5]], ref[bad[:5]], "ratio", (d[bad[:5]]/np.maximum(ref[bad[:5]],1)))
    if bad.size: print("  bad sub-batches:", np.unique(bad >> 8), " of", (K+255)//256)
print("selfcheck failures", dev.selfcheck_failures)
