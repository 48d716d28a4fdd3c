"""Secondary-kernel timings (K2 HoleMap update, K3 ObstacleMap update, K4 Hector match, K5 Hector grid update,
fused search+update) on one MI355X, with the algorithmic byte counts of SURVEY.md sec.8d.  Prints one JSON object."""
import json, math, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import slam.net_amd.capi as capi, slam.net_amd.coreslam as cs, slam.net_amd.hector as hs, slam.net_amd.sim as sim

out = {}
ctx = cs.Context(0)
segs = sim.default_field()
for size in (1024, 2048, 4096):
    dev = cs.CoreSlamDevice(ctx, 40.0, size, size // 4)
    rng = sim.PCG32(1234); traj = sim.trajectory(40)
    scans = [sim.make_scan(segs, p, 1080, rng)[1] for p in traj]
    for i in range(8):
        dev.set_scan(scans[i]); dev.update_holemap(traj[i]); dev.update_obstaclemap(traj[i])
    # (per-launch event pairs; three passes over the same 32 scans, the median pass is reported: a single slow launch -- the
    # box is shared with nothing, but clocks and the host's scheduler wander -- moves a 32-launch mean by a third)
    passes = []
    for rep_ in range(3):
        ctx.timing_reset(); ctx.timing_enable(-1)
        px = 0
        for i in range(8, 40):
            dev.set_scan(scans[i]); dev.update_holemap(traj[i]); px += dev.last_holemap_pixels; dev.update_obstaclemap(traj[i])
        ms2, n2 = ctx.timing_get(capi.K_CS_HOLEMAP); ms3, n3 = ctx.timing_get(capi.K_CS_OBSTACLE)
        ctx.timing_enable(0)
        passes.append((ms2 / n2, ms3 / n3, px / n2))
    ms2n = sorted(q[0] for q in passes)[1]; ms3n = sorted(q[1] for q in passes)[1]; pxn = passes[0][2]
    out["k2_holemap_%d" % size] = {"us_per_update": ms2n * 1e3, "blended_px_per_update": pxn,
                                    "algorithmic_GBps": 4 * pxn / (ms2n * 1e-3) / 1e9, "rays_per_s": 1080 / (ms2n * 1e-3), "passes": 3}
    out["k3_obstacle_%d" % (size // 4)] = {"us_per_update": ms3n * 1e3, "passes": 3}
    if size == 2048:      # fused config C3: search (16384 candidates) + both map updates in one call
        dev.set_offsets(sim.gaussian_offsets(16383))
        base = traj[-1]
        dev.set_scan(scans[-1])
        for _ in range(30): dev.search_and_update(base)
        dts = []
        for rep_ in range(5):
            ctx.synchronize()
            t0 = time.perf_counter()
            for _ in range(200): dev.search_and_update(base)
            ctx.synchronize()        # (the call returns with the pose; the last call's map updates belong to the figure)
            dts.append((time.perf_counter() - t0) / 200)
        dt = sorted(dts)[2]
        out["c3_fused_search_update_2048"] = {"us_per_scan": dt * 1e6, "scans_per_s": 1 / dt, "batches_of_200": 5}
    dev.close()

# Hector: 3-level 2048^2 pyramid, 1080 rays (config C4)
rep = hs.MapRepMultiMap(40.0 / 2048, (2048, 2048), 3, ctx=ctx)
rng = sim.PCG32(3)
scans = []
for it in range(20):
    p = np.array([20 + 0.05 * it, 20 + 0.02 * it, 0.01 * it], np.float32)
    xy = sim.make_scan(segs, p, 1080, rng)[1]; scans.append((xy, p))
for xy, p in scans[:10]: rep.UpdateByScan(hs.ScanCloud(xy), p)
p5 = []
for rep_ in range(3):
    ctx.timing_reset(); ctx.timing_enable(-1)
    for xy, p in scans[10:]: rep.UpdateByScan(hs.ScanCloud(xy), p)
    ms5, n5 = ctx.timing_get(capi.K_HS_UPDATE)
    p5.append(ms5 / n5)
ms5, n5 = sorted(p5)[1], 1
m = hs.ScanMatcher(4)
xy, p = scans[-1]; scan = hs.ScanCloud(xy); hint = p + np.array([0.1, -0.08, 0.03], np.float32)
for _ in range(3): m.MatchData(rep, scan, hint)
ctx.timing_reset()
for _ in range(50): m.MatchData(rep, scan, hint)
ms4, n4 = ctx.timing_get(capi.K_HS_MATCH)
t0 = time.perf_counter()
for _ in range(50): m.MatchData(rep, scan, hint)
wall1 = (time.perf_counter() - t0) / 50
B = 4096
hints = np.tile(hint, (B, 1)) + np.random.default_rng(0).normal(0, 0.05, (B, 3)).astype(np.float32) * np.array([1, 1, 0.2], np.float32)
m.MatchDataBatch(rep, scan, hints); ctx.timing_reset()
for _ in range(5): m.MatchDataBatch(rep, scan, hints)
ms4b, n4b = ctx.timing_get(capi.K_HS_MATCH)
ctx.timing_enable(0)
pt_iters = 1080 * 9
out["k5_hector_update_3lvl_2048"] = {"us_per_update": ms5 / n5 * 1e3}
out["k4_hector_match_3lvl_2048"] = {"kernel_us_single": ms4 / n4 * 1e3, "blocking_call_us_single": wall1 * 1e6,
                                    "batch": B, "kernel_us_batch": ms4b / n4b * 1e3, "matches_per_s_batched": B / (ms4b / n4b * 1e-3),
                                    "point_iterations_per_s_batched": B * pt_iters / (ms4b / n4b * 1e-3),
                                    "algorithmic_GBps_batched": B * pt_iters * 24 / (ms4b / n4b * 1e-3) / 1e9}
print(json.dumps(out, indent=1))
