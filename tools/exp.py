"""One-off timing experiments on a single MI355X (GPU only) -- the drivers behind the figures quoted in DESIGN.md.

    python tools/exp.py k1 [candidates sigma_theta_deg map_size rays]   K1: per-kernel-class timings of one search configuration
    python tools/exp.py k2 [map_size]                                    K2 / K3: HoleMap and ObstacleMap update, us per update
    python tools/exp.py k5                                               K5 / K4: Hector grid update and single match
    python tools/exp.py proc [map_size candidates]                       CoreSLAMProcessor.Update, us per scan end to end
    python tools/exp.py prochost                                         host-side cost of the per-scan steps around the fused call
    python tools/exp.py hsproc [side levels rays min_dist]               HectorSLAMProcessor.Update, us per scan
    python tools/exp.py c3 [calls]                                       slamhip_cs_search_and_update on one scan, back to back: us per call, and the calls' distribution
    python tools/exp.py pcie                                             PCIe-inclusive rate of slamhip_cs_distance_pxcs (never bench.py's `value`)

Build with SLAMHIP_K1_TIMES=1 / SLAMHIP_K2_TIMES=1 / SLAMHIP_K4_TIMES=1 (python -m slam.net_amd.build --force) to get the
in-kernel phase stamps of the respective kernel printed by the k1 / k2 / k5 experiments.
"""
import math
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402


def exp_k1(argv):
    """K1 experiment driver: per-kernel-class timings for one search configuration (GPU only)."""
    import slam.net_amd.capi as capi, slam.net_amd.coreslam as cs, slam.net_amd.hector as hs, slam.net_amd.sim as sim  # noqa: E401,F401

    K = int(argv[0]) if len(argv) > 0 else 16384
    sig_deg = float(argv[1]) if len(argv) > 1 else 10.0
    size = int(argv[2]) if len(argv) > 2 else 2048
    R = int(argv[3]) if len(argv) > 3 else 1080
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

def exp_k2(argv):
    """K2 experiment driver: HoleMap update timing at one size (GPU only)."""
    import slam.net_amd.capi as capi, slam.net_amd.coreslam as cs, slam.net_amd.hector as hs, slam.net_amd.sim as sim  # noqa: E401,F401
    size = int(argv[0]) if len(argv) > 0 else 2048
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

def exp_k5(argv):
    """K5/K4 experiment driver: Hector grid update + match timing (GPU only)."""
    import slam.net_amd.capi as capi, slam.net_amd.coreslam as cs, slam.net_amd.hector as hs, slam.net_amd.sim as sim  # noqa: E401,F401
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

def exp_proc(argv):
    """End-to-end CoreSLAMProcessor.Update latency per scan (GPU only): host scan prep + search + both map updates."""
    import slam.net_amd.capi as capi, slam.net_amd.coreslam as cs, slam.net_amd.hector as hs, slam.net_amd.sim as sim  # noqa: E401,F401
    size = int(argv[0]) if len(argv) > 0 else 2048
    K = int(argv[1]) if len(argv) > 1 else 16384
    ctx = cs.Context(0)
    segs = sim.default_field(); rng = sim.PCG32(5); traj = sim.trajectory(80)
    proc = cs.CoreSLAMProcessor(40.0, size, size // 4, traj[0], 0.1, math.radians(10.0), (K - 1) // 64, 64, ctx=ctx)
    scans = [sim.make_scan(segs, p, 1080, rng) for p in traj]
    def seg(i):
        rays, xy = scans[i]
        return [cs.ScanSegment(rays, np.zeros(3, np.float32))]
    for i in range(10): proc.Update(seg(i))
    ctx.synchronize()
    t0 = time.perf_counter()
    for i in range(10, 210): proc.Update(seg(10 + i % 60))
    ctx.synchronize()            # (Update returns with the pose; the last scan's map updates belong to the figure)
    dt = (time.perf_counter() - t0) / 200
    print("CoreSLAMProcessor.Update (%d^2, %d candidates): %.1f us per scan" % (size, K, dt * 1e6))

def exp_prochost(argv):
    """Host-side cost of the per-scan steps around the fused search + update (GPU only)."""
    import slam.net_amd.capi as capi, slam.net_amd.coreslam as cs, slam.net_amd.hector as hs, slam.net_amd.sim as sim  # noqa: E401,F401
    ctx = cs.Context(0); dev = cs.CoreSlamDevice(ctx, 40.0, 2048, 512)
    segs = sim.default_field(); rng = sim.PCG32(5); traj = sim.trajectory(40)
    scans = [sim.make_scan(segs, p, 1080, rng)[1] for p in traj]
    for i in range(8):
        dev.set_scan(scans[i]); dev.update_holemap(traj[i]); dev.update_obstaclemap(traj[i])
    def timeit(f, n=200):
        f(); t0 = time.perf_counter()
        for _ in range(n): f()
        return (time.perf_counter() - t0) / n * 1e6
    print("set_scan            %.1f us" % timeit(lambda: dev.set_scan(scans[9])))
    print("generate_offsets    %.1f us" % timeit(lambda: (dev.generate_offsets(16383, 0.1, math.radians(10.0), seed=1, stream=2), ctx.synchronize())))
    dev.generate_offsets(16383, 0.1, math.radians(10.0), seed=1, stream=2)
    print("search_and_update   %.1f us" % timeit(lambda: dev.search_and_update(traj[9])))
    def both():
        dev.set_scan(scans[9]); dev.generate_offsets(16383, 0.1, math.radians(10.0), seed=1, stream=2); dev.search_and_update(traj[9])
    print("all three           %.1f us" % timeit(both))

def exp_hsproc(argv):
    """HectorSLAMProcessor.Update per scan (GPU only), on its own context: set_scan + match + gated grid update."""
    import slam.net_amd.capi as capi, slam.net_amd.coreslam as cs, slam.net_amd.hector as hs, slam.net_amd.sim as sim  # noqa: E401,F401
    side = int(argv[0]) if len(argv) > 0 else 2048
    levels = int(argv[1]) if len(argv) > 1 else 3
    rays = int(argv[2]) if len(argv) > 2 else 1080
    every = float(argv[3]) if len(argv) > 3 else 0.0      # MinDistanceDiffForMapUpdate (0: every scan updates the map)
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

def exp_c3(argv):
    """Configuration C3: the fused search + both map updates on the headline scan, `calls` times back to back (what bench.py's
    c3 entry times), then the same calls timed one by one: median / p90 / p99 of a call."""
    import slam.net_amd.coreslam as cs, slam.net_amd.sim as sim  # noqa: E401
    calls = int(argv[0]) if argv else 300
    K, size, R = 16384, 2048, 1080
    ctx = cs.Context(0); dev = cs.CoreSlamDevice(ctx, 40.0, size, size // 4)
    segs = sim.default_field(); rng = sim.PCG32(1234); traj = sim.trajectory(31)
    for p in traj[:-1]:
        _, xy = sim.make_scan(segs, p, R, rng); dev.set_scan(xy); dev.update_holemap(p)
    _, xy = sim.make_scan(segs, traj[-1], R, rng)
    base = (traj[-1] + np.array([0.03, -0.02, math.radians(1.0)], np.float32)).astype(np.float32)
    dev.set_scan(xy); dev.set_offsets(sim.gaussian_offsets(K - 1, 0.1, math.radians(10.0)))
    for _ in range(30): dev.search_and_update(base)
    ctx.synchronize()
    t0 = time.perf_counter()
    for _ in range(calls): dev.search_and_update(base)
    ctx.synchronize()
    print("fused us per call %.2f (%d calls back to back, the last call's updates included)" % ((time.perf_counter() - t0) / calls * 1e6, calls))
    ts = []
    for _ in range(max(calls, 1000)):
        t1 = time.perf_counter(); dev.search_and_update(base); ts.append((time.perf_counter() - t1) * 1e6)
    ts = np.array(ts)
    print("one call, us: median %.1f | p90 %.1f | p99 %.1f | mean %.1f | max %.1f" % (np.median(ts), np.percentile(ts, 90), np.percentile(ts, 99), ts.mean(), ts.max()))

def exp_pcie(argv):
    """PCIe-inclusive rate of the explicit-candidate entry point: slamhip_cs_distance_pxcs with 16 384 host candidates per call (256 KB up, 8 bytes back), blocking; never the `value` of bench.py."""
    import slam.net_amd.capi as capi, slam.net_amd.coreslam as cs, slam.net_amd.hector as hs, slam.net_amd.sim as sim  # noqa: E401,F401
    K, size = 16384, 2048
    ctx = cs.Context(0); dev = cs.CoreSlamDevice(ctx, 40.0, size, size // 4)
    segs = sim.default_field(); rng = sim.PCG32(1234); traj = sim.trajectory(31)
    for p in traj[:-1]:
        _, xy = sim.make_scan(segs, p, 1080, rng); dev.set_scan(xy); dev.update_holemap(p)
    _, xy = sim.make_scan(segs, traj[-1], 1080, rng)
    dev.set_scan(xy)
    offs = sim.gaussian_offsets(K - 1, 0.1, math.radians(10.0))
    poses = np.vstack([traj[-1][None], traj[-1][None] + offs]).astype(np.float32)
    th = poses[:, 2].astype(np.float64)
    pxcs = np.stack([poses[:, 0] * dev.hole_scale + 0.5, poses[:, 1] * dev.hole_scale + 0.5, np.cos(th) * dev.hole_scale, np.sin(th) * dev.hole_scale], 1).astype(np.float32)
    for _ in range(10): dev.distance_pxcs(pxcs, want_all=False)
    t0 = time.perf_counter()
    for _ in range(200): dev.distance_pxcs(pxcs, want_all=False)
    dt = (time.perf_counter() - t0) / 200
    print("distance_pxcs, %d host candidates per call: %.1f us per call -> %.3g evals/s (PCIe and the unsorted-candidate path included)" % (K, dt * 1e6, K / dt))


if __name__ == "__main__":
    which = sys.argv[1] if len(sys.argv) > 1 else ""
    fn = globals().get("exp_" + which)
    if fn is None:
        print(__doc__)
        sys.exit(2)
    fn(sys.argv[2:])
