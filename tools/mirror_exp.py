"""The figures behind DESIGN.md's section on the asynchronous host mirror (GPU only).

    python tools/mirror_exp.py            2048^2 map, 1080 rays, 16 384 candidates

1. one request on an idle device after a fused scan at a FIXED pose: the host time of slamhip_cs_holemap_mirror_async, the time
   from the request until the data have landed (slamhip_cs_holemap_mirror_wait), the pixels pushed and the resulting GB/s;
   the fused call's time to the pose while a push is in flight;
2. CoreSLAMProcessor.Update along a trajectory (a MOVING robot: what changes per scan is what a real run changes): us per scan
   without a mirror, with one request per scan, with one request per fourth scan.
Run it with an ordinary NumPy array (the staged form: the array does not own its pages) and with a page-aligned one (direct).
"""
import math
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402


def aligned_u16(n):
    raw = np.zeros(n + 4096, np.uint16)
    ofs = (-raw.ctypes.data % 4096) // 2
    return raw, raw[ofs:ofs + n]


def main():
    import slam.net_amd.coreslam as cs
    import slam.net_amd.sim as sim
    ctx = cs.Context(0)
    segs = sim.default_field()
    for mode in ("staged (ordinary array)", "direct (page-aligned array)"):
        dev = cs.CoreSlamDevice(ctx, 40.0, 2048, 512)
        rng = sim.PCG32(1234)
        traj = sim.trajectory(31)
        for p in traj[:-1]:
            _, xy = sim.make_scan(segs, p, 1080, rng)
            dev.set_scan(xy); dev.update_holemap(p)
        _, xy = sim.make_scan(segs, traj[-1], 1080, rng)
        base = (traj[-1] + np.array([0.03, -0.02, math.radians(1.0)], np.float32)).astype(np.float32)
        dev.set_scan(xy); dev.set_offsets(sim.gaussian_offsets(16383))
        keep, m = (None, np.zeros(2048 * 2048, np.uint16)) if mode.startswith("staged") else aligned_u16(2048 * 2048)
        dev.holemap_mirror_async(m); dev.holemap_mirror_wait()
        ts, tp, px = [], [], 0
        for _ in range(30):
            dev.search_and_update(base); ctx.synchronize()
            t0 = time.perf_counter(); dev.holemap_mirror_async(m); t1 = time.perf_counter()
            _, px = dev.holemap_mirror_wait(); t2 = time.perf_counter()
            ts.append(t1 - t0); tp.append(t2 - t0)
        lat = []
        for _ in range(30):
            ctx.synchronize(); dev.holemap_mirror_async(m)
            t0 = time.perf_counter(); dev.search_and_update(base); lat.append(time.perf_counter() - t0); dev.holemap_mirror_wait()
        print("%s: request %.1f us of host time | request -> landed %.1f us | %d px -> %.1f GB/s | pose with a push in flight %.1f us"
              % (mode, np.median(ts) * 1e6, np.median(tp) * 1e6, px, px * 2 / np.median(tp) / 1e9, np.median(lat) * 1e6))
        dev.holemap_mirror_release(); dev.close()
        # the processor along a trajectory
        ptraj = sim.trajectory(80); rngp = sim.PCG32(5)
        pscans = [sim.make_scan(segs, p, 1080, rngp)[0] for p in ptraj]
        proc = cs.CoreSLAMProcessor(40.0, 2048, 512, ptraj[0], 0.1, math.radians(10.0), 16383 // 64, 64, ctx=ctx)
        zero = np.zeros(3, np.float32)
        for i in range(10): proc.Update([cs.ScanSegment(pscans[i], zero)])
        ctx.synchronize(); t0 = time.perf_counter()
        for i in range(200): proc.Update([cs.ScanSegment(pscans[10 + i % 60], zero)])
        ctx.synchronize(); dt0 = (time.perf_counter() - t0) / 200
        keep2, pm = (None, np.zeros(2048 * 2048, np.uint16)) if mode.startswith("staged") else aligned_u16(2048 * 2048)
        proc.device.holemap_mirror_async(pm); proc.device.holemap_mirror_wait()
        pxs = []; t0 = time.perf_counter()
        for i in range(200):
            proc.Update([cs.ScanSegment(pscans[10 + i % 60], zero)])
            proc.device.holemap_mirror_async(pm)
            if i % 10 == 9: pxs.append(proc.device.holemap_mirror_wait()[1])
        proc.device.holemap_mirror_wait(); ctx.synchronize(); dt1 = (time.perf_counter() - t0) / 200
        ok = bool((pm == proc.device.holemap_download()).all())
        t0 = time.perf_counter()
        for i in range(200):
            proc.Update([cs.ScanSegment(pscans[10 + i % 60], zero)])
            if i % 4 == 3: proc.device.holemap_mirror_async(pm)
        proc.device.holemap_mirror_wait(); ctx.synchronize(); dt4 = (time.perf_counter() - t0) / 200
        print("   CoreSLAMProcessor.Update, moving robot: %.1f us per scan without a mirror | %.1f with a request per scan (%d px per request) | %.1f with one per 4th scan | mirror == download: %s"
              % (dt0 * 1e6, dt1 * 1e6, int(np.mean(pxs)), dt4 * 1e6, ok))
        proc.device.holemap_mirror_release(); proc.Dispose()
        del keep, keep2
    ctx.close()


if __name__ == "__main__":
    main()
