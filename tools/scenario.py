#!/usr/bin/env python3
"""Headless scenario runner: the simulator's scan loop (Simulation/MainWindow.xaml.cs:136-210) without the GUI.

A lidar stands still for --hold scans (the maps initialise without matching), then drives one closed lap around the
inner obstacle of the default field (Simulation/Field.cs:45-69, scale 30, offset (5, 5)); every scan goes through the CoreSLAMProcessor and HectorSLAMProcessor mirrors exactly as `Scan()` feeds them:
CoreSLAM gets one segment posed at its own last estimate (:159), Hector gets the robot-frame cloud with its last match as
the hint and maps without matching for the first 10 loops (:179).  Prints one JSON object: the trajectory-error curve of
both estimators (distance / heading error against the true pose every --every scans, RMS, maximum, error on return to the
start) and the wall time per Update.

    python tools/scenario.py [--scans 490 --rays 400 --hole-map 256 --obstacle-map 64 --iterations 1000 --threads 4]

Needs a GPU (the mirrors call libslamhip); defaults are the simulator's constructor arguments (:69-86).
"""
import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402


def rad_diff(a, b):
    d = (a - b + math.pi) % (2.0 * math.pi) - math.pi
    return d


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scans", type=int, default=0, help="0 = one full lap")
    ap.add_argument("--step", type=float, default=0.1, help="metres between scans")
    ap.add_argument("--rays", type=int, default=400)                     # numScanPoints :35
    ap.add_argument("--hole-map", type=int, default=256)                 # :69
    ap.add_argument("--obstacle-map", type=int, default=64)
    ap.add_argument("--iterations", type=int, default=1000)
    ap.add_argument("--threads", type=int, default=4)
    ap.add_argument("--hector-side", type=int, default=400)              # :76
    ap.add_argument("--hector-levels", type=int, default=4)
    ap.add_argument("--hold", type=int, default=12, help="scans taken standing at the start before driving off: CoreSLAM maps "
                    "its first PositionSearchBeginning (5) scans and Hector its first 10 loops at the given pose without matching")
    ap.add_argument("--every", type=int, default=25)
    ap.add_argument("--seed", type=int, default=7)
    ap.add_argument("--lattice", action="store_true", help="CoreSLAM's candidates as a heading lattice (slamhip_csproc_set_lattice; from 12 289 candidates on)")
    a = ap.parse_args()

    import slam.net_amd.coreslam as cs
    import slam.net_amd.hector as hs
    import slam.net_amd.sim as sim

    segs = sim.default_field()
    traj, lap_len = sim.lap_trajectory(a.scans or None, a.step)
    traj = np.concatenate([np.repeat(traj[:1], a.hold, axis=0), traj[1:]])
    start = traj[0].copy()
    ctx = cs.Context(0)
    core = cs.CoreSLAMProcessor(40.0, a.hole_map, a.obstacle_map, start, 0.1, math.radians(10.0), a.iterations, a.threads, ctx=ctx)
    core.HoleWidth = 2.0                                                 # :71
    core.SetSeed(a.seed)
    if a.lattice:
        core.SetLattice(True)
    ctx_h = cs.Context(0)
    hect = hs.HectorSLAMProcessor(40.0 / a.hector_side, (a.hector_side, a.hector_side), start, a.hector_levels, a.threads, ctx=ctx_h)
    hect.MinDistanceDiffForMapUpdate = 0.4                               # :78-79
    hect.MinAngleDiffForMapUpdate = math.radians(8.0)
    for l, it in enumerate([7, 4, 4, 4][:a.hector_levels]):              # :83-86
        hect.MapRep.Maps[l].EstimateIterations = it

    rng = sim.PCG32(a.seed)
    curve = []
    err = {"core": [], "hector": []}
    t_core = t_hect = 0.0
    hector_lost_at = None
    for loop, tp in enumerate(traj):
        rays, xy = sim.make_scan(segs, tp, a.rays, rng)
        # (CoreSLAMProcessor.Update returns with the pose while its map updates run on: the two processors have a context --
        # a stream -- each, as two independent SLAM instances of a host would, so neither waits for the other's device work)
        t0 = time.perf_counter()
        core.Update([cs.ScanSegment(rays, core.Pose)])                   # :159-160
        t1 = time.perf_counter()
        hect.Update(hs.ScanCloud(xy), hect.MatchPose, loop < 10)         # :179
        t2 = time.perf_counter()
        if loop >= 10:                                                   # steady state only (first calls allocate)
            t_core += t1 - t0
            t_hect += t2 - t1
        row = {}
        for name, est in (("core", core.Pose), ("hector", hect.MatchPose)):
            d = float(math.hypot(est[0] - tp[0], est[1] - tp[1]))
            ang = float(abs(math.degrees(rad_diff(float(est[2]), float(tp[2])))))
            err[name].append((d, ang))
            row[name] = [round(d, 4), round(ang, 3)]
        if hector_lost_at is None and (err["hector"][-1][0] > 1.0 or err["hector"][-1][1] > 10.0):   # :184-196
            hector_lost_at = loop
        if loop % a.every == 0 or loop == len(traj) - 1:
            row["scan"] = loop
            curve.append(row)

    def stats(e):
        e = np.array(e)
        return {"rms_m": round(float(np.sqrt((e[:, 0] ** 2).mean())), 4), "max_m": round(float(e[:, 0].max()), 4),
                "rms_deg": round(float(np.sqrt((e[:, 1] ** 2).mean())), 3), "max_deg": round(float(e[:, 1].max()), 3),
                "final_m": round(float(e[-1, 0]), 4), "final_deg": round(float(e[-1, 1]), 3)}

    ext = hect.MapRep.Maps[0].GetMapExtends()
    n_t = max(len(traj) - 10, 1)
    out = {
        "scenario": "one lap (%.1f m, %d scans, %.2f m apart) around the inner obstacle of the default field" % (lap_len, len(traj), a.step),
        "config": {"rays": a.rays, "hole_map": a.hole_map, "obstacle_map": a.obstacle_map, "candidates_per_scan": a.iterations * a.threads, "heading_lattice": bool(a.lattice),
                   "hector": "%d^2 x %d levels" % (a.hector_side, a.hector_levels), "measure_error_m": sim.MEASURE_ERROR},
        "coreslam": stats(err["core"]), "hector": stats(err["hector"]), "hector_first_large_difference_at": hector_lost_at,
        "us_per_update": {"coreslam": round(t_core / n_t * 1e6, 1), "hector": round(t_hect / n_t * 1e6, 1)},
        "hector_map_extends_level0": list(ext),
        "holemap_nonreset_pixels": int((core.HoleMap.Pixels != 32750).sum()),
        "curve": curve,
    }
    print(json.dumps(out))
    hect.Dispose()
    core.Dispose()
    ctx_h.close()
    ctx.close()


if __name__ == "__main__":
    main()
