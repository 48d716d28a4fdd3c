"""How much does the platform's MathF.Cos / MathF.Sin matter?  (CPU only; uses the oracle, so this is test infrastructure.)

The reference forms c, s with the C runtime's cosf / sinf (CoreSLAM/CoreSLAMProcessor.cs:234-235), which is not bit-reproducible
across platforms; libslamhip's pose-taking entry points use a deterministic, correctly rounded sin / cos (csrc/det_trig.h) and the
(px, py, c, s)-taking ones leave trigonometry to the caller.  This script runs the oracle's Monte-Carlo search on the headline
workload (2048^2 HoleMap, 1080 rays, 16 384 candidates) over many scans twice -- ORACLE_TRIG_DET and ORACLE_TRIG_LIBM (glibc) --
and reports how many per-candidate distances differ and how often the arg-min moves.

usage: python tools/trig_sensitivity.py [--scans 1000] [--out profiles/r03_trig_sensitivity.json]
"""
import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np  # noqa: E402
import oracle_c as oc  # noqa: E402
import slam.net_amd.sim as sim  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--scans", type=int, default=1000)
ap.add_argument("--size", type=int, default=2048)
ap.add_argument("--rays", type=int, default=1080)
ap.add_argument("--cands", type=int, default=16384)
ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "r03_trig_sensitivity.json"))
a = ap.parse_args()

size, R, K = a.size, a.rays, a.cands
scale = size / 40.0
segs = sim.default_field()
rng = sim.PCG32(1234)
pix = np.full(size * size, 32750, np.uint16)
oc.set_trig_mode(oc.TRIG_DET)
lap, _ = sim.lap_trajectory(None, 0.1)
for p in lap[::8]:                                     # the map: one mapping update every 0.8 m of the lap, at the true poses
    _, xy = sim.make_scan(segs, p, R, rng)
    oc.update_holemap(pix, size, scale, xy, p, 0.6, 50)

# trig itself: how many of the candidates' angles get a different c or s
t0 = time.time()
n_flip = n_dist_diff = n_dist = 0
max_abs = 0
max_rel = 0.0
winner_gap = []           # when the arg-min moves: |distance(det winner) - distance(libm winner)| evaluated in ONE mode (det)
pose_gap = []             # ... and how far apart the two winning poses are (m, rad)
per_scan_frac = []
for i in range(a.scans):
    true_pose = lap[(7 * i) % len(lap)]
    _, xy = sim.make_scan(segs, true_pose, R, rng)
    base = (np.asarray(true_pose, np.float32) + np.array([0.03, -0.02, math.radians(1.0)], np.float32)).astype(np.float32)
    offs = sim.gaussian_offsets(K - 1, 0.1, math.radians(10.0), seed=1000 + i)
    oc.set_trig_mode(oc.TRIG_DET)
    bi_d, pose_d, bd_d, all_d = oc.search(pix, size, scale, xy, base, offs)
    oc.set_trig_mode(oc.TRIG_LIBM)
    bi_l, pose_l, bd_l, all_l = oc.search(pix, size, scale, xy, base, offs)
    diff = all_d != all_l
    nd = int(diff.sum())
    n_dist_diff += nd
    n_dist += K
    per_scan_frac.append(nd / K)
    if nd:
        dd = np.abs(all_d.astype(np.int64) - all_l.astype(np.int64))[diff]
        max_abs = max(max_abs, int(dd.max()))
        max_rel = max(max_rel, float((dd / np.maximum(all_d[diff].astype(np.float64), 1.0)).max()))
    if bi_d != bi_l:
        n_flip += 1
        winner_gap.append(int(abs(int(all_d[bi_d]) - int(all_d[bi_l]))))
        pose_gap.append([float(math.hypot(pose_d[0] - pose_l[0], pose_d[1] - pose_l[1])), float(abs(pose_d[2] - pose_l[2]))])
oc.set_trig_mode(oc.TRIG_DET)
out = {
    "what": "oracle Monte-Carlo search, deterministic correctly-rounded sin/cos (ORACLE_TRIG_DET, = libslamhip's pose entry points) vs glibc cosf/sinf (ORACLE_TRIG_LIBM, "
            "standing in for a platform CRT behind MathF.Cos/Sin, CoreSLAMProcessor.cs:234-235)",
    "workload": {"map": size, "rays": R, "candidates": K, "scans": a.scans, "sigma_xy_m": 0.1, "sigma_theta_deg": 10.0,
                 "poses": "every 7th pose of one lap around the inner obstacle (sim.lap_trajectory), a new noisy scan and a new candidate list per scan",
                 "map_built_by": "oracle mapping updates at every 8th true pose of the lap (%d updates)" % len(lap[::8])},
    "distances_compared": n_dist, "distances_different": n_dist_diff, "fraction_of_distances_different": n_dist_diff / max(n_dist, 1),
    "per_scan_fraction_different": {"median": float(np.median(per_scan_frac)), "max": float(np.max(per_scan_frac))},
    "max_abs_distance_difference": max_abs, "max_relative_distance_difference": max_rel,
    "argmin_flips": n_flip, "argmin_flip_rate": n_flip / max(a.scans, 1),
    "when_flipped": {"distance_gap_between_the_two_winners_det_mode": {"median": float(np.median(winner_gap)) if winner_gap else None, "max": max(winner_gap) if winner_gap else None},
                     "pose_gap_m_rad_max": [max(g[0] for g in pose_gap), max(g[1] for g in pose_gap)] if pose_gap else None},
    "seconds": round(time.time() - t0, 1),
}
print(json.dumps(out, indent=1))
if a.out:
    with open(a.out, "w") as f:
        json.dump(out, f, indent=1)
