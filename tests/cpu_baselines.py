"""CPU baselines of SURVEY.md sec.8d on the host cores of this box (no GPU work): the ParallelWorker-structured port of
the reference search (oracle/cpu_baseline.c, kind "port" -- the C# reference cannot run here) at T = 1, 4 and nproc
(capped at 64, WaitHandle.WaitAll's limit), for config C1 (400^2 map, 360 rays, 1000 iterations per thread) and for the
headline workload's map and scan (2048^2, 1080 rays).  Prints one JSON object."""
import json, math, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import oracle_c as oc
import slam.net_amd.sim as sim

oc.set_trig_mode(oc.TRIG_DET)
out = {"cores_online": os.cpu_count()}
segs = sim.default_field()
for name, size, R, iters in (("C1_400_360", 400, 360, 1000), ("headline_2048_1080", 2048, 1080, 256)):
    scale = size / 40.0
    pix = np.full(size * size, 32750, np.uint16)
    rng = sim.PCG32(1234); traj = sim.trajectory(31)
    for p in traj[:-1]:
        _, xy = sim.make_scan(segs, p, R, rng)
        oc.update_holemap(pix, size, scale, xy, p, 0.6, 50)
    _, xy = sim.make_scan(segs, traj[-1], R, rng)
    base = (traj[-1] + np.array([0.03, -0.02, math.radians(1.0)], np.float32)).astype(np.float32)
    res = {}
    for T in sorted(set([1, 4, min(os.cpu_count() or 1, 64)])):
        offs = sim.gaussian_offsets(T * iters, 0.1, math.radians(10.0), seed=42)
        secs, evals, _, _ = oc.cpu_baseline_search(pix, size, scale, xy, base, offs, T, iters, 5)      # warm-up: 5 scans
        scans = max(int(4.0 / (secs / 5)), 100)                                                          # >= 100 scans, ~4 s
        secs, evals, bi, bd, per = oc.cpu_baseline_search_timed(pix, size, scale, xy, base, offs, T, iters, scans)
        per_scan_evals = evals / scans
        res["T%d" % T] = {"evals_per_s": evals / secs, "evals_per_s_median": per_scan_evals / float(np.median(per)),
                          "evals_per_s_p95_scan": per_scan_evals / float(np.percentile(per, 95)),
                          "scans": scans, "iterations_per_thread": iters, "seconds": secs}
    out[name] = res
print(json.dumps(out, indent=1))
