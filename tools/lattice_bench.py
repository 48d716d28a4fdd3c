"""K1 over a heading lattice (slamhip_cs_generate_offsets_lattice) against the same search over the plain device-generated list
and over an i.i.d. Gaussian list handed in by the host (bench.py's headline list): us per launch between two device
synchronisations, 200 enqueue-only searches each.  `python tools/lattice_bench.py [size rays]`; one JSON object."""
import json
import math
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import slam.net_amd.coreslam as cs
import slam.net_amd.sim as sim


def main():
    size = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
    rays = int(sys.argv[2]) if len(sys.argv) > 2 else 1080
    ctx = cs.Context(0)
    dev = cs.CoreSlamDevice(ctx, 40.0, size, max(size // 4, 1))
    segs = sim.default_field(); rng = sim.PCG32(1234); traj = sim.trajectory(31)
    for p in traj[:-1]:
        _, xy = sim.make_scan(segs, p, rays, rng); dev.set_scan(xy); dev.update_holemap(p, 0.6, 50)
    _, xy = sim.make_scan(segs, traj[-1], rays, rng)
    base = (traj[-1] + np.array([0.03, -0.02, math.radians(1.0)], np.float32)).astype(np.float32)
    dev.set_scan(xy)

    def t_search(K):
        for _ in range(20): dev.search_shard_enqueue(base, 0, K)
        ctx.synchronize()
        t0 = time.perf_counter()
        for _ in range(200): dev.search_shard_enqueue(base, 0, K)
        ctx.synchronize()
        return (time.perf_counter() - t0) / 200 * 1e6

    out = {"map": size, "rays": rays, "sigma_xy_m": 0.1, "sigma_theta_deg": 10.0, "us_per_search": {}}
    for K in (4001, 16384, 65536, 262144):
        row = {}
        dev.generate_offsets(K - 1, 0.1, math.radians(10.0), seed=1, stream=2)
        row["generated_stratified_headings"] = t_search(K)
        dev.generate_offsets(K - 1, 0.1, math.radians(10.0), seed=1, stream=2, lattice=True)
        row["generated_heading_lattice"] = t_search(K)
        dev.set_offsets(sim.gaussian_offsets(K - 1, 0.1, math.radians(10.0), seed=42))
        row["host_list_iid_gaussian"] = t_search(K)
        row["lattice_vs_stratified"] = row["generated_heading_lattice"] / row["generated_stratified_headings"]
        out["us_per_search"][str(K)] = {k: round(v, 3) for k, v in row.items()}
    assert dev.selfcheck_failures == 0
    print(json.dumps(out))


if __name__ == "__main__":
    main()
