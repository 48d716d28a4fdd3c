#!/usr/bin/env python3
"""Randomised parity soak (GPU + CPU oracle): random map sizes, poses (also at and beyond the map edge), ray counts, hole
widths, candidate lists and Hector pyramids; every integer output must equal the oracle's bit for bit, Hector match poses
within 1e-4.  Prints one line per case and a summary; exit code 1 on the first mismatch.

    python tests/fuzz_parity.py [--seconds 120 --seed 1]
"""
import argparse
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))          # test infrastructure: the checker

import numpy as np  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=120.0)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--dump", default="", help="npz file for the inputs and device outputs of a failing map case")
    a = ap.parse_args()
    import oracle_c as oc
    import slam.net_amd.coreslam as cs
    import slam.net_amd.hector as hs
    import slam.net_amd.sim as sim
    oc.set_trig_mode(oc.TRIG_DET)
    rng = np.random.default_rng(a.seed)
    segs = sim.default_field()
    ctx = cs.Context(0)
    t_end = time.time() + a.seconds
    n_cases = 0
    n_near = 0
    while time.time() < t_end:
        n_cases += 1
        kind = n_cases % 3 if n_cases % 16 else 3
        if kind == 3:
            # CoreSLAMProcessor.Update sequences: the state machine above the kernels (scan counter, odometry bookkeeping,
            # multi-segment clouds, empty scans, Reset) against the oracle's, estimate and maps after every scan
            size = int(rng.choice([64, 256, 400, 1024])); osize = int(rng.choice([16, 64, 100]))
            start = np.array([rng.uniform(12, 28), rng.uniform(12, 28), rng.uniform(-3, 3)], np.float32)
            K = int(rng.choice([1, 200, 1500])); thr = int(rng.choice([1, 4]))
            if os.environ.get("FUZZ_TRACE"): print("case %d kind 3: processor size %d/%d K %dx%d" % (n_cases, size, osize, K, thr), flush=True)
            proc = cs.CoreSLAMProcessor(40.0, size, osize, start, 0.1, 0.17, K, thr, ctx=ctx)
            ref = oc.CSProc(40.0, size, osize, start)
            hw = float(rng.choice([0.6, 2.0])); q = int(rng.choice([50, 200])); sb = int(rng.choice([0, 2, 5])); mh = int(rng.choice([10, 3]))
            proc.HoleWidth = hw; proc.Quality = q; proc.PositionSearchBeginning = sb; proc.MaxObstacleHits = mh
            ref.set_params(quality=q, hole_width=hw, search_beginning=sb, max_hits=mh)
            prng = sim.PCG32(int(rng.integers(1, 1 << 30)))
            ok = True; desc = "processor size %d/%d K %dx%d hw %.1f q %d begin %d" % (size, osize, K, thr, hw, q, sb)
            true = start.astype(np.float64).copy()
            for it in range(int(rng.integers(4, 14))):
                true += [rng.uniform(-0.1, 0.1), rng.uniform(-0.1, 0.1), rng.uniform(-0.03, 0.03)]
                R = int(rng.choice([0, 1, 90, 400])) if rng.random() < 0.3 else 360
                rays = sim.make_scan(segs, true.astype(np.float32), R, prng)[0] if R else np.zeros((0, 2), np.float32)
                est = proc.Pose.copy()
                nseg = int(rng.integers(1, 4)) if rays.shape[0] >= 3 else 1
                cuts = np.sort(rng.choice(np.arange(1, max(rays.shape[0], 2)), nseg - 1, replace=False)) if nseg > 1 else np.zeros(0, int)
                seg_start = np.concatenate([[0], cuts, [rays.shape[0]]]).astype(np.int32)
                odo = (est + np.array([rng.uniform(-0.05, 0.05), rng.uniform(-0.05, 0.05), rng.uniform(-0.02, 0.02)], np.float32)).astype(np.float32)
                seg_poses = np.stack([(odo + np.array([0.01 * (k - nseg + 1), 0.0, 0.002 * (k - nseg + 1)], np.float32)).astype(np.float32) for k in range(nseg)])
                offs = sim.gaussian_offsets(K * thr, 0.1, 0.17, seed=int(rng.integers(1, 1 << 30)))
                proc.SetOffsets(offs)
                proc.Update([cs.ScanSegment(rays[seg_start[k]:seg_start[k + 1]], seg_poses[k], k == nseg - 1) for k in range(nseg)])
                ref.update(seg_poses, seg_start, rays, offs)
                if not ((proc.Pose == ref.pose).all() and (proc.HoleMap.Pixels == ref.holemap).all() and (proc.ObstacleMap.Pixels.ravel() == ref.obstaclemap.ravel()).all()):
                    ok = False; desc += " || differs at scan %d: pose %s vs %s" % (it, proc.Pose, ref.pose); break
                if rng.random() < 0.08:
                    proc.Reset(); ref.reset()
            proc.Dispose(); ref.close()
        elif kind in (0, 1):
            size = int(rng.choice([64, 120, 256, 400, 513, 1024, 1536, 2048, 2048, 4096]))
            osize = int(rng.choice([16, 64, 100, 256]))
            R = int(rng.choice([1, 7, 90, 360, 1080, 1080, 2500, 4097]))
            hw = float(rng.choice([0.1, 0.6, 2.0, 5.0]))
            q = int(rng.choice([1, 50, 128, 255]))
            mh = int(rng.choice([10, 1, 127, -3]))
            phys = float(rng.choice([40.0, 40.0, 25.0, 100.0]))                 # pixels per metre = size / phys
            if os.environ.get("FUZZ_TRACE"):
                print("case %d kind %d: size %d/%d R %d hw %.1f q %d mh %d phys %.0f" % (n_cases, kind, size, osize, R, hw, q, mh, phys), flush=True)
            dev = cs.CoreSlamDevice(ctx, phys, size, osize)
            ref = np.full(size * size, 32750, np.uint16)
            oref = np.full(osize * osize, -5, np.int8)
            prng = sim.PCG32(int(rng.integers(1, 1 << 30)))
            pose = np.array([rng.uniform(4, 36), rng.uniform(4, 36), rng.uniform(-7, 7) * (30.0 if rng.random() < 0.1 else 1.0)], np.float32)
            if rng.random() < 0.15:
                pose[0] = rng.choice([-0.3, 0.0, 39.99, 40.2])         # robot at / beyond the map edge
            ok = True
            why = []
            trace = []
            for it in range(int(rng.integers(1, 5))):
                p = (pose + np.array([0.1 * it, -0.05 * it, 0.02 * it], np.float32)).astype(np.float32)
                inside = 5.5 < p[0] < 34.5 and 5.5 < p[1] < 34.5
                if inside:
                    rays, xy = sim.make_scan(segs, p, R, prng)
                else:
                    ang = rng.uniform(-np.pi, np.pi, R); rad = rng.uniform(0.01, 30.0, R)
                    xy = np.stack([rad * np.cos(ang), rad * np.sin(ang)], 1).astype(np.float32)
                if rng.random() < 0.3:
                    xy = xy[rng.permutation(xy.shape[0])]                # ray order matters for the blend
                if rng.random() < 0.1 and xy.shape[0] > 4:               # hostile points: far away, duplicates, at the robot
                    xy = xy.copy()
                    xy[0] = [3.0e4, -2.0e4]; xy[1] = xy[2]; xy[3] = [0.0, 0.0]
                    if rng.random() < 0.3: xy[4 % xy.shape[0]] = [np.nan, 1.0]
                if xy.shape[0] == 0:
                    continue
                trace.append((xy.copy(), p.copy()))
                dev.set_scan(xy)
                dev.update_holemap(p, hw, q)
                dev.update_obstaclemap(p, mh)
                n = oc.update_holemap(ref, size, dev.hole_scale, xy, p, hw, q)
                oc.update_obstaclemap(oref, osize, dev.obst_scale, xy, p, mh)
                if dev.last_holemap_pixels != n:
                    ok = False; why.append("pixel count %d vs %d at update %d" % (dev.last_holemap_pixels, n, it))
            if rng.random() < 0.25 and ok and not os.environ.get("FUZZ_NO_MIRROR"):
                # round 4: the asynchronous mirror -- requested now (everything is news to a fresh array), then once more after
                # another update of the last scan: both times the array equals a full download
                if os.environ.get("FUZZ_TRACE"): print("   mirror", flush=True)
                mir = np.zeros(size * size, np.uint16)
                dev.holemap_mirror_async(mir); dev.holemap_mirror_wait()
                okm = bool((mir == dev.holemap_download()).all())
                if trace:
                    dev.update_holemap(trace[-1][1], hw, q)
                    oc.update_holemap(ref, size, dev.hole_scale, trace[-1][0], trace[-1][1], hw, q)
                    dev.holemap_mirror_async(mir); dev.holemap_mirror_wait()
                    okm = okm and bool((mir == dev.holemap_download()).all())
                dev.holemap_mirror_release()
                if not okm:
                    ok = False; why.append("asynchronous mirror differs from a full download")
            got = dev.holemap_download()
            gob = dev.obstaclemap_download().ravel()
            if not (got == ref).all():
                bad = np.flatnonzero(got != ref)
                ok = False; why.append("holemap: %d pixels differ, first %s got %s want %s" % (bad.size, [(int(b) % size, int(b) // size) for b in bad[:4]], got[bad[:4]], ref[bad[:4]]))
            if not (gob == oref).all():
                bad = np.flatnonzero(gob != oref)
                ok = False; why.append("obstaclemap: %d cells differ, first %s got %s want %s" % (bad.size, [(int(b) % osize, int(b) // osize) for b in bad[:4]], gob[bad[:4]], oref[bad[:4]]))
            desc = "maps size %d/%d rays %d hw %.1f q %d pose %s" % (size, osize, R, hw, q, np.round(pose, 2))
            if ok and kind == 1 and xy.shape[0] > 0:
                K = int(rng.choice([1, 2, 300, 1023, 1024, 1025, 2049, 4096, 12289, 16384, 20000, 65536, 70000, 98304, 120000]))     # (65 536 and more: groups of 2048 candidates; up to 12 288: 512)
                sxy, sth = float(rng.choice([0.0, 0.02, 0.1, 0.5])), float(rng.choice([0.0, 0.01, 0.17, 0.8, 3.0]))
                base = (pose + np.array([0.03, -0.02, 0.017], np.float32)).astype(np.float32)
                if rng.random() < 0.5:
                    offs = sim.gaussian_offsets(K - 1, sxy, sth, seed=int(rng.integers(1, 1 << 30)))
                    dev.set_offsets(offs)
                    gp, gd, gi = dev.search(base)
                    fused = False
                else:
                    lattice = rng.random() < 0.5                      # (round 4: the opt-in heading lattice -- the search kernel's LAT variant from 12 289 candidates on)
                    dev.generate_offsets(K - 1, sxy, sth, seed=int(rng.integers(1, 1 << 30)), stream=n_cases, lattice=lattice)
                    fused = rng.random() < 0.5
                    gp, gd, gi = dev.search_and_update(base, hw, q, mh)[:3] if fused else dev.search(base)
                    offs = dev.offsets_download()
                rbi, rpose, rbd, rall = oc.search(got, size, dev.hole_scale, xy, base, offs)
                ok = gi == rbi and gd == rbd and bool((np.asarray(gp)[:2] == rpose[:2]).all())
                if ok and not fused and rng.random() < 0.5:
                    # round 4: the enqueue-only search into the handle's result ring, three times in a row -- the second launch
                    # under one layout makes the ray ranges' cost cuts, every launch must find its result word rested
                    if os.environ.get("FUZZ_TRACE"): print("   ring K %d" % K, flush=True)
                    slots = [dev.search_shard_enqueue(base, 0, K) for _ in range(3)]      # (round 6: with SLAMHIP_K1_PLAN_ALWAYS=1 each of these races its plan launch)
                    keys = [dev.key_read(sl) for sl in slots]
                    ok = all(k == ((int(rbd) << 32) | int(rbi)) for k in keys)
                    if not ok: why.append("ring search: keys %s, oracle idx %d dist %d" % (keys, rbi, rbd))
                if not ok:
                    why.append("search: got idx %d dist %d pose %s, oracle idx %d dist %d pose %s" % (gi, gd, gp, rbi, rbd, rpose))
                    if a.dump:
                        poses_ = np.vstack([base[None], base[None] + offs]).astype(np.float32)
                        dd_ = dev.distance_poses(poses_)[0] if K <= 200000 else np.zeros(0, np.int32)
                        np.savez(a.dump + ".search.npz", size=size, xy=xy, base=base, offs=offs, got=got, rall=rall, dd=dd_, gi=gi, gd=gd, fused=fused)
                if ok and fused:
                    # the fused call also drew both maps from the winner's pose, theta normalised (:746-751)
                    wp = np.array([rpose[0], rpose[1], oc.normalize_angle(float(rpose[2]))], np.float32)
                    ok = bool((np.asarray(gp) == wp).all())
                    n = oc.update_holemap(ref, size, dev.hole_scale, xy, wp, hw, q)
                    oc.update_obstaclemap(oref, osize, dev.obst_scale, xy, wp, mh)
                    ok = ok and dev.last_holemap_pixels == n and bool((dev.holemap_download() == ref).all()) \
                        and bool((dev.obstaclemap_download().ravel() == oref).all())
                    if not ok: why.append("fused update: maps or pose differ (pose %s vs %s)" % (gp, wp))
                elif ok and K <= 4096:
                    # every candidate's distance through the explicit-pose entry points, with a few hostile poses mixed in
                    poses = np.vstack([base[None], base[None] + offs]).astype(np.float32)
                    dd, bi, bd = dev.distance_poses(poses)
                    ok = bool((dd == rall).all()) and bi == rbi and bd == rbd
                    bad_poses = poses.copy()
                    bad_poses[1 % K] = [np.nan, 1.0, 0.0]; bad_poses[K // 2] = [1e30, -1e30, 3.0]; bad_poses[K - 1] = [20.0, 20.0, np.inf]
                    dd2 = dev.distance_poses(bad_poses)[0]
                    keep = np.ones(K, bool); keep[[1 % K, K // 2, K - 1]] = False
                    ok = ok and bool((dd2[keep] == rall[keep]).all())
                    pxcs = np.stack([oc.pose_to_pxcs(pp, dev.hole_scale) for pp in poses[:256]])
                    dd3 = dev.distance_pxcs(pxcs)[0]
                    ok = ok and bool((dd3 == rall[:len(pxcs)]).all())
                    if not ok: why.append("distance_poses / distance_pxcs differ")
                ok = ok and dev.selfcheck_failures == 0
                desc += " | search K %d sigma %.2f/%.2f -> idx %d dist %d" % (K, sxy, sth, gi, gd)
            if why: desc += " || " + "; ".join(why)
            if not ok and a.dump:
                np.savez(a.dump, size=size, osize=osize, hw=hw, q=q, got=got, gob=gob, n_updates=len(trace),
                         **{"xy%d" % i: t[0] for i, t in enumerate(trace)}, **{"p%d" % i: t[1] for i, t in enumerate(trace)})
            dev.close()
        else:
            side = int(rng.choice([64, 200, 401, 1024, 2048]))
            levels = int(rng.choice([1, 2, 3, 4]))
            cell = 40.0 / side
            R = int(rng.choice([8, 180, 1080, 3000]))
            side_h = side if rng.random() < 0.7 else int(rng.choice([side // 2 + 3, side + 37]))      # rectangular maps too
            if os.environ.get("FUZZ_TRACE"): print("case %d kind 2: hector %dx%d levels %d R %d" % (n_cases, side, side_h, levels, R), flush=True)
            rep = hs.MapRepMultiMap(cell, (side, side_h), levels, ctx=ctx)
            ref = oc.make_pyramid(cell, side, side_h, levels)
            ff, fo = 0.4, 0.9
            if rng.random() < 0.2:
                ff, fo = float(rng.choice([0.3, 0.45])), float(rng.choice([0.6, 0.8, 0.95]))
                rep.SetUpdateFactorFree(ff); rep.SetUpdateFactorOccupied(fo)
                for g in ref:
                    g.set_factors(ff, fo)
            prng = sim.PCG32(int(rng.integers(1, 1 << 30)))
            pose = np.array([rng.uniform(8, 32), rng.uniform(8, 32), rng.uniform(-3, 3)], np.float32)
            ok = True
            htrace = []
            for it in range(int(rng.integers(2, 6))):
                p = (pose + np.array([0.06 * it, 0.03 * it, 0.01 * it], np.float32)).astype(np.float32)
                rays, xy = sim.make_scan(segs, p, R, prng)
                if xy.shape[0] == 0:
                    continue
                if rng.random() < 0.1 and xy.shape[0] > 5:             # hostile points: far away, duplicates, at the origin, NaN
                    xy = xy.copy()
                    xy[0] = [3.0e4, -2.0e4]; xy[1] = xy[2]; xy[3] = [0.0, 0.0]; xy[4] = [np.nan, 1.0]
                org = (float(rng.uniform(-0.5, 0.5)), float(rng.uniform(-0.5, 0.5))) if rng.random() < 0.2 else (0.0, 0.0)
                htrace.append((xy.copy(), np.concatenate([p, np.array(org, np.float32)])))
                rep.UpdateByScan(hs.ScanCloud(xy, (org[0], org[1], 0.0)), p)
                for g in ref:
                    g.update_by_scan(xy, p, origin=org)
            for l in range(levels):
                c = rep.Maps[l].GetCells()
                ok = ok and bool((c["update_index"] == ref[l].cells["update_index"]).all() and (c["value"] == ref[l].cells["value"]).all())
                ok = ok and rep.Maps[l].GetMapExtends() == ref[l].map_extends()
            hint = (p + np.array([0.05, -0.04, 0.02], np.float32)).astype(np.float32)
            m = hs.ScanMatcher(4).MatchData(rep, hs.ScanCloud(xy), hint)
            # the batched form: up to 8 hints run the single match's kernel (the same floats), larger batches 256 lanes per hint
            mb3 = hs.ScanMatcher(4).MatchDataBatch(rep, hs.ScanCloud(xy), np.stack([hint] * 3))
            batch_same = bool(all((np.asarray(mb3[i]) == np.asarray(m)).all() or (np.isnan(np.asarray(m)).any() and np.isnan(np.asarray(mb3[i])).any()) for i in range(3)))
            mb12 = np.asarray(hs.ScanMatcher(4).MatchDataBatch(rep, hs.ScanCloud(xy), np.stack([hint] * 12)))
            batch_same = batch_same and bool((mb12 == mb12[0][None]).all() or np.isnan(mb12).any())
            w = oc.match_pyramid(ref, xy, hint, [3] * levels, 4)
            cells_ok = ok
            # (a handful of rays gives a near-singular Hessian: the 1e-7 differences between summation orders -- the
            # reference's own result depends on its thread count there -- are amplified without bound, so the match is
            # only compared for scans that constrain the pose)
            # The reference sums its fp32 terms thread chunk by thread chunk, so its own result moves with the thread
            # count; that spread (oracle at 1 and 4 threads) is the floor of any comparison and is added to the tolerance.
            # A match that runs away from its hint (coarse grids, few iterations) amplifies those differences from
            # iteration to iteration and is not compared either.
            near0 = n_near
            if R >= 180 and math.hypot(w[0] - hint[0], w[1] - hint[1]) < 0.5 and abs(w[2] - hint[2]) < 0.1:
                w1 = oc.match_pyramid(ref, xy, hint, [3] * levels, 1)
                tol = 1e-4 + 10.0 * np.abs(w1 - w)
                def accepted(res, scale=1.0):
                    """res against the oracle's w: directly, or as what the reference arithmetic gives at another thread count / for a
                    hint a digit or two away (returns (ok, needed_the_neighbourhood))"""
                    res = np.asarray(res)
                    if bool(np.all(np.abs(res - w) < scale * tol)):
                        return True, False
                    # The interpolation takes floor() of the map coordinates and tests them against the map bounds
                    # (ScanMatcher.cs:216-225): on a sparse map the result is a discontinuous function of the pose, and a
                    # last-digit difference in an intermediate estimate can move a point into the neighbouring cell and the
                    # answer by millimetres.  The device result must then be what the reference arithmetic gives for a hint
                    # a digit or two away.
                    # the reference's own thread counts first: another chunking of the fp32 sums is enough to flip it
                    alt = np.array([oc.match_pyramid(ref, xy, hint, [3] * levels, T) for T in (2, 3, 5, 8, 16, 32, 64)])      # (up to the 64 threads WaitHandle.WaitAll allows the reference)
                    good = bool(np.any(np.all(np.abs(alt - res[None]) < scale * tol, axis=1)))
                    prng2 = np.random.default_rng(n_cases)
                    # (on coarse grids the intermediate estimates differ by up to ~1e-5: three scales of perturbation)
                    outs = np.array([oc.match_pyramid(ref, xy, (hint * (1.0 + prng2.uniform(-3e-7, 3e-7, 3)) + prng2.uniform(-sc, sc, 3)).astype(np.float32), [3] * levels, 4)
                                     for sc in (1e-6, 1e-5, 3e-5) for _ in range(40)])
                    good = good or bool(np.any(np.all(np.abs(outs - res[None]) < scale * tol, axis=1)))
                    if not good:
                        # many different answers in that neighbourhood (a chaotic case): inside their envelope is all one can ask
                        # (or the reference's own answers over that cloud spread by more than the tolerance)
                        distinct = len({tuple(np.round(o, 5)) for o in outs})
                        spread = outs.max(0) - outs.min(0)
                        pad = scale * tol + (spread if bool(np.any(spread > 1e-4)) else 0.0)
                        good = (distinct >= 8 or bool(np.any(spread > 1e-4))) and \
                            bool(np.all(res > outs.min(0) - pad) and np.all(res < outs.max(0) + pad))
                    return good, True
                close, nb = accepted(m)
                if nb:
                    n_near += 1
                ok = ok and close
                if close and n_near == near0:                        # (a well-conditioned match: the other summation order of the large batch agrees as well --
                    # directly, or, like the single match above, as the reference's own answer a digit away: round 6, seed 6102, a 401-cell
                    # 4-level pyramid whose single match met the oracle to 2e-5 while the 256-lane batch kernel's sums sent a point across a cell border)
                    close12, nb12 = accepted(mb12[0], 2.0)
                    ok = ok and close12
                    if nb12:
                        n_near += 1
            ok = ok and batch_same
            desc = "hector side %d levels %d rays %d pose %s" % (side, levels, R, np.round(pose, 2))
            if not ok:
                desc += " | cells equal: %s, match %s vs oracle %s (hint %s), batches consistent: %s, batch of 12: %s" % (cells_ok, np.asarray(m), w, hint, batch_same, mb12[0])
                if a.dump:
                    np.savez(a.dump, side=side, side_h=side_h, ff=ff, fo=fo, levels=levels, cell=cell, hint=hint, m=np.asarray(m), w=w, n_updates=len(htrace),
                             **{"xy%d" % i: t[0] for i, t in enumerate(htrace)}, **{"p%d" % i: t[1] for i, t in enumerate(htrace)})
            rep.close()
        print(("ok   " if ok else "FAIL ") + desc, flush=True)
        if not ok:
            print("MISMATCH after %d cases (seed %d)" % (n_cases, a.seed))
            sys.exit(1)
    print("fuzz: %d cases, all equal to the oracle (seed %d); %d Hector matches equal to the oracle's for a hint one or two digits away" % (n_cases, a.seed, n_near))
    ctx.close()


if __name__ == "__main__":
    main()
