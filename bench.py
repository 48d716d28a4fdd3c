#!/usr/bin/env python3
"""bench.py -- headline benchmark: candidate-pose distance evaluations per second.

Workload (BASELINE.json `metric` / north_star): CoreSLAM Monte-Carlo search on a 2048^2 HoleMap with a
1080-ray scan, --cands candidate poses per GPU per step (default 16384 = BASELINE.json configs[1]/[2]).
One "step" = one full search over this rank's shard of the flat candidate list = ONE launch of K1 (k1_search_tiled: batched
distance over all rays from LDS-staged HoleMap tiles, per-candidate accumulation and arg-min; the packed key of a step lands in a
result word owned by the handle, slamhip_cs_search_shard_enqueue) accompanied by ONE small launch on a stream of its own that
makes the search's plan (k1_plan: the candidate transform pose + jitter -> px,py,c,s with deterministic trigonometry, once per
candidate, and every workgroup's tile steps) -- both inside the timed region, every step.  With N > 1 GPUs the per-rank packed
(distance << 32 | index) keys are min-all-reduced over RCCL, enqueued behind the search (the keys of 16 steps per collective).  All
inputs (map, scan, jitter list) are resident in HBM before the timed region.  Weak scaling: per-GPU candidates are fixed as N grows.

`value` is the SAME form at every N -- the enqueue-only search, nothing returns to the host per step -- so that a 1 -> N curve
measures the exchange, not a change of form.  The BLOCKING per-scan form (K1, the collective, the reduced key on the host before
the next step: what a SLAM loop pays, CoreSLAMProcessor.cs:732 -> :750) is measured after the timed region at every N, N = 1
included, and reported beside it (`config.per_scan_blocking_us_per_step`, `multi_gpu.per_scan_blocking_us_per_step`).

Timed region: barrier + device synchronise, t0, K x (one C call: the search's two launches [+ the collective's call]), one event
record, ONE device synchronise, t1 -- event creation, the first event's record and every other synchronisation sit outside.

Launch:  python bench.py [--gpus N --steps K --warmup W]
         N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
                --master-port P bench.py --gpus N --steps K --warmup W
Rank 0 prints ONE JSON line.
"""
import argparse
import gc
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0      # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--size", type=int, default=2048, help="HoleMap side in pixels")
    ap.add_argument("--rays", type=int, default=1080)
    ap.add_argument("--cands", type=int, default=16384, help="candidate poses per GPU per step")
    ap.add_argument("--map-updates", type=int, default=30)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=30.0)
    ap.add_argument("--no-kernel-timing", action="store_true", help="no HIP-event timing of K1 (no roofline object)")
    ap.add_argument("--clock-warmup", type=int, default=3000,
                    help="untimed steps in front of the W warm-up steps that bring the device to its sustained clocks (0: none)")
    ap.add_argument("--no-extras", action="store_true", help="skip the other BASELINE.json configurations measured after the timed region")
    return ap.parse_args()


def main():
    a = parse()
    import torch
    import torch.distributed as dist
    import slam.net_amd.capi as capi
    import slam.net_amd.coreslam as cs
    import slam.net_amd.sim as sim
    import slam.net_amd.distributed as D

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        if rank == 0:
            print("bench.py: --gpus %d but WORLD_SIZE=%d (launch with torch.distributed.run)" % (a.gpus, world), file=sys.stderr)
        sys.exit(2)
    # SLAMHIP_BENCH_BACKEND=gloo (tests only): the N > 1 flow on a box with fewer GPUs than ranks -- ranks share devices and
    # the key travels over gloo instead of RCCL.  The driver's runs use the default: one GPU per rank, RCCL over xGMI.
    backend = os.environ.get("SLAMHIP_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local = local % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend=backend)

    # ---- synthetic world: default field, 30 mapping updates, scan 31 -----------------------
    ctx = cs.Context(local)
    dev = cs.CoreSlamDevice(ctx, 40.0, a.size, max(a.size // 4, 1))
    segs = sim.default_field()
    rng = sim.PCG32(1234)
    traj = sim.trajectory(a.map_updates + 1)
    for p in traj[:-1]:
        _, xy = sim.make_scan(segs, p, a.rays, rng)
        dev.set_scan(xy)
        dev.update_holemap(p, 0.6, 50)
    true_pose = traj[-1]
    _, xy = sim.make_scan(segs, true_pose, a.rays, rng)
    assert xy.shape[0] == a.rays, "every ray must hit (closed field)"
    base = (true_pose + np.array([0.03, -0.02, math.radians(1.0)], np.float32)).astype(np.float32)
    K_total = a.cands * world                          # flat candidates, index 0 = un-jittered pose
    offs = sim.gaussian_offsets(K_total - 1, 0.1, math.radians(10.0), seed=42)
    dev.set_scan(xy)
    dev.set_offsets(offs)
    first, count = D.shard_range(rank, world, K_total)            # contiguous block of the flat list per rank

    key = torch.full((1,), -1, dtype=torch.int64, device="cuda")
    ext = torch.cuda.ExternalStream(ctx.stream, device=torch.device("cuda", local))
    if world > 1:
        torch.cuda.set_stream(ext)          # the collectives are enqueued on the library's stream, behind the search
    # the per-step host path, bound once: one C call (the search launch) and, with N > 1, one collective
    import ctypes as C
    search_fn = capi.lib().slamhip_cs_search_shard_async
    search_args = (dev._h, capi.fptr(base), int(first), int(count), C.c_void_p(key.data_ptr()))
    assert key.dtype == torch.int64 and key.numel() == 1           # packed (distance << 32 | index): MIN on int64 is exact

    # N = 1: the enqueue-only search whose result word the handle owns (slamhip_cs_search_shard_enqueue: no caller memory, so the
    # kernel needs no final arriver -- the end of the launch is the completion); the last step's word is read after the region
    ring_fn = capi.lib().slamhip_cs_search_shard_enqueue
    ring_slot = C.c_void_p()
    ring_args = (dev._h, capi.fptr(base), int(first), int(count), C.byref(ring_slot))

    def step_ring():
        capi.check(ring_fn(*ring_args))

    def step_torch():
        capi.check(search_fn(*search_args))
        if world > 1:
            dist.all_reduce(key, op=dist.ReduceOp.MIN)             # one 8-byte RCCL min all-reduce per step

    def step_torch_per_scan():                                     # ... and the reduced key on the host before the next step
        step_torch()
        return int(key.item())

    # N > 1: the library's own communicator (slamhip_comm_*).  It is checked against the torch.distributed path on this very
    # workload first -- the blocking per-scan call and the asynchronous batched one; any failure or disagreement on any rank
    # falls back to torch.distributed (SLAMHIP_BENCH_COLLECTIVE=torch forces it; "lib1" exercises the library path on one rank).
    coll = os.environ.get("SLAMHIP_BENCH_COLLECTIVE", "lib")
    comm = None
    collective = "none"
    collective_ranks = 1
    if (world > 1 and backend == "nccl" and coll == "lib") or (world == 1 and coll == "lib1"):
        ok = 1
        try:
            comm = D.LibComm(ctx, rank, world)
            k_sync = comm.search_allreduce(dev, base, first, count)
            lib_step = comm.bind_step(dev, base, first, count)
            lib_step()
            k_lib = comm.wait()
            step_torch()
            ctx.synchronize(); torch.cuda.synchronize()
            ok = int(k_lib == int(key.item()) and k_sync == k_lib)
            collective_ranks = comm.info()[1]
        except Exception as e:                                     # noqa: BLE001 -- any failure means "use the torch path"
            print("bench.py: library communicator unavailable on rank %d (%s); using torch.distributed" % (rank, e), file=sys.stderr)
            ok = 0
        if world > 1:
            okt = torch.tensor([ok], dtype=torch.int32, device="cuda")
            dist.all_reduce(okt, op=dist.ReduceOp.MIN)
            ok = int(okt.item())
        if not ok and comm is not None:
            comm.close(); comm = None
    # ONE form of a step at every N: the enqueue-only search -- nothing returns to the host per step.  N = 1: the ring search; N > 1:
    # the same search launch + the key's min all-reduce enqueued behind it (library path: the keys of 16 steps per ncclAllReduce on the
    # communicator's own stream; torch.distributed path: one all_reduce per step on the search's stream).  The BLOCKING per-scan form
    # -- K1, the collective and the reduced key back on the host before the next step, what a SLAM loop pays per scan
    # (CoreSLAMProcessor.cs:732 -> :750) -- is measured after the timed region at every N, N = 1 included (`per_scan_blocking_us_per_step`).
    if comm is not None:
        comm.set_batch(16)
        step = lib_step
        step_blocking = comm.bind_search_allreduce(dev, base, first, count)
        collective = ("rccl ncclAllReduce(min, uint64, 16): the keys of 16 steps per collective, issued by libslamhip on the communicator's own stream "
                      "behind one event (slamhip_cs_search_allreduce_async); nothing returns to the host per step")
    elif world > 1:
        step = step_torch
        step_blocking = step_torch_per_scan
        collective = "%s all_reduce(min, 8 B) per step via torch.distributed on the search's stream, nothing read back per step" % ("rccl" if backend == "nccl" else backend)
        collective_ranks = dist.get_world_size()
    else:
        step = step_ring if os.environ.get("SLAMHIP_BENCH_N1_STEP", "ring") == "ring" else step_torch

        def step_blocking():
            return dev.search_shard(base, first, count)

    def sync_all():
        ctx.synchronize()
        if comm is not None:
            comm.synchronize()
        torch.cuda.synchronize()

    final_key = None
    # The device's clocks: a launch of 19 us after a pause runs at the clocks of an idle chip -- the governor needs tens of
    # milliseconds of sustained work to reach what a scan loop that runs for seconds sees (measured on MI355X: 19.2 us per step after 6
    # launches, 18.9 after 20, 17.7 after 2000; 262 144 candidates: 125.8 / 119.4 / 114.9 us).  `value` is a throughput, so the
    # timed region is measured at sustained clocks: `--clock-warmup` untimed steps (default 3000, ~60 ms) run in front of the W
    # warm-up steps; the same K steps measured BEFORE them, cold, are reported beside it (config.cold_clocks).
    cold = None
    # (the interpreter's cyclic garbage collector stays out of the timed regions: with torch imported a full collection is a
    # pause of tens of milliseconds -- it once landed in a loop of 100 blocking calls and read as 440 us per call instead of 70;
    # collected HERE, before the clocks are brought up: a pause of that length lets them fall again)
    gc.collect()
    gc.disable()
    if a.clock_warmup > 0:
        for _ in range(max(a.warmup, 1)):
            step()
        sync_all()
        if world > 1:
            dist.barrier()
        tc = time.perf_counter()
        for _ in range(a.steps):
            step()
        sync_all()
        if world > 1:
            dist.barrier()
        cold = (time.perf_counter() - tc) / a.steps
        for _ in range(a.clock_warmup):
            step()
        sync_all()
        # (two milliseconds for the HIP runtime's own housekeeping: right behind thousands of launches its launch calls take 2 - 5 x as long
        # for the next ~25 calls -- measured with the search's two launches per step: the first 20-step region behind the warm-up 19 - 24 us
        # per step, every later one, or the first one behind this pause, 17.2 - 17.6; short enough for the clocks to stay up: tens of
        # milliseconds let them fall)
        time.sleep(0.002)
    for _ in range(max(a.warmup, 1)):
        final_key = step()
    sync_all()
    # K1's average launch duration for the roofline figure.  At N = 1 a step is exactly one K1 launch, so two HIP events on the
    # operator's stream around the timed region give it (per-launch event pairs cost ~8 us per step on this stack and would
    # depress `value`); the first is recorded BEFORE t0.  At N > 1 the timed region's stream also carries the collective and the
    # hand-over, so K1 is timed in the overlapped pass after the region (library path: the operator's stream carries K1 only)
    # or per launch in a short pass of its own (torch.distributed path).
    two_events = (world == 1 or comm is not None) and not a.no_kernel_timing
    ev0 = torch.cuda.Event(enable_timing=True)
    ev1 = torch.cuda.Event(enable_timing=True)
    if world > 1:
        dist.barrier()
    sync_all()
    if two_events:
        ev0.record(ext)
    t0 = time.perf_counter()
    step_ts = None
    if os.environ.get("SLAMHIP_BENCH_STEP_TIMES"):                 # developer aid: the host's time per step inside the timed region (printed after it)
        step_ts = [t0]
        for _ in range(a.steps):
            final_key = step()
            step_ts.append(time.perf_counter())
    else:
        for _ in range(a.steps):
            final_key = step()
    if two_events:
        ev1.record(ext)
    if comm is not None:
        comm.synchronize()                  # (the keys of the last, partial batch are exchanged inside the region too)
    torch.cuda.synchronize()                # (the whole device: the library's streams included)
    if world > 1:
        dist.barrier()
        torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    gc.enable()
    if step_ts is not None:
        dts = np.diff(step_ts) * 1e6
        print("bench.py host us per step: mean %.2f p50 %.2f p90 %.2f max %.2f; enqueue done after %.1f us, the region's end (event, synchronise) took %.1f us; %s"
              % (dts.mean(), np.median(dts), np.percentile(dts, 90), dts.max(), (step_ts[-1] - t0) * 1e6, (t0 + elapsed - step_ts[-1]) * 1e6,
                 ("all steps: %s" % np.round(dts, 1)) if a.steps <= 40 else ("the ten longest: %s" % np.round(np.sort(dts)[-10:], 1))), file=sys.stderr)
    if comm is not None:
        final_key = comm.wait()                                    # (flushes the last batch of keys; the reduced key of the last step)
    else:
        final_key = dev.key_read(ring_slot.value) if step is step_ring else int(key.item())
    k1_ms, k1_n = (0.0, 0)
    k1_how = "two HIP events on the operator's stream around the timed region"
    if two_events:
        k1_ms, k1_n = ev0.elapsed_time(ev1), a.steps
    # ---- after the timed region: the blocking per-scan form, at every N ----------------------------------------------
    n_ps = max(min(a.steps, 100), 20)
    for _ in range(3):
        step_blocking()
    sync_all()
    if world > 1:
        dist.barrier()
    gc.disable()
    tb = time.perf_counter()
    for _ in range(n_ps):
        k_blocking = step_blocking()
    dt_ps = time.perf_counter() - tb
    gc.enable()
    sync_all()
    per_scan_us = dt_ps / n_ps * 1e6
    blocking_key_ok = bool(int(k_blocking) == int(final_key))
    # ---- ... and K1 as a per-scan flow sees it: a NEW scan in front of every search (what CoreSLAMProcessor.Update launches: no layout
    # or cut made for this very scan -- the previous scan's are taken over when they are legal -- and a plan whose inputs changed).
    # Eight scans from poses a few centimetres apart, 48 blocking searches, K1's duration from per-launch HIP event pairs (the
    # library's own timers); beside it the same pairs around repeated searches of ONE scan (the headline's launch, same instrument).
    first_search = None
    if world == 1 and not a.no_kernel_timing:
        try:
            rng_f = sim.PCG32(777)
            scans_f = []
            for k in range(8):
                p_f = (true_pose + np.array([0.02 * k, 0.01 * k, 0.002 * k], np.float32)).astype(np.float32)
                scans_f.append(sim.make_scan(segs, p_f, a.rays, rng_f)[1])

            def k1_events(new_scan_every_step, n):
                for k in range(6):
                    if new_scan_every_step:
                        dev.set_scan(scans_f[k % 8])
                    dev.search_shard(base, first, count)
                ctx.timing_reset(); ctx.timing_enable(1 << capi.K_CS_DISTANCE)
                for k in range(n):
                    if new_scan_every_step:
                        dev.set_scan(scans_f[k % 8])
                    dev.search_shard(base, first, count)
                ms_f, n_f = ctx.timing_get(capi.K_CS_DISTANCE)
                ctx.timing_enable(0)
                return ms_f / max(n_f, 1) * 1e3
            us_new = k1_events(True, 48)
            dev.set_scan(xy)
            us_same = k1_events(False, 48)
            first_search = {"first_search_of_new_scan_us": round(us_new, 3), "repeated_search_of_one_scan_us_same_instrument": round(us_same, 3),
                            "how": "per-launch HIP event pairs around K1 (slamhip_ctx_timing_*), 48 blocking searches each; a new scan = slamhip_cs_set_scan of one of eight "
                                   "scans cast a few centimetres apart in front of every search"}
        except Exception as e:                                     # noqa: BLE001 -- an extra must never cost the line
            first_search = {"error": repr(e)}
        dev.set_scan(xy)
    multi = None
    if world > 1 or comm is not None:
        # ---- ... and, N > 1: this rank's search alone in the SAME form as the timed region (no collective: what N = 1 runs), K1's launch time where the
        # region could not give it, the bare collective ----------------------
        n_ov = max(a.steps, 100)
        for _ in range(5):
            step_ring()
        sync_all()
        if world > 1:
            dist.barrier()
        t1 = time.perf_counter()
        for _ in range(n_ov):
            step_ring()
        sync_all()
        dt_single = time.perf_counter() - t1
        if world > 1:
            dist.barrier()
        gc.enable()
        if not two_events and not a.no_kernel_timing:
            k1_how = "per-launch HIP event pairs, 50 launches after the timed region"
            ctx.timing_reset()
            ctx.timing_enable(1 << capi.K_CS_DISTANCE)
            for _ in range(50):
                dev.search_shard_async(base, first, count, key.data_ptr())
            sync_all()
            k1_ms, k1_n = ctx.timing_get(capi.K_CS_DISTANCE)
            ctx.timing_enable(0)
        ar_us = None
        if comm is not None:
            ar_us = comm.allreduce_probe(200)
        elif world > 1:                                            # the torch.distributed collective, events on torch's stream
            for _ in range(3):
                dist.all_reduce(key, op=dist.ReduceOp.MIN)
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            e0.record()
            for _ in range(100):
                dist.all_reduce(key, op=dist.ReduceOp.MIN)
            e1.record()
            torch.cuda.synchronize()
            ar_us = e0.elapsed_time(e1) * 1e3 / 100
        # the replica check: every rank built its HoleMap by the same updates -- the ranks must hold the same bits
        # (every rank joins the check's collectives whatever happened to it locally: the library's call contributes words that
        # cannot match and reports afterwards, the torch.distributed form is handed None -- a rank that skipped them would leave
        # the others waiting)
        replicas = None
        try:
            if comm is not None:
                replicas = comm.replicas_equal(dev)
            else:
                words = None
                try:
                    words = dev.maps_checksum()
                except Exception as e:                             # noqa: BLE001
                    print("bench.py: maps_checksum failed on rank %d: %s" % (rank, e), file=sys.stderr)
                replicas = D.replicas_equal(words)
        except Exception as e:                                     # noqa: BLE001 -- a health check must never cost the line
            replicas = "failed: %s" % e
        # the fused per-scan form on every rank (slamhip_cs_search_allreduce_and_update: search, exchange, winner decoded on the device,
        # the replicas' map updates behind it -- no host hop): three scans, then the replica check again.  The first scan sees the
        # map of the timed region, so its key must be the timed region's; the maps move on from there (saved first for the oracle).
        pix_saved = dev.holemap_download() if rank == 0 else None
        fused = None
        if comm is not None:
            try:
                f_first = None
                tf0 = time.perf_counter()
                for k in range(3):
                    f_pose, f_dist, f_idx = comm.search_allreduce_and_update(dev, base, first, count, 0.6, 50, 10)
                    if k == 0:
                        f_first = ((int(f_dist) & 0xFFFFFFFF) << 32) | (int(f_idx) & 0xFFFFFFFF)
                ctx.synchronize()
                tf = (time.perf_counter() - tf0) / 3
                fused = {"first_scan_key_equals_search_key": bool(f_first == int(final_key)), "us_per_scan": tf * 1e6,
                         "replicas_equal_after": comm.replicas_equal(dev)}
            except Exception as e:                                 # noqa: BLE001
                fused = {"error": repr(e)}
        if world > 1:
            t = torch.tensor([elapsed, dt_single, ar_us if ar_us is not None else 0.0, per_scan_us], dtype=torch.float64, device="cuda")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed, dt_single, ar_max, per_scan_us = float(t[0].item()), float(t[1].item()), float(t[2].item()), float(t[3].item())
            ar_us = ar_max if ar_us is not None else None
        multi = {"timed_form": "enqueue-only search + key all-reduce enqueued behind it (the form N = 1 times, plus the exchange)",
                 "us_per_step": elapsed / a.steps * 1e6,
                 "single_rank_same_form_us_per_step": dt_single / n_ov * 1e6,
                 "efficiency_same_form": (dt_single / n_ov) / (elapsed / a.steps),
                 "efficiency_same_form_is": "this rank's search alone, same enqueue-only form, same process (max over ranks) / the timed region's step: what the exchange costs a step -- NOT the driver's 1 -> N scaling efficiency",
                 "per_scan_blocking_us_per_step": per_scan_us,
                 "allreduce_us": ar_us, "collective_ranks": collective_ranks, "replicas_equal": replicas,
                 "fused_scan_allreduce_and_update": fused}

    if rank == 0:
        evals = float(K_total) * a.steps
        value = evals / elapsed
        bytes_per_eval = 2 * a.rays + 16                     # BASELINE.md sec.3 / SURVEY.md sec.8d
        roof = None
        if k1_n > 0:
            avg_s = (k1_ms / k1_n) * 1e-3
            achieved = a.cands * bytes_per_eval / avg_s / 1e9
            traffic, traffic_src = pmc_traffic(a)
            peak_m, peak_src = measured_peak()
            valu = valu_ceiling(a, a.cands * bytes_per_eval)
            roof = {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic, "traffic_source": traffic_src,
                    "peak_measured": peak_m, "peak_measured_source": peak_src,
                    "kernel": "k1_search_tiled", "avg_launch_us": round(avg_s * 1e6, 3), "launches": int(k1_n),
                    "bytes_per_launch": a.cands * bytes_per_eval,
                    "timing": k1_how, "valu": valu}
            if first_search:
                roof.update(first_search)
        out = {
            "metric": "candidate-pose distance evals/sec on 2048^2 map, 1080-ray scan, 1/2/4/8 GPU",
            "value": value, "unit": "evals/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": elapsed / a.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32 transform + u16 gather / integer sum", "data": "synthetic",
            "config": {"workload": "CoreSLAM Monte-Carlo distance search, %dx%d HoleMap, %d rays, %d candidates/GPU/step"
                                   % (a.size, a.size, a.rays, a.cands),
                       "map": a.size, "rays": a.rays, "candidates_per_gpu": a.cands, "candidates_total": K_total,
                       "collective": collective, "collective_ranks": collective_ranks,
                       "timed_region": ("%d steps, each one search launch (+ its plan launch on a stream of its own); one device synchronise inside" % a.steps) if world == 1 else
                                       ("%d steps, each one search launch + the key's all-reduce enqueued behind it -- the same enqueue-only form as N = 1; one device synchronise inside" % a.steps),
                       "per_scan_blocking_us_per_step": per_scan_us,
                       "per_scan_blocking_is": "the same search as a BLOCKING call per step (K1%s, the key on the host before the next step): %d steps after the timed region; its key equals the timed region's: %s"
                                               % (" + the collective" if world > 1 or comm is not None else "", n_ps, blocking_key_ok),
                       "clock_warmup_steps": a.clock_warmup,
                       "cold_clocks": None if cold is None else {
                           "ms_per_step": cold * 1e3, "value": K_total / cold,
                           "note": "the same K steps behind the same W warm-up steps, measured before the clock warm-up: a burst on an idle chip"},
                       "search_plan": dict(zip(("searches_with_a_plan_launch", "without", "host_waits_for_a_plan_slot", "plans_skipped_inputs_in_flight"), dev.plan_stats)),
                       "best_index": final_key & 0xFFFFFFFF, "best_distance": final_key >> 32},
            "roofline": roof,
        }
        if multi is not None:
            out["multi_gpu"] = multi
        checks = []
        if world == 1 and not a.no_extras:
            # the other configurations of BASELINE.json and larger candidate counts, measured in this very process AFTER the
            # timed region (they are not part of `value`): a few seconds in all.  `checks` collects, per configuration, the
            # inputs and the GPU's answer; the CPU leg below compares them with the oracle (the only place it is used).
            try:
                out["other_workloads"] = other_workloads(a, ctx, dev, segs, xy, base, bytes_per_eval, checks)
            except Exception as e:                                 # noqa: BLE001 -- extras must never cost the headline line
                out["other_workloads"] = {"error": repr(e)}
            gc.enable()
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(a, dev, xy, base, offs, final_key, checks)
            if "other_workloads" in out and "error" not in out["other_workloads"]:
                out["other_workloads"]["winners_match_oracle"] = out["cpu_baseline"].pop("other_workloads_match", None)
        if world > 1:
            # N > 1: the reduced key of the last step against ONE oracle search over the whole flat list of K_total candidates
            # (rank 0's host, ~0.3 s at 131 072 candidates), so that a scaling line never carries a winner nobody verified; and
            # the CPU baseline on a short budget (the other ranks wait in the process group's teardown)
            try:
                sys.path.insert(0, os.path.join(ROOT, "oracle"))
                import oracle_c as oc
                oc.set_trig_mode(oc.TRIG_DET)
                obi, _, obd, _ = oc.search(pix_saved if pix_saved is not None else dev.holemap_download(), dev.hole_size, dev.hole_scale, xy, base, offs)
                out["config"]["winner_matches_oracle"] = bool(((int(obd) << 32) | int(obi)) == int(final_key))
            except Exception as e:                                 # noqa: BLE001
                out["config"]["winner_matches_oracle"] = "check failed: %r" % (e,)
            if not a.no_cpu_baseline:
                try:
                    a.cpu_seconds = min(a.cpu_seconds, 6.0)
                    a.cpu_short = True
                    n1 = a.cands - 1                               # (the per-GPU workload: rank 0's block of the list)
                    out["cpu_baseline"] = cpu_baseline(a, dev, xy, base, offs[:n1], None, (), pix=pix_saved)
                    out["cpu_baseline"]["sample"] += " (N > 1: short budget, one GPU's share of the candidates)"
                except Exception as e:                             # noqa: BLE001
                    out["cpu_baseline"] = {"error": repr(e)}
        print(json.dumps(out))
        sys.stdout.flush()
    if comm is not None:
        comm.close()
    dev.close()
    ctx.close()
    if world > 1:
        dist.destroy_process_group()



def other_workloads(a, ctx, dev, segs, xy, base, bytes_per_eval, checks):
    """Secondary figures, same process, after the headline's timed region: the headline search at larger candidate counts
    (the launch is latency-bound at 16 384 candidates: these show the kernel's throughput), and BASELINE.json's other
    configurations -- C2 (1024^2 map, 16 384 candidates), C3 (search + HoleMap / ObstacleMap update fused, one blocking
    call per scan), C4 (Hector match on a 3-level 2048^2 pyramid), C5's share of one GPU (4096^2 map, 32 768 candidates)."""
    import slam.net_amd.capi as capi
    import slam.net_amd.coreslam as cs
    import slam.net_amd.hector as hs
    import slam.net_amd.sim as sim
    import torch

    def time_search(d, pose, count, steps):
        # (at sustained clocks, like the headline: ~60 ms of the same launches first)
        tw = time.perf_counter()
        while True:
            for _ in range(20):
                d.search_shard_enqueue(pose, 0, count)
            ctx.synchronize()
            if time.perf_counter() - tw > 0.06 or a.clock_warmup <= 0:
                break
        t0 = time.perf_counter()
        for _ in range(steps):
            d.search_shard_enqueue(pose, 0, count)
        ctx.synchronize()
        dt = (time.perf_counter() - t0) / steps
        return {"us_per_step": dt * 1e6, "evals_per_s": count / dt,
                "roofline_frac": count * bytes_per_eval / dt / 1e9 / HBM_PEAK_GBS}

    out = {}
    sweep = {}
    gc.collect()
    gc.disable()                   # (re-enabled before returning; see main)
    for K, steps in ((65536, 100), (262144, 50), (1048576, 25)):
        dev.set_offsets(sim.gaussian_offsets(K - 1, 0.1, math.radians(10.0), seed=42))
        sweep[str(K)] = time_search(dev, base, K, steps)
    out["search_candidates_per_step_sweep_%d_map" % a.size] = sweep
    # BESIDE the Monte-Carlo figures, never instead of them: the same search over the device generator's lists -- headings stratified
    # (slamhip_cs_generate_offsets) and on the opt-in heading lattice (slamhip_cs_generate_offsets_lattice: the candidates of a lane
    # share their heading, the kernel forms the ray products once per lane).  Same sigmas, same kernel otherwise.
    lat = {}
    for K, steps in ((65536, 100), (262144, 50)):
        dev.generate_offsets(K - 1, 0.1, math.radians(10.0), seed=42, stream=1)
        plain = time_search(dev, base, K, steps)
        dev.generate_offsets(K - 1, 0.1, math.radians(10.0), seed=42, stream=1, lattice=True)
        lattice = time_search(dev, base, K, steps)
        lat[str(K)] = {"generated_stratified_headings": plain, "generated_heading_lattice": lattice}
    out["search_generated_lists_%d_map_not_the_headline" % a.size] = lat

    def mapped(size, rays=1080, updates=30):
        d = cs.CoreSlamDevice(ctx, 40.0, size, max(size // 4, 1))
        rng = sim.PCG32(1234)
        traj = sim.trajectory(updates + 1)
        for p in traj[:-1]:
            _, s_xy = sim.make_scan(segs, p, rays, rng)
            d.set_scan(s_xy)
            d.update_holemap(p, 0.6, 50)
        _, s_xy = sim.make_scan(segs, traj[-1], rays, rng)
        d.set_scan(s_xy)
        d._bench_xy = s_xy                                         # (kept for the parity records)
        return d, (traj[-1] + np.array([0.03, -0.02, math.radians(1.0)], np.float32)).astype(np.float32)

    def search_check(name, d, pose, offs):
        """the blocking search's answer + everything the CPU leg needs to repeat it with the oracle"""
        _, dist_, idx_ = d.search(pose)
        checks.append({"name": name, "pix": d.holemap_download(), "size": d.hole_size, "scale": d.hole_scale, "xy": d._bench_xy,
                       "base": pose, "offs": offs, "gpu": (int(idx_), int(dist_))})

    d2, b2 = mapped(1024)
    o2 = sim.gaussian_offsets(16383, 0.1, math.radians(10.0), seed=42)
    d2.set_offsets(o2)
    out["c2_search_1024_map_16384_candidates"] = time_search(d2, b2, 16384, 200)
    search_check("c2", d2, b2, o2)
    d2.close()
    # C5: 262 144 candidates over 8 GPUs on a 4096^2 map = 32 768 per GPU: rank 0's block of the flat list is timed, and the
    # min over the eight blocks' keys (this GPU playing every rank in turn) is the winner the CPU leg checks
    d5, b5 = mapped(4096)
    o5 = sim.gaussian_offsets(262143, 0.1, math.radians(10.0), seed=42)
    d5.set_offsets(o5)
    out["c5_one_gpu_share_4096_map_32768_candidates"] = time_search(d5, b5, 32768, 100)
    keys5 = [d5.search_shard(b5, 32768 * r, 32768) for r in range(8)]
    k5 = min(keys5)
    checks.append({"name": "c5_8_shards_of_262144", "pix": d5.holemap_download(), "size": d5.hole_size, "scale": d5.hole_scale, "xy": d5._bench_xy,
                   "base": b5, "offs": o5, "gpu": (int(k5 & 0xFFFFFFFF), int(k5 >> 32))})
    d5.close()
    # C3: one call per scan = search (16 384 candidates) + HoleMap update + ObstacleMap update.  The call returns when the
    # winner's pose is back (K1's final arriver delivers it); the map updates are enqueued behind the search and run on, and
    # the next call's search is ordered behind them.  us_per_scan: 100 calls back to back INCLUDING the last call's updates
    # (synchronised inside the timed region); us_to_pose: one call on an idle device, until it returns.
    d3, b3 = mapped(2048)
    o3 = sim.gaussian_offsets(16383, 0.1, math.radians(10.0), seed=42)
    d3.set_offsets(o3)
    pix_before = d3.holemap_download()
    pose3, dist3, idx3 = d3.search_and_update(b3)                  # the first fused scan, kept for the CPU leg's check
    checks.append({"name": "c3_fused", "pix": pix_before, "size": d3.hole_size, "scale": d3.hole_scale, "xy": d3._bench_xy, "base": b3,
                   "offs": o3, "gpu": (int(idx3), int(dist3)), "pose": np.asarray(pose3, np.float32), "pix_after": d3.holemap_download()})
    for _ in range(5):
        d3.search_and_update(b3)
    ctx.synchronize()
    t0 = time.perf_counter()
    for _ in range(100):
        d3.search_and_update(b3)
    ctx.synchronize()
    dt = (time.perf_counter() - t0) / 100
    lat = []
    for _ in range(50):
        ctx.synchronize()
        t1 = time.perf_counter()
        d3.search_and_update(b3)
        lat.append(time.perf_counter() - t1)
    ctx.synchronize()
    # the same with the host mirrors a source-compatible C# shim keeps (HoleMap.Pixels is read directly, HoleMap.cs:27,
    # Simulation/MainWindow.xaml.cs:229): after every scan the HoleMap rectangle the scan touched (slamhip_cs_holemap_mirror) and
    # the whole ObstacleMap come back -- and, for comparison, the whole HoleMap (8 MiB at 2048^2)
    mirror = d3.holemap_download()
    d3.holemap_mirror(mirror)
    t0 = time.perf_counter()
    for _ in range(50):
        d3.search_and_update(b3)
        rect = d3.holemap_mirror(mirror)
        d3.obstaclemap_download()
    ctx.synchronize()
    dt_m = (time.perf_counter() - t0) / 50
    t0 = time.perf_counter()
    for _ in range(20):
        d3.search_and_update(b3)
        d3.holemap_download()
        d3.obstaclemap_download()
    ctx.synchronize()
    dt_f = (time.perf_counter() - t0) / 20
    mirror_ok = bool((mirror == d3.holemap_download()).all())
    # ... and the asynchronous, span-exact mirror (slamhip_cs_holemap_mirror_async): after every scan ONE request -- the spans the
    # scan's rays crossed are snapshotted on the device and pushed into the host array from a copy stream while the next scan
    # runs; the array is waited for (slamhip_cs_holemap_mirror_wait, the C# `Pixels` getter) once per scan, just before the next
    # request, as a caller that looks at the map after every scan would.  us_to_pose: the fused call on an idle device with a push
    # of the previous scan still in flight -- what the mirror costs the scan itself.
    amirror = np.zeros(2048 * 2048, np.uint16)
    d3.holemap_mirror_async(amirror)
    d3.holemap_mirror_wait()
    for _ in range(5):
        d3.search_and_update(b3)
        d3.holemap_mirror_async(amirror)
    d3.holemap_mirror_wait()
    ctx.synchronize()
    t0 = time.perf_counter()
    for _ in range(100):
        d3.search_and_update(b3)
        d3.holemap_mirror_async(amirror)           # (waits for the previous push first: one in flight)
    _, pushed_px = d3.holemap_mirror_wait()
    ctx.synchronize()
    dt_a = (time.perf_counter() - t0) / 100
    amirror_ok = bool((amirror == d3.holemap_download()).all())
    lat_a = []
    for _ in range(30):
        ctx.synchronize()
        d3.holemap_mirror_async(amirror)
        t1 = time.perf_counter()
        d3.search_and_update(b3)
        lat_a.append(time.perf_counter() - t1)
        d3.holemap_mirror_wait()
    d3.holemap_mirror_release()
    out["c3_fused_search_and_map_updates_2048"] = {"us_per_scan": dt * 1e6, "scans_per_s": 1.0 / dt,
                                                    "us_per_scan_with_async_mirror": dt_a * 1e6, "async_mirror_pixels_per_scan": int(pushed_px),
                                                    "async_mirror_equals_full_download": amirror_ok,
                                                    "us_to_pose_with_async_push_in_flight_median": float(np.median(lat_a)) * 1e6,
                                                    "search_evals_per_s": 16384 / dt,
                                                    "us_to_pose_idle_device_median": float(np.median(lat)) * 1e6,
                                                    "us_per_scan_with_mirror": dt_m * 1e6, "us_per_scan_with_full_downloads": dt_f * 1e6,
                                                    "mirror_rect_x0_y0_x1_y1": list(rect), "mirror_equals_full_download": mirror_ok}
    d3.close()
    # the same through the CoreSLAMProcessor.Update mirror (CoreSLAMProcessor.cs:717-752): host scan (polar ranges) -> cartesian
    # cloud, sort + upload, 16 384 candidates generated on the device, search, both map updates, pose back -- 200 scans along the
    # trajectory, the last scan's map updates inside the timed region
    traj = sim.trajectory(80)
    rngp = sim.PCG32(5)
    pscans = [sim.make_scan(segs, p, 1080, rngp)[0] for p in traj]
    proc = cs.CoreSLAMProcessor(40.0, 2048, 512, traj[0], 0.1, math.radians(10.0), 16383 // 64, 64, ctx=ctx)
    zero = np.zeros(3, np.float32)
    for i in range(10):
        proc.Update([cs.ScanSegment(pscans[i], zero)])
    ctx.synchronize()
    t0 = time.perf_counter()
    per_scan = []
    for i in range(200):
        ts = time.perf_counter()
        proc.Update([cs.ScanSegment(pscans[10 + i % 60], zero)])
        per_scan.append(time.perf_counter() - ts)
    ctx.synchronize()
    dt = (time.perf_counter() - t0) / 200
    if os.environ.get("SLAMHIP_BENCH_STEP_TIMES"):
        sp = sorted(per_scan)
        print("[bench] Update through the mirror: median %.1f p90 %.1f max %.1f us; over 3x median: %s" % (
            sp[100] * 1e6, sp[180] * 1e6, sp[-1] * 1e6, [(i, round(t * 1e6)) for i, t in enumerate(per_scan) if t > 3 * sp[100]][:20]), file=sys.stderr)
    out["coreslam_processor_update_2048_map_1080_rays_16384_candidates"] = {"us_per_scan": dt * 1e6, "scans_per_s": 1.0 / dt,
                                                                            "caller": "Python mirror of the C# class over ctypes (the interpreter's part of a scan is ~8 us)",
                                                                            "searches_launched_ahead_abandoned_layout_remade_refused": list(proc.device.prelaunch_stats),
                                                                            "prepared_lists_served_prepared": list(proc.device.prepared_lists())}
    # the same loop with the host mirror of the HoleMap kept up to date: one asynchronous request per scan behind the Update
    # (what the C# shim's default MirrorMaps does) -- a MOVING robot, so what changes per scan is what a real run changes
    pm = np.zeros(2048 * 2048, np.uint16)
    proc.device.holemap_mirror_async(pm)
    proc.device.holemap_mirror_wait()
    px_acc = 0
    ctx.synchronize()
    t0 = time.perf_counter()
    for i in range(200):
        proc.Update([cs.ScanSegment(pscans[10 + i % 60], zero)])
        proc.device.holemap_mirror_async(pm)       # (waits for the previous push first)
        if i % 20 == 19:
            px_acc += proc.device.holemap_mirror_wait()[1]
    proc.device.holemap_mirror_wait()
    ctx.synchronize()
    dt_pm = (time.perf_counter() - t0) / 200
    pm_ok = bool((pm == proc.device.holemap_download()).all())
    # ... and the mirror refreshed WHEN IT IS READ (the shim's default MirrorMode.OnRead: the `Pixels` getter requests what changed
    # since its last read and waits for it; the scans themselves only keep the row spans): a reader every 33 scans -- a display
    # at 30 frames per second beside a 1 kHz scan loop
    ctx.synchronize()
    t0 = time.perf_counter()
    reads, t_read = 0, 0.0
    for i in range(198):
        proc.Update([cs.ScanSegment(pscans[10 + i % 60], zero)])
        if i % 33 == 32:
            tr = time.perf_counter()
            proc.device.holemap_mirror_async(pm)
            proc.device.holemap_mirror_wait()
            t_read += time.perf_counter() - tr; reads += 1
    ctx.synchronize()
    dt_or = (time.perf_counter() - t0) / 198
    or_ok = bool((pm == proc.device.holemap_download()).all())
    proc.device.holemap_mirror_release()
    out["coreslam_processor_update_2048_map_1080_rays_16384_candidates"].update({
        "us_per_scan_with_async_mirror": dt_pm * 1e6, "async_mirror_pixels_per_scan_sampled": px_acc // 10, "async_mirror_equals_full_download": pm_ok,
        "us_per_scan_with_mirror_read_every_33_scans": dt_or * 1e6, "us_per_mirror_read": t_read / max(reads, 1) * 1e6,
        "mirror_on_read_equals_full_download": or_ok})
    proc.Dispose()
    # ... and from a NATIVE caller of the C-ABI (tests/abi_harness.c --bench-proc: gcc, dlopen, slamhip_csproc_update in a C loop):
    # what a P/Invoke caller pays per scan.  A process of its own, while this one is idle.
    try:
        import shutil
        import subprocess
        import tempfile
        cc = shutil.which("gcc") or shutil.which("cc")
        if cc:
            with tempfile.TemporaryDirectory() as td:
                exe = os.path.join(td, "abi_harness")
                subprocess.check_call([cc, "-O1", "-o", exe, os.path.join(ROOT, "tests", "abi_harness.c"), "-ldl", "-lm"])
                r = subprocess.run([exe, capi.SO_PATH, "--bench-proc", "2048", "1080", "16385", "300"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=300)
                txt = r.stdout.decode(errors="replace")
                if r.returncode == 0 and "proc_us_per_scan" in txt:
                    us = float(txt.split("proc_us_per_scan")[1].split()[0])
                    us_cold = float(txt.split("from idle clocks:")[1].split(";")[0]) if "from idle clocks:" in txt else None
                    out["coreslam_processor_update_native_caller_2048_map_1080_rays_16385_candidates"] = {
                        "us_per_scan": us, "scans_per_s": 1e6 / us, "us_per_scan_first_300_scans_from_idle_clocks": us_cold,
                        "caller": "tests/abi_harness.c --bench-proc (C, dlopen): the simulator's field (the headline scan's scene), a moving robot; 300 scans timed after 1800 untimed ones (sustained clocks)"}
                else:
                    out["coreslam_processor_update_native_caller_2048_map_1080_rays_16385_candidates"] = {"error": txt[-400:]}
                # HectorSLAMProcessor.Update the same way (--bench-hsproc: 2048^2 x 3 levels, 1080 rays, every scan updating the grids)
                r = subprocess.run([exe, capi.SO_PATH, "--bench-hsproc", "2048", "3", "1080", "300"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=300)
                txt = r.stdout.decode(errors="replace")
                if r.returncode == 0 and "hsproc_us_per_scan" in txt:
                    us = float(txt.split("hsproc_us_per_scan")[1].split()[0])
                    out["hector_processor_update_native_caller_2048_pyramid_3_levels_1080_rays"] = {
                        "us_per_scan": us, "scans_per_s": 1e6 / us,
                        "caller": "tests/abi_harness.c --bench-hsproc (C, dlopen): the same field, every scan matched and the grids updated; 300 scans timed after 1800 untimed ones"}
                else:
                    out["hector_processor_update_native_caller_2048_pyramid_3_levels_1080_rays"] = {"error": txt[-400:]}
    except Exception as e:                                         # noqa: BLE001
        out["coreslam_processor_update_native_caller_2048_map_1080_rays_16385_candidates"] = {"error": repr(e)}
    # C4: Hector Gauss-Newton match, 3-level 2048^2 pyramid, 1080 rays
    rep = hs.MapRepMultiMap(40.0 / 2048, (2048, 2048), 3, ctx=ctx)
    rng = sim.PCG32(3)
    scans = []
    for it in range(12):
        p = np.array([20 + 0.05 * it, 20 + 0.02 * it, 0.01 * it], np.float32)
        scans.append((sim.make_scan(segs, p, 1080, rng)[1], p))
    for s_xy, p in scans:
        rep.UpdateByScan(hs.ScanCloud(s_xy), p)
    m = hs.ScanMatcher(4)
    s_xy, p = scans[-1]
    scan = hs.ScanCloud(s_xy)
    hint = p + np.array([0.1, -0.08, 0.03], np.float32)
    for _ in range(5):
        m.MatchData(rep, scan, hint)
    t0 = time.perf_counter()
    for _ in range(100):
        m.MatchData(rep, scan, hint)
    dt = (time.perf_counter() - t0) / 100
    B = 4096
    hints = np.tile(hint, (B, 1)) + np.random.default_rng(0).normal(0, 0.05, (B, 3)).astype(np.float32) * np.array([1, 1, 0.2], np.float32)
    m.MatchDataBatch(rep, scan, hints)
    t0 = time.perf_counter()
    for _ in range(5):
        m.MatchDataBatch(rep, scan, hints)
    dtb = (time.perf_counter() - t0) / 5
    out["c4_hector_match_3_level_2048_pyramid"] = {"us_per_match_blocking": dt * 1e6, "batched_hints": B,
                                                    "batched_matches_per_s_incl_transfers": B / dtb}
    # Per-kernel roofline fractions of the secondary kernels (SURVEY.md sec.8d's algorithmic bytes / the kernel's own launch time,
    # HIP events on the library's stream: slamhip_ctx_timing_*), so that the line shows them instead of leaving them to be derived:
    #   K2 4 B per blended pixel | K4 24 B per point-iteration (3 levels x 3 iterations x rays) | K5 16 B per touched cell + 4 B for
    #   its cached probability (touched = cells whose update index moved, counted on the host from two downloads).
    try:
        sec = {}
        PEAK = 8.0e12
        dk = cs.CoreSlamDevice(ctx, 40.0, 2048, 512)
        rngk = sim.PCG32(1234)
        trajk = sim.trajectory(40)
        scank = [sim.make_scan(segs, p, 1080, rngk)[1] for p in trajk]
        for i in range(8):
            dk.set_scan(scank[i]); dk.update_holemap(trajk[i])
        ctx.timing_reset(); ctx.timing_enable(1 << capi.K_CS_HOLEMAP)
        px = 0
        for i in range(8, 40):
            dk.set_scan(scank[i]); dk.update_holemap(trajk[i]); px += dk.last_holemap_pixels
        ms2, n2 = ctx.timing_get(capi.K_CS_HOLEMAP)
        ctx.timing_enable(0)
        us2 = ms2 / max(n2, 1) * 1e3
        sec["k2_holemap_update_2048_1080_rays"] = {"us_per_launch_hip_events": us2, "blended_pixels_per_update": px // max(n2, 1), "algorithmic_bytes_per_update": 4 * (px // max(n2, 1)),
                                                   "achieved_GBps": 4.0 * px / max(n2, 1) / (us2 * 1e-6) / 1e9, "roofline_frac": 4.0 * px / max(n2, 1) / (us2 * 1e-6) / PEAK,
                                                   "note": "a latency / instruction-bound kernel, not a bandwidth one (DESIGN.md sec.4 K2); stand-alone updates with host work between them: idle clocks"}
        dk.close()
        # K5: every scan updating the three grids
        ctx.timing_reset(); ctx.timing_enable(1 << capi.K_HS_UPDATE)
        for s5, p5 in scans:
            rep.UpdateByScan(hs.ScanCloud(s5), p5)
        ms5, n5 = ctx.timing_get(capi.K_HS_UPDATE)
        ctx.timing_enable(0)
        before = [mm.GetCells()["update_index"].copy() for mm in rep.Maps]
        rep.UpdateByScan(hs.ScanCloud(scans[0][0]), scans[0][1])
        touched = int(sum(int((mm.GetCells()["update_index"] != b).sum()) for mm, b in zip(rep.Maps, before)))
        us5 = ms5 / max(n5, 1) * 1e3
        sec["k5_hector_grid_update_3_level_2048"] = {"us_per_launch_hip_events": us5, "touched_cells_per_update": touched, "algorithmic_bytes_per_update": 20 * touched,
                                                     "achieved_GBps": 20.0 * touched / (us5 * 1e-6) / 1e9, "roofline_frac": 20.0 * touched / (us5 * 1e-6) / PEAK}
        # K4: the batched matcher (throughput form) and the single match (a latency chain), kernel time only
        ctx.timing_reset(); ctx.timing_enable(1 << capi.K_HS_MATCH)
        for _ in range(3):
            m.MatchDataBatch(rep, scan, hints)
        msb, nb = ctx.timing_get(capi.K_HS_MATCH)
        ctx.timing_reset()
        for _ in range(30):
            m.MatchData(rep, scan, hint)
        ms1, n1 = ctx.timing_get(capi.K_HS_MATCH)
        ctx.timing_enable(0)
        bytes_match = 3 * 3 * 1080 * 24
        usb, us1 = msb / max(nb, 1) * 1e3, ms1 / max(n1, 1) * 1e3
        sec["k4_hector_match_batched_4096_hints"] = {"us_per_launch_hip_events": usb, "matches_per_s_kernel_only": B / (usb * 1e-6), "algorithmic_bytes_per_match": bytes_match,
                                                     "achieved_GBps": B * bytes_match / (usb * 1e-6) / 1e9, "roofline_frac": B * bytes_match / (usb * 1e-6) / PEAK}
        sec["k4_hector_match_single"] = {"us_per_launch_hip_events": us1, "algorithmic_bytes_per_match": bytes_match, "roofline_frac": bytes_match / (us1 * 1e-6) / PEAK,
                                         "note": "nine dependent Gauss-Newton iterations: a latency chain by construction"}
        out["secondary_kernels_roofline"] = sec
    except Exception as e:                                         # noqa: BLE001
        out["secondary_kernels_roofline"] = {"error": repr(e)}
    # the headline list is restored for the CPU baseline's parity spot-check
    dev.set_offsets(sim.gaussian_offsets(a.cands - 1, 0.1, math.radians(10.0), seed=42))
    gc.enable()
    return out


def pmc_traffic(a):
    """HBM-side bytes per K1 launch from the committed rocprofv3 --pmc passes (FETCH_SIZE and WRITE_SIZE collected in
    separate runs of this very command by tools/prof_bench.sh, gfx950 read-side correction applied;
    profiles/rNN_k1_traffic.json, newest round first).  PMC counters cannot be read from inside the timed run, so the
    figure is only reported when the committed profile was taken on the same workload; otherwise null."""
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_k1_traffic.json")), reverse=True):
        try:
            with open(path) as f:
                t = json.load(f)
            w = t["workload"]
            if (w["map"], w["rays"], w["candidates_per_gpu"]) == (a.size, a.rays, a.cands):
                return int(t["hbm_bytes_per_launch_gfx950_corrected"]), "replayed from %s (a separate rocprofv3 --pmc run of this command, not this run)" % os.path.relpath(path, ROOT)
        except Exception:
            pass
    return None, None


def valu_ceiling(a, bytes_per_launch):
    """What actually binds the dominant kernel at this size: its VALU work.  Replayed from the committed SQ-counter profile of this
    command (profiles/rNN_k1_sq.json, newest round first; PMC counters cannot be read inside the timed run): busy_us = the time the
    VALU pipes of a SIMD are busy per launch = SQ_ACTIVE_INST_VALU quad-cycles x 4 / 1024 SIMDs / the shader clock MEASURED in the same
    profile (GRBM_GUI_ACTIVE / launch duration, tools/prof_summary.py -- not an assumed clock), launch_us = the launch's duration in the
    same profile, frac_of_hbm_roofline_if_valu_bound = the HBM-roofline fraction the launch would reach if it took only busy_us -- the
    ceiling of this kernel at this candidate count, whatever its latencies.  None without a matching profile or when the profile's
    figures are not self-consistent (busy_us must not exceed launch_us)."""
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_k1_sq.json")), reverse=True):
        try:
            with open(path) as f:
                t = json.load(f)
            h = t["headline_16384_candidates"]
            if a.cands != 16384 or a.size != 2048 or a.rays != 1080:
                return None
            d = h["derived"]
            if "shader_clock_ghz_measured" not in d:
                continue                                           # (a profile of an earlier round: its busy estimate assumed a clock)
            busy = float(d["valu_busy_us_per_simd_if_evenly_spread"])
            launch = float(h["avg_launch_ns_rocprof_stats"]) * 1e-3
            if not busy <= launch:
                return {"error": "profile not self-consistent: valu busy %.2f us > launch %.2f us" % (busy, launch), "source": os.path.relpath(path, ROOT)}
            return {"busy_us": round(busy, 2), "launch_us": round(launch, 2), "busy_frac_of_launch": round(busy / launch, 3),
                    "shader_clock_ghz_measured": round(float(d["shader_clock_ghz_measured"]), 3),
                    "frac_of_hbm_roofline_if_valu_bound": round(bytes_per_launch / (busy * 1e-6) / 1e9 / HBM_PEAK_GBS, 3),
                    "wave_cycles_waiting_frac": round(float(d.get("fraction_of_wave_cycles_waiting_waitcnt_or_barrier", 0.0)), 3),
                    "source": "replayed from %s (separate rocprofv3 --pmc passes of this command, not this run)" % os.path.relpath(path, ROOT)}
        except Exception:
            pass
    return None


def measured_peak():
    """Read bandwidth of a stream over 4 GiB on this box (tools/hbm_probe.py, committed as profiles/rNN_hbm_probe.json, newest
    round first): the ceiling the hardware delivers, next to the 8 TB/s specification `peak` is quoted from.  GB/s or null."""
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_hbm_probe.json")), reverse=True):
        try:
            with open(path) as f:
                return float(json.load(f)["4_GiB"]["sum_GBps_read"]), "replayed from %s (tools/hbm_probe.py; not measured in this run)" % os.path.relpath(path, ROOT)
        except Exception:
            pass
    return None, None


def cpu_baseline(a, dev, xy, base, offs, gpu_key, checks=(), pix=None):
    """The reference's ParallelWorker-structured CPU search (oracle/cpu_baseline.c, kind = "port": the C#
    reference cannot run here) on the SAME map / scan / candidates, bounded to ~a.cpu_seconds of CPU work at
    T = min(nproc, 64) threads, plus short runs at T = 1 and T = 4 and BASELINE.json's config C1 (400^2 map,
    360 rays, 1000 iterations per thread; Simulation/MainWindow.xaml.cs:69) at T = 1, 4, nproc.
    `value` is total evaluations / total seconds of the long run (the mean; thread wake-ups make a few scans
    slow: the per-scan median is reported beside it).  This is also where the GPU's winners are compared with
    the oracle: the headline's and those of the other configurations (`checks`)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle_c as oc
    import slam.net_amd.sim as sim
    oc.set_trig_mode(oc.TRIG_DET)
    if pix is None:
        pix = dev.holemap_download()
    # WaitHandle.WaitAll caps the reference's ParallelWorker at 64 threads (BaseSLAM/ParallelWorker.cs:115)
    T = min(os.cpu_count() or 1, 64)
    n = offs.shape[0]
    iters = max(n // T, 1)
    secs, evals, bi, bd = oc.cpu_baseline_search(pix, dev.hole_size, dev.hole_scale, xy, base, offs, T, iters, 1)
    rate = evals / secs
    scans = max(int(a.cpu_seconds * rate / evals) - 1, 1)
    secs2, evals2, bi, bd, per = oc.cpu_baseline_search_timed(pix, dev.hole_size, dev.hole_scale, xy, base, offs, T, iters, scans)
    short = bool(getattr(a, "cpu_short", False))                   # (N > 1: the headline sample only, a few seconds)
    if secs2 < 10.0 and not short:           # (the first scans run cold: the calibration above undershoots) -- aim for ~15 s of measured work
        scans = max(int(scans * 15.0 / max(secs2, 1e-3)), scans)
        secs2, evals2, bi, bd, per = oc.cpu_baseline_search_timed(pix, dev.hole_size, dev.hole_scale, xy, base, offs, T, iters, scans)
    # parity spot-check on the full candidate list of the GPU step (single oracle pass, ~0.1 s)
    rbi, _, rbd, _ = oc.search(pix, dev.hole_size, dev.hole_scale, xy, base, offs)
    same = bool(((rbd << 32) | rbi) == gpu_key) if gpu_key is not None else None

    def short_run(pix_, size_, scale_, xy_, base_, offs_, T_, iters_, seconds):
        s_, e_, _, _ = oc.cpu_baseline_search(pix_, size_, scale_, xy_, base_, offs_, T_, iters_, 3)        # warm-up: 3 scans
        sc = max(int(seconds / max(s_ / 3, 1e-6)), 20)
        s_, e_, _, _, per_ = oc.cpu_baseline_search_timed(pix_, size_, scale_, xy_, base_, offs_, T_, iters_, sc)
        return {"evals_per_s": e_ / s_, "median_scan": (e_ / sc) / float(np.median(per_)), "scans": sc, "seconds": round(s_, 2),
                "iterations_per_thread": iters_}

    sweep = {}
    for Ts in (1, 4):
        if Ts < T and not short:
            sweep["T%d" % Ts] = short_run(pix, dev.hole_size, dev.hole_scale, xy, base, offs, Ts, max(n // Ts // 16, 1), 2.5)
    # config C1: the reference's own CPU-runnable case -- map built by 30 oracle mapping updates (the GPU plays no part)
    c1 = {}
    try:
        if short:
            raise StopIteration
        size1, R1, it1 = 400, 360, 1000
        scale1 = size1 / 40.0
        pix1 = np.full(size1 * size1, 32750, np.uint16)
        segs = sim.default_field()
        rng = sim.PCG32(1234)
        traj = sim.trajectory(31)
        for p in traj[:-1]:
            _, xy1 = sim.make_scan(segs, p, R1, rng)
            oc.update_holemap(pix1, size1, scale1, xy1, p, 0.6, 50)
        _, xy1 = sim.make_scan(segs, traj[-1], R1, rng)
        base1 = (traj[-1] + np.array([0.03, -0.02, math.radians(1.0)], np.float32)).astype(np.float32)
        for Ts in sorted(set([1, 4, T])):
            offs1 = sim.gaussian_offsets(Ts * it1, 0.1, math.radians(10.0), seed=42)
            c1["T%d" % Ts] = short_run(pix1, size1, scale1, xy1, base1, offs1, Ts, it1, 2.5)
    except StopIteration:
        c1 = {"skipped": "N > 1 run: short CPU budget"}
    except Exception as e:                                         # noqa: BLE001
        c1 = {"error": repr(e)}
    # the other configurations' winners (bench.other_workloads collected the inputs and the GPU's answers)
    match = {}
    for c in checks:
        try:
            obi, opose, obd, _ = oc.search(c["pix"], c["size"], c["scale"], c["xy"], c["base"], c["offs"])
            ok = (int(obi), int(obd)) == c["gpu"]
            if "pose" in c:                                        # fused search + update: pose and the HoleMap after the update
                opose[2] = oc.normalize_angle(opose[2])
                ok = ok and bool((np.asarray(opose, np.float32) == c["pose"]).all())
                ref = c["pix"].copy()
                oc.update_holemap(ref, c["size"], c["scale"], c["xy"], opose)
                ok = ok and bool((ref == c["pix_after"]).all())
            match[c["name"]] = bool(ok)
        except Exception as e:                                     # noqa: BLE001
            match[c["name"]] = repr(e)
    return {"value": evals2 / secs2, "unit": "evals/s", "cores": T, "threads": T, "host_cores": os.cpu_count(),
            "cores_is": "the threads the baseline ran (the reference's ParallelWorker is capped at 64 by WaitHandle.WaitAll); host_cores = the box's online cores",
            "kind": "port",
            "sample": "%d scans x %d threads x (%d jitters + base) on the same map/scan/candidates, %.1f s"
                      % (scans, T, iters, secs2),
            "value_is": "mean over the sample (total evaluations / total seconds); median_scan and p95_scan are per-scan rates",
            "median_scan": (evals2 / scans) / float(np.median(per)), "p95_scan": (evals2 / scans) / float(np.percentile(per, 95)),
            "argmin_matches_gpu": same,
            "threads_sweep_headline": sweep, "c1_400_map_360_rays_1000_iterations_per_thread": c1,
            "other_workloads_match": match}


if __name__ == "__main__":
    main()
