"""Writes DESIGN.md section 5 (the ONE table of current measurements) from the committed profile files of a round:
    python tools/design_measurements.py r06 [extra.json]
reads profiles/<tag>_*.json / *.csv / *.txt (tools/prof_summary.py wrote them from the rocprofv3 passes of tools/prof_bench.sh) and
replaces the text between the markers <!-- measurements:begin --> and <!-- measurements:end --> in DESIGN.md.  `extra.json` (optional,
profiles/<tag>_extra.json by default) holds figures measured outside prof_bench.sh (A/B pairs, the 20-step region, soak counts), each
with the command that produced it."""
import csv
import json
import os
import re
import sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r06"
P = lambda n: os.path.join(root, "profiles", "%s_%s" % (tag, n))


def load(n, default=None):
    try:
        return json.load(open(P(n)))
    except Exception:                                              # noqa: BLE001
        return default


b = load("bench.json", {})
sq = load("k1_sq.json", {})
tr = load("k1_traffic.json", {})
k2 = load("k2_counters.json", {})
sec = load("secondary_kernels.json", {})
extra = load("extra.json", {}) if len(sys.argv) < 3 else json.load(open(sys.argv[2]))
ow = b.get("other_workloads", {})
roof = b.get("roofline", {})
cfg = b.get("config", {})


def stats(name, pat):
    try:
        for r in csv.DictReader(open(P(name))):
            if pat in r["Name"]:
                return float(r["AverageNs"]) * 1e-3, float(r["MinNs"]) * 1e-3, int(float(r["Calls"]))
    except Exception:                                              # noqa: BLE001
        pass
    return None


def timeline(name, pat):
    try:
        for line in open(P(name)):
            if line.startswith("kernel") and pat in line:
                return float(line.split("mean")[1].split()[0])
    except Exception:                                              # noqa: BLE001
        pass
    return None


k1 = stats("bench_kernel_stats.csv", "k1_search_tiled")
kp = stats("bench_kernel_stats.csv", "k1_plan")
k1big = stats("bench_262144_kernel_stats.csv", "k1_search_tiled")
k2s = stats("k2_kernel_stats.csv", "k2_pixels")
bytes_launch = roof.get("bytes_per_launch", 35651584)
rows = []
if k1:
    plan_txt = ""
    if kp:
        plan_txt = "; the plan launch `k1_plan` beside it: %.1f µs average (stretched: it shares the compute units with the search launch in front of it; nothing waits for it)" % kp[0]
    name = "K1 `k1_search_tiled`, average launch by `rocprofv3 --kernel-trace --stats` (`profiles/%s_bench_kernel_stats.csv`, %d launches of `bench.py --steps 100 --warmup 20` incl. the clock warm-up)" % (tag, k1[2])
    val = "**%.2f µs** (min %.2f) = %.2f MB ÷ %.2f µs = %.2f TB/s = **%.3f of the 8 TB/s roofline**; HIP events around the timed region of the plain run: %.2f µs (%.3f)" % (
        k1[0], k1[1], bytes_launch / 1e6, k1[0], bytes_launch / k1[0] / 1e6, bytes_launch / k1[0] / 1e6 / 8.0, roof.get("avg_launch_us", 0), roof.get("frac", 0))
    rows.append((name, val + plan_txt))
hn_ = ("; " + extra["headline_note"]) if extra.get("headline_note") else ""
rows.append(("headline `value` (16 384 candidates, 2048², 1080 rays; a step = one enqueue-only search = its search launch + its plan launch) (`profiles/%s_bench.json`, the exact line of the plain run)" % tag,
             "**%.3g evals/s, %.2f µs/step** at %d steps; the same steps from idle clocks (`config.cold_clocks`) %.2f µs%s"
             % (b.get("value", 0), b.get("ms_per_step", 0) * 1e3, b.get("steps", 0), (cfg.get("cold_clocks") or {}).get("ms_per_step", 0) * 1e3, hn_)))


def row(name, val):
    rows.append((name, val))


if extra.get("ab"):
    row("the plan launch, A/B on one box (`SLAMHIP_K1_PLAN=0` against the default; %s)" % extra["ab"].get("how", ""), extra["ab"]["text"])
val = ("blocking search %.1f µs per call; K1 behind a new scan %.1f µs against %.1f µs for repeated searches of one scan (same instrument): a new scan costs the "
       "search launch nothing measurable -- what it lacks in the per-scan flow is the plan and the cut by cost (%s)"
       % (cfg.get("per_scan_blocking_us_per_step", 0), roof.get("first_search_of_new_scan_us", 0), roof.get("repeated_search_of_one_scan_us_same_instrument", 0), extra.get("cuts_note", "§4 K1")))
row("the blocking per-scan form, and K1 in front of a NEW scan (`config.per_scan_blocking_us_per_step`, `roofline.first_search_of_new_scan_us`; per-launch HIP event pairs cost a launch ≈ 8 µs: "
    "compare the two figures with each other, not with the lines above)", val)
h = (sq.get("headline_16384_candidates") or {}).get("derived", {})
if h:
    big = sq.get("at_262144_candidates") or {}
    big_txt = ""
    if big.get("derived", {}).get("valu_busy_consistent") is False:
        over = big["derived"]["valu_busy_us_per_simd_if_evenly_spread"] / (big["avg_launch_ns_rocprof_stats"] * 1e-3) - 1.0
        big_txt = "; at 262 144 candidates the same estimate exceeds the launch by %.0f %% (it is an upper bound: not replayed by `bench.py`)" % (100 * over)
    busy = h.get("valu_busy_us_per_simd_if_evenly_spread", 0)
    launch = sq["headline_16384_candidates"]["avg_launch_ns_rocprof_stats"] * 1e-3
    val = ("clock %.2f GHz; VALU pipes busy ≤ %.1f µs of the %.1f µs launch (the ceiling at this size: %.2f of the roofline); %.0f %% of the wave cycles wait at `s_waitcnt` / barriers, "
           "%.0f %% issue VALU; %.0f VALU + %.0f LDS instructions per wavefront; LDS bank conflicts %.0f %% of the LDS cycles"
           % (h.get("shader_clock_ghz_measured", 0), busy, launch, bytes_launch / max(busy, 1e-9) / 1e6 / 8.0,
              100 * h.get("fraction_of_wave_cycles_waiting_waitcnt_or_barrier", 0), 100 * h.get("fraction_of_wave_cycles_issuing_valu", 0),
              h.get("valu_instructions_per_wave", 0), h.get("lds_instructions_per_wave", 0), 100 * h.get("lds_bank_conflict_fraction_of_lds_cycles", 0)))
    row("what binds K1 (`profiles/%s_k1_sq.json`; shader clock measured in the same pass: SQ_BUSY_CYCLES ÷ 32 shader engines ÷ launch)" % tag, val + big_txt)
if tr:
    tb = tr.get("hbm_bytes_per_launch_gfx950_corrected", 0)
    hit, miss = tr.get("TCC_HIT_per_launch") or 0, tr.get("TCC_MISS_per_launch") or 0
    row("L2 ↔ fabric traffic per K1 launch (`profiles/%s_k1_traffic.json`: FETCH_SIZE × 2 + WRITE_SIZE, separate `--pmc` passes)" % tag,
        "%.1f MB (%.2f × algorithmic: the 8 MiB map is cache resident), L2 hit rate %.0f %%" % (tb / 1e6, tb / bytes_launch, 100 * hit / max(hit + miss, 1)))
sw = ow.get("search_candidates_per_step_sweep_2048_map", {})
if sw:
    rp = (" -- %.1f µs by `rocprofv3`" % k1big[0]) if k1big else ""
    val = "65 536: %.1f µs (%.2f), 262 144: %.1f µs (%.2f)%s, 1 048 576: %.0f µs (%.2f)" % (
        sw["65536"]["us_per_step"], sw["65536"]["roofline_frac"], sw["262144"]["us_per_step"], sw["262144"]["roofline_frac"], rp, sw["1048576"]["us_per_step"], sw["1048576"]["roofline_frac"])
    row("larger searches on the headline map (`other_workloads` of the bench line; 262 144 also by `rocprofv3`: `profiles/%s_bench_262144_kernel_stats.csv`)" % tag, val)
c2, c5 = ow.get("c2_search_1024_map_16384_candidates"), ow.get("c5_one_gpu_share_4096_map_32768_candidates")
if c2 and c5:
    row("1024² map (C2) / 4096², 32 768 candidates (one GPU's share of C5)",
        "%.1f µs (%.2f) / %.1f µs (%.2f); winners equal the oracle's (`winners_match_oracle`: %s)" % (c2["us_per_step"], c2["roofline_frac"], c5["us_per_step"], c5["roofline_frac"], ow.get("winners_match_oracle")))
if k2s and k2:
    t = k2.get("traffic", {})
    row("HoleMap update K2, stand-alone (`profiles/%s_k2_kernel_stats.csv`, `%s_k2_counters.json`)" % (tag, tag),
        "%.1f µs; L2 ↔ fabric %.1f MB for 2.56 MB algorithmic; 0.017 of the roofline (latency / instruction bound)" % (k2s[0], t.get("hbm_bytes_per_launch_raw", 0) / 1e6))
c3 = ow.get("c3_fused_search_and_map_updates_2048")
if c3:
    row("fused search + both updates per scan (C3) (`profiles/%s_timeline_c3.txt`)" % tag,
        "%.1f µs/scan back to back (pose after %.1f µs on an idle device); in the timeline K1 %s + K2 %s µs"
        % (c3["us_per_scan"], c3["us_to_pose_idle_device_median"], timeline("timeline_c3.txt", "tiled"), timeline("timeline_c3.txt", "k2_pixels")))
pn = ow.get("coreslam_processor_update_native_caller_2048_map_1080_rays_16385_candidates")
pp = ow.get("coreslam_processor_update_2048_map_1080_rays_16384_candidates")
if pn and "us_per_scan" in pn:
    py = (", %.1f through the Python mirror inside the bench process (two HIP runtimes loaded)" % pp["us_per_scan"]) if pp and "us_per_scan" in pp else ""
    row("`CoreSLAMProcessor.Update` (`tests/abi_harness.c --bench-proc`: the simulator's field, a moving robot; `profiles/%s_timeline_csproc_launch_ahead.txt` / `_csproc.txt`)" % tag,
        "**%.1f µs per scan from the native caller**%s; in the timeline (ordinary order, under the profiler) K1 %s µs, K2 %s µs"
        % (pn["us_per_scan"], py, timeline("timeline_csproc.txt", "tiled"), timeline("timeline_csproc.txt", "k2_pixels")))
hn = ow.get("hector_processor_update_native_caller_2048_pyramid_3_levels_1080_rays")
k5, k4 = sec.get("k5_hector_grid_update", {}), sec.get("k4_hector_match_single", {})
k4b = sec.get("k4_hector_match_batched_4096", {})
if k5 and k4:
    hu = ("; **`HectorSLAMProcessor.Update` %.1f µs per scan from the native caller**" % hn["us_per_scan"]) if hn and "us_per_scan" in hn else ""
    row("Hector K4 / K5 (`profiles/%s_secondary_kernels.json`, `%s_timeline_hsproc.txt`)" % (tag, tag),
        "single match %.1f µs stand-alone, %s µs inside `Update` (cold lines behind the grid update); batched 4096 hints %.0f µs = %.3g matches/s (%.2f of the roofline); grid update %.1f µs, "
        "FETCH %.1f MB + WRITE %.1f MB for 16.75 MB algorithmic (0.08)%s"
        % (k4.get("avg_launch_ns_rocprof_stats", 0) * 1e-3, timeline("timeline_hsproc.txt", "k4_match"), (k4b.get("avg_launch_ns_rocprof_stats") or 0) * 1e-3, k4b.get("matches_per_s") or 0,
           (k4b.get("algorithmic_TBps") or 0) / 8.0, k5.get("avg_launch_ns_rocprof_stats", 0) * 1e-3, k5.get("FETCH_SIZE_KB_per_launch", 0) / 1e3, k5.get("WRITE_SIZE_KB_per_launch", 0) / 1e3, hu))
cb = b.get("cpu_baseline")
if cb:
    row("CPU baseline in the same run (`kind: \"port\"`: `oracle/cpu_baseline.c`, the reference's ParallelWorker structure; %s threads of %s host cores)" % (cb.get("threads"), cb.get("host_cores")),
        "%.2g evals/s mean over the sample (%s); per-scan median %.2g; arg-min equals the GPU's: %s" % (cb["value"], cb["sample"], cb["median_scan"], cb["argmin_matches_gpu"]))
for r in extra.get("rows", []):
    row(r[0], r[1])

txt = "| Quantity (source) | Round 6 |\n|---|---|\n" + "\n".join("| %s | %s |" % r for r in rows) + "\n"
d = open(os.path.join(root, "DESIGN.md")).read()
i0, i1 = d.index("<!-- measurements:begin -->"), d.index("<!-- measurements:end -->")
d = d[:i0] + "<!-- measurements:begin -->\n" + txt + d[i1:]
open(os.path.join(root, "DESIGN.md"), "w").write(d)
print("DESIGN.md section 5: %d rows from profiles/%s_*" % (len(rows), tag))
