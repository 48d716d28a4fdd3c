"""Turn the rocprofv3 passes of tools/prof_bench.sh into the committed profile files.
usage: python tools/prof_summary.py r02   (reads gpurun_out/r02/prof, writes profiles/r02_*)"""
import collections, csv, glob, json, os, sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", tag, "prof")
dst = os.path.join(root, "profiles")


def find(sub, pattern):
    hits = glob.glob(os.path.join(src, sub, "**", pattern), recursive=True)
    return hits[0] if hits else None


def pmc(sub):
    """mean counter value per dispatch, per kernel"""
    f = find(sub, "*counter_collection.csv")
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    disp = collections.defaultdict(set)
    if not f:
        return {}
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        disp[k].add(r["Dispatch_Id"])
    return {k: {c: v / max(len(disp[k]), 1) for c, v in cs.items()} for k, cs in agg.items()}


# 1. per-kernel statistics
stats = find("stats", "*kernel_stats.csv")
rows = list(csv.DictReader(open(stats))) if stats else []
with open(os.path.join(dst, tag + "_bench_kernel_stats.csv"), "w") as f:
    if rows:
        w = csv.DictWriter(f, fieldnames=list(rows[0].keys()), quoting=csv.QUOTE_NONNUMERIC)
        w.writeheader()
        w.writerows(rows)
k1 = next((r for r in rows if "k1_search_tiled" in r["Name"]), None)

# 2. the bench lines of the runs
bench = {}
for name in ("bench", "stats_bench", "fetch_bench"):
    p = os.path.join(src, name + ".json")
    if os.path.exists(p):
        for line in open(p):
            if line.startswith("{"):
                bench[name] = json.loads(line)
if "bench" in bench:
    json.dump(bench["bench"], open(os.path.join(dst, tag + "_bench.json"), "w"), indent=1)

# 3. HBM-side traffic of K1
fetch, write, tcc = pmc("fetch"), pmc("write"), pmc("tcc")
k1name = next((k for k in fetch if "k1_search_tiled" in k), None)
if k1name:
    fs = fetch[k1name].get("FETCH_SIZE", 0.0)          # KB per dispatch
    ws = write.get(k1name, {}).get("WRITE_SIZE", 0.0)
    cfg = bench.get("fetch_bench", bench.get("bench", {})).get("config", {})
    out = {
        "kernel": k1name,
        "workload": {"map": cfg.get("map"), "rays": cfg.get("rays"), "candidates_per_gpu": cfg.get("candidates_per_gpu")},
        "command": "tools/prof_bench.sh: rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE | TCC_HIT TCC_MISS (separate passes) -- python3 bench.py --steps 100 --warmup 20 --no-cpu-baseline",
        "FETCH_SIZE_KB_per_launch": fs, "WRITE_SIZE_KB_per_launch": ws,
        "hbm_bytes_per_launch_raw": (fs + ws) * 1024.0,
        "hbm_bytes_per_launch_gfx950_corrected": (2.0 * fs + ws) * 1024.0,
        "note": "FETCH_SIZE on gfx950 reports half the bytes of wide coalesced reads (MI355X_MICROARCH.md, HBM): the tile staging loads are 16 B per lane, so the corrected figure doubles the read side.  Infinity-Cache hits are counted: this is L2 <-> fabric traffic, not DRAM traffic (the 8 MiB map is cache resident).  The write side is the per-candidate accumulator atomics, performed at the memory side (uncalibrated).",
        "TCC_HIT_per_launch": tcc.get(k1name, {}).get("TCC_HIT"), "TCC_MISS_per_launch": tcc.get(k1name, {}).get("TCC_MISS"),
        "k1_avg_ns_rocprof_stats": float(k1["AverageNs"]) if k1 else None,
        "k1_avg_us_hip_events_same_run": bench.get("stats_bench", {}).get("roofline", {}).get("avg_launch_us"),
    }
    json.dump(out, open(os.path.join(dst, tag + "_k1_traffic.json"), "w"), indent=1)
json.dump({"fetch": fetch, "write": write, "tcc": tcc}, open(os.path.join(dst, tag + "_bench_pmc_per_kernel.json"), "w"), indent=1)
print("wrote profiles/%s_*; K1 rocprof avg %s ns" % (tag, k1["AverageNs"] if k1 else "?"))
