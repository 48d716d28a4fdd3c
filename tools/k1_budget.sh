#!/bin/bash
# The search kernel's instruction budget (on the GPU box): SQ_INSTS_VALU / SALU / LDS and the launch time of k1_search_tiled at the headline
# workload for the developer builds SLAMHIP_K1_EXP = 0 .. 4 (parts of the kernel left out, wrong results: distance.hip K1_EXP); the
# differences are what each part costs.   bash tools/k1_budget.sh [tag]  ->  gpurun_out/<tag>/k1_budget.txt
tag=${1:-k1b}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
bench="python3 $root/bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-extras"
: > $out/k1_budget.txt
# (the EXP builds compute WRONG results: they are side variants selected with SLAMHIP_LIB -- tools/build_variant.py -- and never replace
# slam.net_amd/libslamhip.so, so a killed run leaves the tree's library as it was)
for e in 0 1 2 3 4; do
  (cd $root && python3 tools/build_variant.py WORK k1exp$e -DK1_EXP=$e > /dev/null 2>&1)
  export SLAMHIP_LIB=$root/slam.net_amd/build/variants/k1exp$e.so
  rm -rf /tmp/k1b_$e
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/k1b_$e/st -o s -- $bench > /dev/null 2>&1
  timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES --output-format csv -d /tmp/k1b_$e/pm -o p -- $bench > /dev/null 2>&1
  python3 - /tmp/k1b_$e $e >> $out/k1_budget.txt <<'PY'
import csv, glob, sys, collections
d, e = sys.argv[1], sys.argv[2]
line = "EXP %s:" % e
for f in glob.glob(d + "/st/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k1_search_tiled" in r["Name"]: line += " launch avg %.2f us (min %.2f, %s calls)" % (float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, r["Calls"])
acc = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob(d + "/pm/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k1_search_tiled" in r["Kernel_Name"]:
            a = acc[r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
w = acc["SQ_WAVES"][0] / max(acc["SQ_WAVES"][1], 1)
for k in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS"):
    v = acc[k][0] / max(acc[k][1], 1)
    line += " | %s %.0f (%.0f per wave)" % (k[9:], v, v / max(w, 1))
print(line)
PY
done
unset SLAMHIP_LIB
cat $out/k1_budget.txt
