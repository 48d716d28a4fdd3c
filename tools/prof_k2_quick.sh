#!/bin/bash
# quick rocprofv3 look at the HoleMap update (on the GPU box): average launch by --kernel-trace --stats, then one pass of SQ counters.
# usage: bash tools/prof_k2_quick.sh [tag]   (output under gpurun_out/<tag>/k2q)
tag=${1:-k2q}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/$tag/k2q
rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
k2="python3 $root/tools/prof_k2.py"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -o stats -- $k2 > $out/stats.out 2> $out/stats.log
grep -h "k2_\|k3_" $(find $out/stats -name "*kernel_stats.csv") | cut -c1-160
SQ1="SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY"
timeout 300 rocprofv3 --kernel-trace --pmc $SQ1 --output-format csv -d $out/sq1 -o p -- $k2 > $out/sq1.out 2> $out/sq1.log
python3 - "$out" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob(out + "/sq1/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k2_pixels" not in r["Kernel_Name"]: continue
        a = acc[r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
for k, (v, n) in sorted(acc.items()):
    print("%-24s %14.0f per launch (%d dispatches)" % (k, v / max(n, 1), n))
PY
find $out -name "*.csv" -size +200k -delete
