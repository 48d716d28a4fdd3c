#!/bin/bash
# A/B of HoleMap-update builds on ONE box: for every slam.net_amd/build/variants/*.so, three rocprofv3 --kernel-trace --stats runs of
# tools/prof_k2.py; prints the k2_pixels average / minimum per run.   usage: bash tools/ab_k2.sh [tag ...]
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
tags=${@:-$(ls $root/slam.net_amd/build/variants/*.so | xargs -n1 basename | sed 's/.so$//')}
for rep in 1 2 3; do
  for t in $tags; do
    out=/tmp/ab_${t}_${rep}; rm -rf $out
    SLAMHIP_LIB=$root/slam.net_amd/build/variants/$t.so timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out -o s -- python3 $root/tools/${AB_SCRIPT:-prof_k2.py} > /dev/null 2>&1
    python3 - "$out" "$t" "$rep" <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k2_pixels" in r["Name"] or "k1_search" in r["Name"]: print("%-14s run %s: " % (sys.argv[2], sys.argv[3]) + r["Name"][:16] + " avg %.2f us  min %.2f us  (%s calls)" % (float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, r["Calls"]))
PY
  done
done
