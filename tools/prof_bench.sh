#!/bin/bash
# rocprofv3 passes over the headline benchmark (run on the GPU box, e.g. gpurun -- 'bash tools/prof_bench.sh r02'):
#   1. --kernel-trace --stats            per-kernel durations (the K1 average must agree with bench.py's own HIP-event figure)
#   2. --kernel-trace --pmc FETCH_SIZE   \  HBM-side traffic of the dominant kernel, one counter set per pass
#   3. --kernel-trace --pmc WRITE_SIZE   /  (MI355X_MICROARCH.md: FETCH_SIZE costs 3 of the 4 TCC slots)
#   4. --kernel-trace --pmc TCC_HIT TCC_MISS
# Output: gpurun_out/<tag>/prof/{stats,fetch,write,tcc}; tools/prof_summary.py turns it into profiles/<tag>_*.
tag=${1:-r02}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/$tag/prof
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
cmd="python3 $root/bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-extras"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -o stats -- $cmd > $out/stats_bench.json 2> $out/stats.log
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/fetch -o fetch -- $cmd > $out/fetch_bench.json 2> $out/fetch.log
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/write -o write -- $cmd > $out/write_bench.json 2> $out/write.log
rocprofv3 --kernel-trace --pmc TCC_HIT TCC_MISS --output-format csv -d $out/tcc -o tcc -- $cmd > $out/tcc_bench.json 2> $out/tcc.log
cd $root
python3 bench.py > $out/bench.json 2> $out/bench.err
find $out -name "*.csv" | head -20
