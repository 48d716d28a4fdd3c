set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
python bench.py > gpurun_out/bench_r01b.json 2> gpurun_out/bench_r01b.err; tail -1 gpurun_out/bench_r01b.json | cut -c1-1500
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/p_stats -- python3 bench.py --steps 100 --no-cpu-baseline > gpurun_out/p_stats.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/p_fetch -- python3 bench.py --steps 100 --no-cpu-baseline > gpurun_out/p_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/p_write -- python3 bench.py --steps 100 --no-cpu-baseline > gpurun_out/p_write.log 2>&1
rocprofv3 --kernel-trace --pmc TCC_HIT TCC_MISS --output-format csv -d gpurun_out/p_tcc -- python3 bench.py --steps 100 --no-cpu-baseline > gpurun_out/p_tcc.log 2>&1
find gpurun_out/p_stats -name "*kernel_stats.csv" | head -2
python tools/pmc_summary.py fetch=$(dirname $(find gpurun_out/p_fetch -name "*counter_collection.csv" | head -1)) write=$(dirname $(find gpurun_out/p_write -name "*counter_collection.csv" | head -1)) tcc=$(dirname $(find gpurun_out/p_tcc -name "*counter_collection.csv" | head -1)) > gpurun_out/pmc_r01b.json
head -c 1500 gpurun_out/pmc_r01b.json
