#!/bin/bash
# rocprofv3 passes over the headline benchmark (run on the GPU box, e.g. gpurun -- 'bash tools/prof_bench.sh r03'):
#   stats      --kernel-trace --stats            per-kernel durations (the K1 average must agree with bench.py's own HIP-event figure)
#   stats262k  the same at 262 144 candidates per launch (the size at which K1 passes half the HBM roofline)
#   fetch / write / tcc   --pmc FETCH_SIZE | WRITE_SIZE | TCC_HIT TCC_MISS: HBM-side traffic of the dominant kernel, one counter
#              set per pass (MI355X_MICROARCH.md: FETCH_SIZE costs 3 of the 4 TCC slots)
#   sq1 / sq2  --pmc SQ_* (8 SQ slots per pass): waves, busy / wave cycles, VALU / SALU / LDS / VMEM instruction counts, wait
#              cycles, LDS bank conflicts -- the VALU-bound / latency-bound split of K1
#   k2*        the same counters for the HoleMap update (tools/prof_k2.py: 40 updates at 2048^2)
# Counter passes never carry --stats or any trace domain but --kernel-trace (gpurun refuses such combinations).
# Output: gpurun_out/<tag>/prof/*; tools/prof_summary.py turns it into profiles/<tag>_*.
tag=${1:-r03}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/$tag/prof
rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
bench="python3 $root/bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-extras"
SQ1="SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY"
SQ2="SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM"
run() { name=$1; shift; timeout 300 rocprofv3 "$@" > $out/$name.out 2> $out/$name.log; }
# the plain run FIRST: counter passes put the device into the profiler's power state, and what runs on the box for a while afterwards
# clocks differently (round 6: a native-caller Update of 48.7 us behind the passes, 41.3 - 42.4 us on fresh boxes)
(cd $root && timeout 600 python3 bench.py > $out/bench.json 2> $out/bench.err)
run stats     --kernel-trace --stats --output-format csv -d $out/stats -o stats -- $bench
run stats262k --kernel-trace --stats --output-format csv -d $out/stats262k -o stats -- $bench --cands 262144 --steps 50
run fetch     --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/fetch -o p -- $bench
run write     --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/write -o p -- $bench
run tcc       --kernel-trace --pmc TCC_HIT TCC_MISS --output-format csv -d $out/tcc -o p -- $bench
run sq1       --kernel-trace --pmc $SQ1 --output-format csv -d $out/sq1 -o p -- $bench
run sq2       --kernel-trace --pmc $SQ2 --output-format csv -d $out/sq2 -o p -- $bench
run grbm      --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv -d $out/grbm -o p -- $bench
run grbm_262k --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv -d $out/grbm_262k -o p -- $bench --cands 262144 --steps 50
run sq1_262k  --kernel-trace --pmc $SQ1 --output-format csv -d $out/sq1_262k -o p -- $bench --cands 262144 --steps 50
run sq2_262k  --kernel-trace --pmc $SQ2 --output-format csv -d $out/sq2_262k -o p -- $bench --cands 262144 --steps 50
k2="python3 $root/tools/prof_k2.py"
run k2stats   --kernel-trace --stats --output-format csv -d $out/k2stats -o stats -- $k2
run k2fetch   --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/k2fetch -o p -- $k2
run k2write   --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/k2write -o p -- $k2
run k2tcc     --kernel-trace --pmc TCC_HIT TCC_MISS --output-format csv -d $out/k2tcc -o p -- $k2
run k2sq1     --kernel-trace --pmc $SQ1 --output-format csv -d $out/k2sq1 -o p -- $k2
run k2sq2     --kernel-trace --pmc $SQ2 --output-format csv -d $out/k2sq2 -o p -- $k2
# the Hector kernels: the grid update (30 updates of the 3-level 2048^2 pyramid), the single match (a latency chain) and the batched one
k5="python3 $root/tools/prof_k5.py"
k4="python3 $root/tools/prof_k4.py single"
k4b="python3 $root/tools/prof_k4.py batch"
run k5stats   --kernel-trace --stats --output-format csv -d $out/k5stats -o stats -- $k5
run k5fetch   --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/k5fetch -o p -- $k5
run k5write   --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/k5write -o p -- $k5
run k5sq1     --kernel-trace --pmc $SQ1 --output-format csv -d $out/k5sq1 -o p -- $k5
run k5sq2     --kernel-trace --pmc $SQ2 --output-format csv -d $out/k5sq2 -o p -- $k5
run k4stats   --kernel-trace --stats --output-format csv -d $out/k4stats -o stats -- $k4
run k4fetch   --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/k4fetch -o p -- $k4
run k4write   --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/k4write -o p -- $k4
run k4sq1     --kernel-trace --pmc $SQ1 --output-format csv -d $out/k4sq1 -o p -- $k4
run k4sq2     --kernel-trace --pmc $SQ2 --output-format csv -d $out/k4sq2 -o p -- $k4
run k4bstats  --kernel-trace --stats --output-format csv -d $out/k4bstats -o stats -- $k4b
# the fused scan back to back (C3) and CoreSLAMProcessor.Update from the native caller: kernel timelines (tools/trace_gaps.py)
run c3trace   --kernel-trace --output-format csv -d $out/c3trace -o t -- python3 $root/tools/prof_c3.py 300
gcc -O1 -o /tmp/abi_harness $root/tests/abi_harness.c -ldl -lm
# (the ordinary order -- scan tables, then the search launch -- shows the kernels and the idle gap the launch-ahead flow removes; under the
# profiler the host is slower, so in the default flow the prelaunched search is seen waiting for its tables: kept beside it)
SLAMHIP_PRELAUNCH=0 run proctrace --kernel-trace --output-format csv -d $out/proctrace -o t -- /tmp/abi_harness $root/slam.net_amd/libslamhip.so --bench-proc 2048 1080 16385 300
run proctrace_la --kernel-trace --output-format csv -d $out/proctrace_la -o t -- /tmp/abi_harness $root/slam.net_amd/libslamhip.so --bench-proc 2048 1080 16385 300
run hstrace   --kernel-trace --output-format csv -d $out/hstrace -o t -- python3 $root/tools/exp.py hsproc 2048 3 1080 0
cd $root
python3 tools/trace_gaps.py $out/hstrace > $out/timeline_hsproc.txt 2>&1
python3 tools/trace_gaps.py $out/c3trace > $out/timeline_c3.txt 2>&1
python3 tools/trace_gaps.py $out/proctrace > $out/timeline_csproc.txt 2>&1
python3 tools/trace_gaps.py $out/proctrace_la > $out/timeline_csproc_launch_ahead.txt 2>&1
python3 tools/kernels_bench.py > $out/kernels_bench.json 2> $out/kernels_bench.err
# keep what travels back small: the per-dispatch counter CSVs are summarised here, the raw files stay on the box
python3 tools/prof_summary.py $tag --collect
find $out -name "*.csv" -size +200k -delete
ls $out | head -40
