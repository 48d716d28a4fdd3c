#!/bin/bash
# The plan launch A/B (bench.py with SLAMHIP_K1_PLAN=0 against the default, 2000-step and 20-step regions, three rounds on one box) and two
# soak seeds, the second with a plan forced for every launch.   gpurun -- bash tools/ab_plan.sh   ->  gpurun_out/ab_plan/ab.txt
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/ab_plan; mkdir -p $out; : > $out/ab.txt
cd $root
b() { python3 bench.py "$@" --no-cpu-baseline --no-extras 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.3f' % (d['ms_per_step']*1e3))"; }
for rep in 1 2 3; do
  echo "plan1 2000 $(b --steps 2000 --warmup 50)" >> $out/ab.txt
  echo "plan0 2000 $(SLAMHIP_K1_PLAN=0 b --steps 2000 --warmup 50)" >> $out/ab.txt
  echo "plan1 20 $(b --steps 20 --warmup 5)" >> $out/ab.txt
  echo "plan0 20 $(SLAMHIP_K1_PLAN=0 b --steps 20 --warmup 5)" >> $out/ab.txt
done
timeout 400 python3 tests/fuzz_parity.py --seconds 300 --seed 6101 > $out/soak_6101.txt 2>&1; echo "soak 6101 rc $?" >> $out/ab.txt; tail -2 $out/soak_6101.txt >> $out/ab.txt
SLAMHIP_K1_PLAN_ALWAYS=1 timeout 400 python3 tests/fuzz_parity.py --seconds 300 --seed 6102 > $out/soak_6102.txt 2>&1; echo "soak 6102 (PLAN_ALWAYS) rc $?" >> $out/ab.txt; tail -2 $out/soak_6102.txt >> $out/ab.txt
