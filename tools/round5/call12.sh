#!/bin/bash
# round 5, GPU call 12: the final tree -- GPU suite (with durations), rocprofv3 trace + counters of the bench command, the bench line
O=gpurun_out/r5l; mkdir -p $O
timeout 2400 python3 -m pytest tests -m gpu -q --durations=12 > $O/tests.log 2>&1; echo "tests rc=$?"; grep -A14 "slowest" $O/tests.log | cut -c1-200; tail -3 $O/tests.log | cut -c1-300
DSA_PMC_STEPS=1 timeout 2000 bash tools/profile_bench.sh r5l_prof > $O/profile.log 2>&1; tail -16 gpurun_out/r5l_prof/summary.txt | cut -c1-260
timeout 900 python3 bench.py --steps 5 --warmup 2 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; cut -c1-300 $O/bench.json
