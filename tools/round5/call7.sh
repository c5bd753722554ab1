#!/bin/bash
# round 5, GPU call 7: the census by candidate list (one call site of the second look) against the sweep and against no census
O=gpurun_out/r5g; mkdir -p $O
timeout 500 python3 tools/ab_headline.py 1000 smooth nodetect:exact_ties=0,tie_detect=0 default_list: default_sweep:tie_list=0 > $O/ab_census.log 2>&1
DSA_AB_REPS=2 timeout 300 python3 tools/ab_headline.py 1000 checker mode0_list:exact_ties=0 mode0_sweep:exact_ties=0,tie_list=0 >> $O/ab_census.log 2>&1
cat $O/ab_census.log
timeout 600 python3 -m pytest tests/test_gpu_bundles.py -m gpu -q -k "census or generation" > $O/tests.log 2>&1; tail -5 $O/tests.log | cut -c1-300
