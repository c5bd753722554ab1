#!/bin/bash
# round 5, GPU call 5: the two-stream pipeline, the census by candidate list, bundle / exact tests
O=gpurun_out/r5e; mkdir -p $O
timeout 600 python3 tools/ab_headline.py 1000 smooth nopipe_nodetect:exact_ties=0,tie_detect=0,bundle_pipeline=0 pipe_nodetect:exact_ties=0,tie_detect=0 default: default_sweep:tie_list=0 default_nopipe:bundle_pipeline=0 > $O/ab_pipe.log 2>&1
cat $O/ab_pipe.log
DSA_AB_REPS=1 timeout 300 python3 tools/ab_headline.py 1000 checker mode0_list:exact_ties=0 mode0_sweep:exact_ties=0,tie_list=0 >> $O/ab_checker.log 2>&1
cat $O/ab_checker.log
for n in 500 700; do timeout 300 python3 tools/ab_headline.py $n smooth default: nopipe:bundle_pipeline=0 >> $O/ab_shares.log 2>&1; done
cat $O/ab_shares.log
timeout 1500 python3 -m pytest tests/test_gpu_bundles.py tests/test_gpu_exact.py -m gpu -q > $O/tests.log 2>&1
echo "tests rc=$?"; tail -8 $O/tests.log | cut -c1-300
grep -n "census\|pipeline" $O/tests.log | cut -c1-300
