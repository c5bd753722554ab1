#!/bin/bash
# round 5, GPU call 4: the pipeline of groups, the unrelated-maps leg after the far-load fallback, the two-point experiment on the current tree, the GPU suite
O=gpurun_out/r5d; mkdir -p $O
AB=dsurftomo_amd/build/ab
timeout 500 python3 tools/ab_headline.py 1000 smooth nopipe_nodetect:exact_ties=0,tie_detect=0,bundle_pipeline=0 pipe_nodetect:exact_ties=0,tie_detect=0 default: default_nopipe:bundle_pipeline=0 > $O/ab_pipe.log 2>&1
cat $O/ab_pipe.log
timeout 400 python3 tools/ab_headline.py 256 rough auto:exact_ties=0 farall:exact_ties=0,bundle_far_all=1 solo:exact_ties=0,bundle=0 > $O/ab_rough.log 2>&1
cat $O/ab_rough.log
DSA_LIB_PATH=$AB/lib_stride2.so timeout 400 python3 tools/ab_headline.py 1000 smooth stride2_nodetect:exact_ties=0,tie_detect=0 > $O/ab_stride2.log 2>&1
cat $O/ab_stride2.log
for n in 125 250 500; do timeout 300 python3 tools/ab_headline.py $n smooth default: nopipe:bundle_pipeline=0 >> $O/ab_shares.log 2>&1; done
cat $O/ab_shares.log
timeout 1900 python3 -m pytest tests -m gpu -q > $O/tests.log 2>&1
echo "tests rc=$?"; tail -15 $O/tests.log | cut -c1-300
grep -n "pipeline\|beyond the first" $O/tests.log | cut -c1-300
