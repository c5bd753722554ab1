#!/bin/bash
# round 5, GPU call 2: same-box A/B of the bundle kernel's changes, the two-point bytes experiment, the census on the checkerboard, the GPU suite
O=gpurun_out/r5b; mkdir -p $O
AB=dsurftomo_amd/build/ab
DSA_LIB_PATH=$AB/lib_r4.so timeout 300 python3 tools/ab_headline.py 1000 smooth r4_library: > $O/ab_r4.log 2>&1
timeout 600 python3 tools/ab_headline.py 1000 smooth mode0_nodetect:exact_ties=0,tie_detect=0 mode0_farall:exact_ties=0,tie_detect=0,bundle_far_all=1 mode0_census:exact_ties=0 default: \
    tailwide:bundle_tail=1 tailwide_mode0_nodetect:exact_ties=0,tie_detect=0,bundle_tail=1 > $O/ab_new.log 2>&1
DSA_LIB_PATH=$AB/lib_stride2.so timeout 400 python3 tools/ab_headline.py 1000 smooth stride2_nodetect:exact_ties=0,tie_detect=0 stride2_farall:exact_ties=0,tie_detect=0,bundle_far_all=1 > $O/ab_stride2.log 2>&1
DSA_AB_REPS=1 timeout 600 python3 tools/ab_headline.py 1000 checker mode0_census:exact_ties=0 default: > $O/ab_checker.log 2>&1
cat $O/ab_r4.log $O/ab_new.log $O/ab_stride2.log $O/ab_checker.log
timeout 1700 python3 -m pytest tests -m gpu -q > $O/tests.log 2>&1
echo "tests rc=$?"; tail -40 $O/tests.log | cut -c1-400
