#!/bin/bash
# round 5, GPU call 11: the refined boxes in bundles
O=gpurun_out/r5k; mkdir -p $O
timeout 900 python3 -m pytest tests/test_gpu_bundles.py -m gpu -q -x -k "refined_boxes" > $O/tests.log 2>&1; echo "tests rc=$?"; tail -12 $O/tests.log | cut -c1-400
timeout 400 python3 tools/ab_headline.py 1000 smooth refined_solo_nodetect:exact_ties=0,tie_detect=0,bundle_refined=0 refined_bundled_nodetect:exact_ties=0,tie_detect=0 default: > $O/ab.log 2>&1; cat $O/ab.log
timeout 300 python3 tools/ab_headline.py 125 smooth refined_solo:bundle_refined=0 default: >> $O/ab.log 2>&1; tail -2 $O/ab.log
