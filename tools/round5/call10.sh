#!/bin/bash
# round 5, GPU call 10: causal windows of small wide launches (a rank's share of the headline call)
O=gpurun_out/r5j; mkdir -p $O
export DSA_AB_REPS=2
timeout 300 python3 tools/ab_headline.py 125 smooth auto: w175:bundle_window_cells=1.75 w250:bundle_window_cells=2.5 w400:bundle_window_cells=4.0 g16w125:bundle=16,bundle_window_cells=1.25 g4:bundle=4 > $O/ab_125.log 2>&1
timeout 300 python3 tools/ab_headline.py 250 smooth auto: w100:bundle_window_cells=1.0 w150:bundle_window_cells=1.5 w250:bundle_window_cells=2.5 >> $O/ab_125.log 2>&1
timeout 300 python3 tools/ab_headline.py 1000 smooth auto:exact_ties=0,tie_detect=0 w08:exact_ties=0,tie_detect=0,bundle_window_cells=0.8 w05:exact_ties=0,tie_detect=0,bundle_window_cells=0.5 >> $O/ab_125.log 2>&1
cat $O/ab_125.log
