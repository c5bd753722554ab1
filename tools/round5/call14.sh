#!/bin/bash
# round 5, GPU call 14: refined boxes with 128-thread workgroups, first-generation / tail / refined windows; counters of the march at full load
O=gpurun_out/r5n; mkdir -p $O
DSA_AB_REPS=4 timeout 600 python3 tools/ab_headline.py 1000 smooth base: r128:bundle_refined_threads=128 \
  w05:bundle_window_cells=0.5,bundle_window_tail_cells=1.0 w07:bundle_window_cells=0.7,bundle_window_tail_cells=1.0 w08:bundle_window_cells=0.8,bundle_window_tail_cells=1.0 \
  t075:bundle_window_tail_cells=0.75 t125:bundle_window_tail_cells=1.25 t15:bundle_window_tail_cells=1.5 \
  rw08:window_cells=0.8 rw2:window_cells=2.0 rw128_2:window_cells=2.0,bundle_refined_threads=128 base2: > $O/ab.log 2>&1
echo "ab rc=$?"; cut -c1-230 $O/ab.log
timeout 900 bash tools/collect_pmc.sh r5n_xpmc insts,busy,fetch,write,wait - -- python3 tools/exact_rate.py 131 16000 checker 0 0 16 > $O/xpmc.log 2>&1
echo "xpmc rc=$?"; grep -v "^W2026\|^E2026" gpurun_out/r5n_xpmc/summary.txt | grep "^#\|^==\|N=\|xmarch<false" | cut -c1-200
