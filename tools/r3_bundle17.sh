#!/bin/bash
O=gpurun_out/r03_bundle; mkdir -p $O
{
for w in 0.6 1.0 1.5 2.0 3.0; do echo "== 125 sources x 16, bundles of 4, window $w"; DSA_PROBE_BWINDOW=$w timeout 600 python3 tools/bundle_probe.py time 131 125 16 smooth 4 | cut -c1-200; done
for w in 0.6 1.0 1.5 2.0; do echo "== 250 sources x 16, bundles of 8, window $w"; DSA_PROBE_BWINDOW=$w timeout 600 python3 tools/bundle_probe.py time 131 250 16 smooth 8 | cut -c1-200; done
for w in 0.6 1.0 1.5; do echo "== 500 sources x 16, bundles of 16, window $w"; DSA_PROBE_BWINDOW=$w timeout 600 python3 tools/bundle_probe.py time 131 500 16 smooth 16 | cut -c1-200; done
} > $O/small_share_windows.log 2>&1
cat $O/small_share_windows.log
