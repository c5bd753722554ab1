#!/bin/bash
O=gpurun_out/r03_bundle; mkdir -p $O
{
timeout 900 python3 -m pytest tests/test_gpu_bundles.py -x -q 2>&1 | tail -2
echo "== the cycle case: 500 sources x 16, bundles of 16, window 1.5 (461 ms / rounds max 4936 before)"; DSA_PROBE_ROUNDS=1 DSA_PROBE_BWINDOW=1.5 timeout 600 python3 tools/bundle_probe.py time 131 500 16 smooth 0,16 | cut -c1-400
echo "== the same at window 2.0 and 1.0"; DSA_PROBE_BWINDOW=2.0 timeout 600 python3 tools/bundle_probe.py time 131 500 16 smooth 16 | cut -c1-300; DSA_PROBE_BWINDOW=1.0 timeout 600 python3 tools/bundle_probe.py time 131 500 16 smooth 16 | cut -c1-300
echo "== headline"; DSA_PROBE_ROUNDS=1 timeout 600 python3 tools/bundle_probe.py time 131 1000 16 smooth 0,16 | cut -c1-400
echo "== rough"; timeout 600 python3 tools/bundle_probe.py time 131 512 16 rough 0,16 | cut -c1-400
echo "== wild (unrelated +-45 % maps per period)"; DSA_DEBUG_BUNDLE=1 timeout 600 python3 tools/bundle_probe.py time 131 128 16 wild 0,16 | cut -c1-400
} > $O/stale_pullback.log 2>&1
cat $O/stale_pullback.log
