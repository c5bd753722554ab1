#!/bin/bash
O=gpurun_out/r03_bundle; mkdir -p $O
{
echo "== whole drop-in calls with stations (the same sources at every period slot) and DSA_BUNDLE=4, against the oracle"
DSA_BUNDLE=4 DSA_FUZZ_STATIONS=1 timeout 1200 python3 tests/tools/fuzz_boundary.py 30 31 2>&1 | tail -32 | cut -c1-220
DSA_BUNDLE=8 DSA_FUZZ_STATIONS=1 timeout 1500 python3 tests/tools/fuzz_boundary.py 8 32 big 2>&1 | tail -10 | cut -c1-220
} > $O/fuzz_boundary_bundles.log 2>&1
tail -14 $O/fuzz_boundary_bundles.log
