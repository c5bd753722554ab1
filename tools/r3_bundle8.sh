#!/bin/bash
# whole boundary call at the headline size with bundles (default) and without; the inversion step; determinism of the bundle kernel
O=gpurun_out/r03_bundle; mkdir -p $O
{
echo "== headline boundary, bundles automatic"; timeout 900 python3 tests/tools/headline_boundary.py 8 1000 2>&1 | tail -12
echo "== headline boundary, DSA_BUNDLE=0"; DSA_BUNDLE=0 timeout 900 python3 tests/tools/headline_boundary.py 8 1000 2>&1 | tail -12
} > $O/headline.log 2>&1
cut -c1-300 $O/headline.log
{
for rep in 1 2 3; do timeout 600 python3 tools/bundle_probe.py time 131 256 16 rough 16 2>&1 | tail -1 | cut -c1-200; done
} > $O/determinism.log 2>&1
cat $O/determinism.log
