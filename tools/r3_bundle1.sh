#!/bin/bash
# first GPU run of the bundle kernel: parity with unit-by-unit solves on small grids, then throughput at the headline size
mkdir -p gpurun_out/r03_bundle
{
timeout 300 python3 tools/bundle_probe.py check 33 24 16 smooth
timeout 300 python3 tools/bundle_probe.py check 33 24 16 mixed
timeout 300 python3 tools/bundle_probe.py check 33 20 5 rough 0,16,4
timeout 900 python3 tools/bundle_probe.py time 131 256 16 smooth 0,16,8,4
} > gpurun_out/r03_bundle/probe1.log 2>&1
cat gpurun_out/r03_bundle/probe1.log | cut -c1-700
