#!/bin/bash
mkdir -p gpurun_out/r03_bundle
{
timeout 900 python3 -m pytest tests/test_gpu_bundles.py -x -q 2>&1 | tail -22
DSA_PROBE_ROUNDS=1 timeout 900 python3 tools/bundle_probe.py time 131 1000 16 smooth 0,16,8
DSA_PROBE_ROUNDS=1 timeout 900 python3 tools/bundle_probe.py time 131 512 16 rough 0,16,8
DSA_PROBE_ROUNDS=1 timeout 900 python3 tools/bundle_probe.py time 131 512 16 checker 0,16
} > gpurun_out/r03_bundle/probe7.log 2>&1
cat gpurun_out/r03_bundle/probe7.log | cut -c1-520
