#!/bin/bash
mkdir -p gpurun_out/r03_bundle
{
timeout 900 python3 tools/bundle_probe.py time 257 256 8 smooth 0,8,4
timeout 900 python3 tools/bundle_probe.py time 513 128 4 checker 0,4
timeout 900 python3 tools/bundle_probe.py time 65 512 16 smooth 0,16,8,4
timeout 900 python3 tools/bundle_probe.py time 97 512 16 smooth 0,16,8,4
} > gpurun_out/r03_bundle/probe8.log 2>&1
cat gpurun_out/r03_bundle/probe8.log | cut -c1-420
