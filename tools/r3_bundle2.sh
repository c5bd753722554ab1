#!/bin/bash
mkdir -p gpurun_out/r03_bundle
{
timeout 1500 python3 tools/bundle_probe.py time 131 1000 16 smooth 0,16,8,4
timeout 600 python3 tools/bundle_probe.py time 131 512 16 checker 0,16,8
timeout 600 python3 tools/bundle_probe.py time 131 512 16 rough 0,16,8
} > gpurun_out/r03_bundle/probe2.log 2>&1
cat gpurun_out/r03_bundle/probe2.log | cut -c1-500
