#!/bin/bash
mkdir -p gpurun_out/r03_bundle
{
timeout 900 python3 tools/bundle_probe.py time 131 128 16 rough 0,-1,16,8
timeout 900 python3 tools/bundle_probe.py time 131 128 16 checker 0,-1,16,8
timeout 900 python3 tools/bundle_probe.py time 65 128 16 rough 0,-1,16,8
} > gpurun_out/r03_bundle/probe3.log 2>&1
cat gpurun_out/r03_bundle/probe3.log | cut -c1-600
