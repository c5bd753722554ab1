#!/bin/bash
mkdir -p gpurun_out/r03_bundle
{
for w in 0.4 0.5 0.6 0.7; do echo "== window $w"; DSA_PROBE_WINDOW=$w timeout 600 python3 tools/bundle_probe.py time 131 1000 16 smooth 16; done
} > gpurun_out/r03_bundle/probe6.log 2>&1
cat gpurun_out/r03_bundle/probe6.log | cut -c1-330
