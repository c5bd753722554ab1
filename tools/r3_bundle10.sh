#!/bin/bash
O=gpurun_out/r03_bundle; mkdir -p $O
{
timeout 900 python3 tools/bundle_probe.py time 257 256 8 smooth 0,8,4
timeout 1500 python3 tools/bundle_probe.py time 513 128 8 checker 0,8,4
timeout 600 python3 tools/bundle_probe.py time 131 512 16 smooth 16
} > $O/probe10.log 2>&1
cut -c1-420 $O/probe10.log
