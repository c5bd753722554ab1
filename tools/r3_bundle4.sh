#!/bin/bash
mkdir -p gpurun_out/r03_bundle
{
timeout 900 python3 -m pytest tests/test_gpu_bundles.py -x -q 2>&1 | tail -25
for l in base bw3; do echo "== lib $l"; DSA_PROBE_ROUNDS=1 DSA_LIB_PATH=dsurftomo_amd/build/ab/lib_$l.so timeout 900 python3 tools/bundle_probe.py time 131 1000 16 smooth 16,8; done
echo "== solo rounds"; DSA_PROBE_ROUNDS=1 timeout 900 python3 tools/bundle_probe.py time 131 1000 16 smooth 0
} > gpurun_out/r03_bundle/probe4.log 2>&1
cat gpurun_out/r03_bundle/probe4.log | cut -c1-420
