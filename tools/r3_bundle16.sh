#!/bin/bash
O=gpurun_out/r03_bundle; mkdir -p $O
{
timeout 900 python3 -m pytest tests/test_gpu_bundles.py -x -q 2>&1 | tail -2
for l in base shadow base shadow; do echo "== lib $l"; DSA_LIB_PATH=dsurftomo_amd/build/ab/lib_$l.so timeout 900 python3 tools/bundle_probe.py time 131 1000 16 smooth 16 | cut -c1-330; done
} > $O/ab_shadow.log 2>&1
cat $O/ab_shadow.log
