#!/bin/bash
O=gpurun_out/r03_bundle; mkdir -p $O
{
for l in base dyn base dyn; do echo "== lib $l"; DSA_LIB_PATH=dsurftomo_amd/build/ab/lib_$l.so timeout 900 python3 tools/bundle_probe.py time 131 1000 16 smooth 16 | cut -c1-330; done
} > $O/ab_dyn.log 2>&1
cat $O/ab_dyn.log
