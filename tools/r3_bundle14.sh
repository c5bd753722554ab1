#!/bin/bash
O=gpurun_out/r03_bundle; mkdir -p $O
{
for t in 256 128 64; do for g in 16 8; do echo "== threads $t, members $g"; DSA_PROBE_BPOOL=2048 DSA_PROBE_BTHREADS=$t timeout 600 python3 tools/bundle_probe.py time 131 1000 16 smooth $g | cut -c1-330; done; done
} > $O/ab_threads2.log 2>&1
cat $O/ab_threads2.log
