#!/bin/bash
O=gpurun_out/r03_bundle; mkdir -p $O
{
for ns in 125 250; do for t in 256 128 64; do echo "== $ns sources x 16 periods, threads $t"; DSA_PROBE_BPOOL=2048 DSA_PROBE_BTHREADS=$t timeout 600 python3 tools/bundle_probe.py time 131 $ns 16 smooth 0,16,8,4 | cut -c1-200; done; done
} > $O/small_shares.log 2>&1
cat $O/small_shares.log
