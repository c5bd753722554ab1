#!/bin/bash
O=gpurun_out/r03_bundle; mkdir -p $O
{
for p in 1000 768 576 528 0; do echo "== bundle slots $p"; DSA_PROBE_BPOOL=$p timeout 600 python3 tools/bundle_probe.py time 131 1000 16 smooth 0,16 | tail -1 | cut -c1-330; done
timeout 600 python3 -m pytest tests/test_gpu_bundles.py -x -q 2>&1 | tail -1
} > $O/slot_order.log 2>&1
cat $O/slot_order.log
