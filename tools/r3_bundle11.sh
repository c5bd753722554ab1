#!/bin/bash
O=gpurun_out/r03_bundle; mkdir -p $O
{
timeout 900 python3 -m pytest tests/test_gpu_bundles.py -x -q 2>&1 | tail -3
python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-330
echo "== configs[4] medium, one GPU's share in miniature: 4097^2 checkerboard, 256 sources x 24 periods (automatic, then unit by unit)"
timeout 2400 python3 tools/bundle_probe.py time 513 256 24 checker 1,0
} > $O/probe11.log 2>&1
cut -c1-420 $O/probe11.log
