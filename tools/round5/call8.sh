#!/bin/bash
# round 5, GPU call 8: the march in pooled tiles -- tests, then rates at 1025^2 and 4097^2
O=gpurun_out/r5h; mkdir -p $O
timeout 900 python3 -m pytest tests/test_gpu_exact.py -m gpu -q -x -k "pooled" > $O/tests_tiles.log 2>&1; echo "tiles tests rc=$?"; tail -12 $O/tests_tiles.log | cut -c1-400
timeout 300 python3 tools/exact_tiles_probe.py 131 256 16 checker -1,1 > $O/rate_1025.log 2>&1; cat $O/rate_1025.log
timeout 900 python3 tools/exact_tiles_probe.py 515 128 24 checker -1,1 > $O/rate_4097.log 2>&1; cat $O/rate_4097.log
timeout 600 python3 tools/exact_tiles_probe.py 515 400 24 checker 0 >> $O/rate_4097.log 2>&1; tail -1 $O/rate_4097.log
