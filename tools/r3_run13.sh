#!/bin/bash
O=gpurun_out/r3_run13; mkdir -p $O
timeout 900 python3 tests/tools/fuzz_dispersion.py 40 23 > $O/fuzz_dispersion.log 2>&1; tail -3 $O/fuzz_dispersion.log | cut -c1-250
timeout 900 python3 tests/tools/fuzz_boundary.py 30 11 > $O/fuzz_boundary.log 2>&1; tail -3 $O/fuzz_boundary.log | cut -c1-250
timeout 1200 python3 tests/tools/config4_probe.py 1536 8 2 > $O/config4_one_gpu_share.log 2>&1; cat $O/config4_one_gpu_share.log | cut -c1-400
python3 bench.py --steps 5 --warmup 2 > $O/bench.log 2> $O/bench.err; tail -1 $O/bench.log | cut -c1-300
