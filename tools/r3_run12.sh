#!/bin/bash
O=gpurun_out/r3_run12; mkdir -p $O
python3 bench.py --steps 5 --warmup 2 > $O/bench.log 2> $O/bench.err; tail -1 $O/bench.log | cut -c1-1500
timeout 1500 python3 -m pytest tests/test_gpu_exact.py tests/test_gpu_fullsize.py -x -q > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -2 $O/pytest.log | cut -c1-200
