#!/bin/bash
O=gpurun_out/r3_run4; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_fullsize.py -x -q -k "recycled" > $O/pytest_recycle.log 2>&1; echo "pytest recycle rc $?"; tail -5 $O/pytest_recycle.log | cut -c1-220
timeout 2400 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1; echo "pytest gpu rc $?"; tail -4 $O/pytest_gpu.log | cut -c1-220
timeout 900 python3 bench.py --steps 2 --warmup 1 > $O/bench.json 2> $O/bench.err; echo "bench rc $?"; cat $O/bench.json | cut -c1-1500; tail -3 $O/bench.err
timeout 600 python3 tools/exact_probe.py 131 4096 checker 256,384,512,640,768 0 > $O/exact_probe_lds.log 2>&1; grep -v tie_threshold $O/exact_probe_lds.log
