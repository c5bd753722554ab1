#!/bin/bash
# round 3, GPU run 1: exact mode tests + probes, and the solver change against the round-2 bits
O=gpurun_out/r3_run1; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_exact.py -x -q > $O/pytest_exact.log 2>&1; echo "pytest exact rc $?" 
tail -25 $O/pytest_exact.log
timeout 600 python3 tools/exact_probe.py 131 512 checker 1024,2048,4096 0 > $O/exact_probe_checker.log 2>&1; cat $O/exact_probe_checker.log
timeout 600 python3 tools/exact_probe.py 131 512 smooth 2048 0 > $O/exact_probe_smooth.log 2>&1; cat $O/exact_probe_smooth.log
timeout 600 python3 tools/perf_probe.py 131 2048 1.25 smooth 256 > $O/perf_probe.log 2>&1; cat $O/perf_probe.log | cut -c1-260
timeout 600 python3 tools/variant_check.py 131 64 smooth > $O/variant_check.log 2>&1; tail -7 $O/variant_check.log | cut -c1-200
