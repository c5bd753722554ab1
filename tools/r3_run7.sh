#!/bin/bash
O=gpurun_out/r3_run7; mkdir -p $O
timeout 900 python3 tools/exact_probe.py 131 2048 checker 768 0 > $O/exact_probe_checker.log 2>&1; cat $O/exact_probe_checker.log
timeout 900 python3 tools/exact_probe.py 131 1024 rough 768 0 > $O/exact_probe_rough.log 2>&1; cat $O/exact_probe_rough.log
timeout 2400 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1; echo "pytest gpu rc $?"; tail -4 $O/pytest_gpu.log | cut -c1-250
