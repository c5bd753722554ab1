#!/bin/bash
# round 3, GPU run 2: whole GPU suite on the current tree; A/B of the solver's first-step form; exact mode with more units in flight
O=gpurun_out/r3_run2; mkdir -p $O
timeout 2400 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1; echo "pytest gpu rc $?"
tail -60 $O/pytest_gpu.log | cut -c1-250
bash tools/run_ab.sh r3_run2/ab "base step1 base step1" 1.25 4096 smooth 2>&1 | cut -c1-200
timeout 900 python3 tools/exact_probe.py 131 4096 checker 1024,2048 0 > $O/exact_probe_checker_4096.log 2>&1; cat $O/exact_probe_checker_4096.log
