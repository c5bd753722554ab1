#!/bin/bash
O=gpurun_out/r3_run16; mkdir -p $O
timeout 900 python3 -m pytest tests/test_gpu_exact.py -x -q -k "literal" > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -2 $O/pytest.log | cut -c1-200
DSA_PROBE_SKIP_THRESHOLDS=1 timeout 600 python3 tools/exact_probe.py 131 4096 checker 768,640 0 2>&1 | grep -E "fixed point  |exact_ties=2" | tee $O/exact_probe_4096.log
