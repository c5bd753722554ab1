#!/bin/bash
O=gpurun_out/r3_run3; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_exact.py tests/test_gpu_sharded.py -x -q > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -5 $O/pytest.log | cut -c1-200
timeout 900 python3 tools/exact_probe.py 131 4096 checker 768,1024,1536 0 > $O/exact_probe_checker_4096.log 2>&1; cat $O/exact_probe_checker_4096.log
