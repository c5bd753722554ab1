#!/bin/bash
O=gpurun_out/r3_run5; mkdir -p $O
bash tools/run_ab.sh r3_run5/ab "km0 km1 km0 km1" 1.25 8192 smooth 2>&1 | cut -c1-200
bash tools/run_ab.sh r3_run5/abc "km0 km1" 1.25 2048 checker 2>&1 | cut -c1-200
timeout 900 python3 tools/variant_check.py 131 64 rough > $O/variant_check.log 2>&1; tail -7 $O/variant_check.log | cut -c1-200
timeout 1800 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_parity.py -x -q > $O/pytest.log 2>&1; echo "pytest rc $?"; grep -E "known tie|named tie|passed|failed" $O/pytest.log | cut -c1-400
