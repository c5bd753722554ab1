#!/bin/bash
O=gpurun_out/r3_run6; mkdir -p $O
DSA_LIB_PATH=dsurftomo_amd/build/ab/lib_ledger.so timeout 600 python3 tools/ledger_probe.py 131 512 smooth > $O/ledger_counters.json 2> $O/ledger.err; cat $O/ledger_counters.json | cut -c1-1200; tail -2 $O/ledger.err
bash tools/run_ab.sh r3_run6/ab "km0 one km0 one" 1.25 8192 smooth 2>&1 | cut -c1-200
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -x -q -k "not config4 and not rough_known" > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -3 $O/pytest.log | cut -c1-300
