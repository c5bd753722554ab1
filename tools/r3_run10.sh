#!/bin/bash
O=gpurun_out/r3_run10; mkdir -p $O
timeout 1500 python3 tools/exact_probe.py 515 128 checker 768,4000,8000,16000 0 2>&1 | grep -E "fixed point|exact_ties=2" | tee $O/exact_probe_4097.log
timeout 900 python3 tools/exact_probe.py 131 512 checker 768,2048,4096 0 2>&1 | grep -E "fixed point  |exact_ties=2" | tee $O/exact_probe_1025_512.log
