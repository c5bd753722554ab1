#!/bin/bash
O=gpurun_out/r03_final3; mkdir -p $O
bash tools/profile_bench.sh r03_final3 > $O/profile.log 2>&1; tail -16 $O/profile.log | cut -c1-200
cp gpurun_out/r03_final3/pmc_latest.json profiles/pmc_latest.json
python3 bench.py --steps 5 --warmup 2 > $O/bench.log 2> $O/bench.err; tail -1 $O/bench.log | cut -c1-1800
timeout 900 python3 tests/tools/headline_boundary.py > $O/headline_boundary.log 2>&1; tail -14 $O/headline_boundary.log | cut -c1-250
timeout 600 python3 tools/exact_probe.py 131 4096 checker 768 0 > $O/exact_probe.log 2>&1; grep -E "fixed point|exact_ties=2" $O/exact_probe.log
timeout 1200 python3 -m pytest tests/test_gpu_exact.py tests/test_gpu_boundary.py tests/test_gpu_lsmr.py -x -q > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -2 $O/pytest.log | cut -c1-200
