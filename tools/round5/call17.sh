#!/bin/bash
# round 5, GPU call 17: where one lane per ray overtakes four (larger launches), with the final ray kernels; Love's shared reciprocal against the oracle
O=gpurun_out/r5q; mkdir -p $O
for cfg in "131 256 32 9" "131 2048 32 3" "131 4096 32 3" "131 8192 32 3"; do
  set -- $cfg
  for lanes in 1 4; do
    echo "== $cfg lanes $lanes" >> $O/rays.log
    DSA_RAY_LANES=$lanes timeout 600 python3 tools/rays_probe.py $1 $2 $3 $4 2>&1 | grep "pass 1\|Error\|error" | cut -c1-260 >> $O/rays.log
  done
done
cat $O/rays.log
timeout 900 python3 tests/tools/fuzz_dispersion.py 40 23 > $O/fuzz.log 2>&1; echo "fuzz rc=$?"; tail -1 $O/fuzz.log
timeout 1200 python3 -m pytest tests/test_gpu_boundary.py -m gpu -q -x > $O/tests.log 2>&1; echo "tests rc=$?"; tail -2 $O/tests.log | cut -c1-200
