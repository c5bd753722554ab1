#!/bin/bash
# round 5, GPU call 16: (a) the three fuzz differences of seed 101 -- with the committed library's divisions too? (b) rays: four lanes per ray and
# shared reciprocals against the committed kernel, same box; (c) the GPU tests that compare rays and rows with the oracle
O=gpurun_out/r5p; mkdir -p $O
DSA_LIB_PATH=dsurftomo_amd/build/ab/lib_r5head.so timeout 900 python3 tests/tools/fuzz_dispersion.py 60 101 > $O/fuzz_head.log 2>&1; echo "fuzz(head) rc=$?"; grep "<<<\|combinations" $O/fuzz_head.log | cut -c1-200
for cfg in "131 256 32" "131 1024 32" "131 2048 32" "16 14 30"; do
  set -- $cfg
  for v in "head:dsurftomo_amd/build/ab/lib_r5head.so:" "plain1:dsurftomo_amd/build/ab/lib_rayplain.so:1" "plain4:dsurftomo_amd/build/ab/lib_rayplain.so:4" "new1::1" "new4::4" "auto::"; do
    IFS=: read name lib lanes <<< "$v"
    echo "== $cfg $name" >> $O/rays.log
    DSA_LIB_PATH=$lib DSA_RAY_LANES=$lanes timeout 300 python3 tools/rays_probe.py $1 $2 $3 2>&1 | grep "pass 1\|rays/s" | cut -c1-260 >> $O/rays.log
  done
done
cat $O/rays.log
timeout 1200 python3 -m pytest tests/test_gpu_boundary.py tests/test_gpu_parity.py -m gpu -q -x > $O/tests.log 2>&1; echo "tests rc=$?"; tail -3 $O/tests.log | cut -c1-200
timeout 300 python3 tests/tools/taipei_probe.py --no-ref > $O/taipei.log 2>&1; echo "taipei rc=$?"; grep "stages" $O/taipei.log | cut -c1-260
