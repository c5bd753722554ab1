#!/bin/bash
O=gpurun_out/r3_run8; mkdir -p $O
for sz in "131 9 16" "18 9 26"; do
  DSA_LIB_PATH=dsurftomo_amd/build/ab/lib_dold.so DSA_DISP_SAVE=$O/ref_ timeout 600 python3 tools/disp_probe.py $sz 2>&1 | sed 's/^/old: /'
  DSA_LIB_PATH=dsurftomo_amd/build/ab/lib_dnew.so DSA_DISP_COMPARE=$O/ref_ timeout 600 python3 tools/disp_probe.py $sz 2>&1 | sed 's/^/new: /'
done 2>&1 | tee $O/disp_ab.txt
timeout 1800 python -m pytest tests/test_gpu_boundary.py tests/test_gpu_lsmr.py -x -q > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -4 $O/pytest.log | cut -c1-250
