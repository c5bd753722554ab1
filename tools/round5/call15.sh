#!/bin/bash
# round 5, GPU call 15: the Rayleigh secular function with shared-reciprocal divisions -- same-box A/B against the compiler's division, fuzz against the oracle, stage tests
O=gpurun_out/r5o; mkdir -p $O
for i in 1 2; do
  DSA_LIB_PATH=dsurftomo_amd/build/ab/lib_plaindiv.so timeout 300 python3 tools/disp_roofline.py 2>&1 | tail -1 | cut -c1-400 | sed 's/^/plain division:  /' >> $O/ab.log
  timeout 300 python3 tools/disp_roofline.py 2>&1 | tail -1 | cut -c1-400 | sed 's/^/shared reciprocal: /' >> $O/ab.log
done
cat $O/ab.log
timeout 900 python3 tests/tools/fuzz_dispersion.py 60 23 > $O/fuzz.log 2>&1; echo "fuzz rc=$?"; tail -2 $O/fuzz.log
timeout 900 python3 tests/tools/fuzz_dispersion.py 60 101 > $O/fuzz2.log 2>&1; echo "fuzz2 rc=$?"; tail -2 $O/fuzz2.log
timeout 900 python3 -m pytest tests/test_gpu_boundary.py tests/test_gpu_errors.py -m gpu -q -x > $O/tests.log 2>&1; echo "tests rc=$?"; tail -3 $O/tests.log | cut -c1-200
timeout 300 python3 tests/tools/taipei_probe.py --no-ref > $O/taipei.log 2>&1; echo "taipei rc=$?"; tail -12 $O/taipei.log | cut -c1-200
DSA_LIB_PATH=dsurftomo_amd/build/ab/lib_plaindiv.so timeout 300 python3 tests/tools/taipei_probe.py --no-ref > $O/taipei_plain.log 2>&1; tail -12 $O/taipei_plain.log | cut -c1-200
timeout 600 bash tools/collect_pmc.sh r5o_dpmc fp64,insts,busy - -- python3 tools/disp_roofline.py > $O/dpmc.log 2>&1; grep "k_dispersion<2>" gpurun_out/r5o_dpmc/summary.txt | cut -c1-110
