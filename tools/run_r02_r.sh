O=gpurun_out/r02_r; mkdir -p $O
DSA_LIB_PATH=dsurftomo_amd/build/ab/lib_ocr.so timeout 600 python3 -m pytest tests -m gpu -x -v -k "parity or fullsize or boundary" > $O/pytest.log 2>&1
grep -nE "PASSED|FAILED|ERROR|Fatal|fault|Memory|Abort" $O/pytest.log | head -40 | cut -c1-220
