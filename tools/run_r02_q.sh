bash tools/run_ab.sh r02_q "skip ocr" "0.8,1.0,1.25" 1024 smooth
bash tools/run_ab.sh r02_q "skip ocr" "1.25" 256 rough
bash tools/run_ab.sh r02_q "skip ocr" "1.25" 256 checker
DSA_LIB_PATH=dsurftomo_amd/build/ab/lib_ocr.so python3 -m pytest tests -m gpu -x -q -k "parity or fullsize or boundary" 2>&1 | tail -5 | cut -c1-200
