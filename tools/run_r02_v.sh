tools/micro/exact_math_check
bash tools/run_ab.sh r02_v "ocr exact" "1.25" 1024 smooth
bash tools/run_ab.sh r02_v "ocr exact" "1.25" 256 rough
bash tools/run_ab.sh r02_v "ocr exact" "1.25" 256 checker
