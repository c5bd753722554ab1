bash tools/run_ab.sh r02_y "gs2 gs1 gs0" "1.25" 1024 smooth
bash tools/run_ab.sh r02_y "gs2 gs1 gs0" "1.25" 256 rough
bash tools/run_ab.sh r02_y "gs2 gs1 gs0" "1.25" 256 checker
