bash tools/run_ab.sh r02_n "cmp skip" "1.25" 1024 smooth
bash tools/run_ab.sh r02_n "cmp skip" "1.25" 256 rough
