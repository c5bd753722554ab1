bash tools/run_ab.sh r02_act "base act2 w5 act2oc2" "1.25" 1024 smooth
bash tools/run_ab.sh r02_act "base act2" "1.25" 256 rough
