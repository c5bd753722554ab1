bash tools/run_ab.sh r02_oc "base oc1 oc2" "0.8,1.0,1.25,1.6" 1024 smooth
bash tools/run_ab.sh r02_oc "base oc1 oc2" "1.25" 256 rough
