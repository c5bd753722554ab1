bash tools/run_ab.sh r02_p "skip oc3" "0.8,1.0,1.25" 1024 smooth
bash tools/run_ab.sh r02_p "skip oc3" "1.25" 256 rough
bash tools/run_ab.sh r02_p "skip oc3" "1.25" 256 checker
