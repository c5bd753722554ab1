O=gpurun_out/r02_l; mkdir -p $O
python3 -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; tail -8 $O/pytest.log | cut -c1-200
bash tools/run_ab.sh r02_l "act2 cmp" "1.0,1.25" 1024 smooth
bash tools/run_ab.sh r02_l "act2 cmp" "1.25" 256 rough
bash tools/run_ab.sh r02_l "act2 cmp" "1.25" 256 checker
