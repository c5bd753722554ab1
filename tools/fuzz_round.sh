O=gpurun_out/${1:-r02_fuzz}; mkdir -p $O
python3 bench.py --steps 5 --warmup 2 > $O/bench.log 2>&1; tail -1 $O/bench.log | cut -c1-300
for seed in 11 12 13; do python3 tests/tools/fuzz_parity.py 128 $seed; done > $O/fuzz_parity.log 2>&1; grep -E "^nx|worst" $O/fuzz_parity.log | cut -c1-230
python3 tests/tools/fuzz_boundary.py 30 21 > $O/fuzz_boundary.log 2>&1; tail -4 $O/fuzz_boundary.log | cut -c1-200
python3 tests/tools/fuzz_parity.py 48 5 131:smooth:8 131:rough:8 >> $O/fuzz_parity.log 2>&1; tail -3 $O/fuzz_parity.log | cut -c1-230
