O=gpurun_out/r02_u; mkdir -p $O
python3 -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; tail -4 $O/pytest.log | cut -c1-200
bash tools/profile_bench.sh r02_final > $O/profile.log 2>&1; tail -14 $O/profile.log | cut -c1-200
cp gpurun_out/r02_final/pmc_latest.json profiles/pmc_latest.json
python3 bench.py --steps 5 --warmup 2 > $O/bench.log 2>&1; cat $O/bench.log | tail -1
