O=gpurun_out/r02_m; mkdir -p $O
python3 bench.py --steps 3 --warmup 1 > $O/bench.log 2>&1; cut -c1-400 $O/bench.log | tail -2
for pad in 0 16384 30720; do echo "== lds pad $pad"; DSA_LDS_PAD=$pad python3 tools/perf_probe.py 131 1024 1.25 smooth 256 2>&1 | grep -v "phase share" | cut -c1-200; done | tee $O/occupancy_scaling.txt
bash tools/run_ab.sh r02_m "cmp cmpoc1 cmpoc2" "0.8,1.0,1.25,1.6" 1024 smooth
