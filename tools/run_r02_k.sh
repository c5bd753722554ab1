O=gpurun_out/r02_k; mkdir -p $O
for pad in 0 16384 30720 61440; do echo "== lds pad $pad"; DSA_LDS_PAD=$pad python3 tools/perf_probe.py 131 1024 1.25 smooth 256 2>&1 | grep -v "phase share" | cut -c1-200; done | tee $O/occupancy_scaling.txt
