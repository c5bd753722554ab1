O=gpurun_out/r02_t; mkdir -p $O
python3 tools/perf_probe.py 131 1024 1.0,1.25,1.6,2.0 smooth 256,512,128 2>&1 | grep -v "phase share" | cut -c1-200 | tee $O/threads.txt
