O=gpurun_out/r02_z; mkdir -p $O
python3 tools/perf_probe.py 131 1024 1.0,1.25 smooth 256,320,384 2>&1 | grep -v "phase share" | cut -c1-200 | tee $O/threads.txt
python3 tools/perf_probe.py 131 256 1.25 rough 256,320,384 2>&1 | grep -v "phase share" | cut -c1-200 | tee -a $O/threads.txt
