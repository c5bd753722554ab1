set -x
mkdir -p gpurun_out/r02_base
python3 -m pytest tests -m gpu -x -q > gpurun_out/r02_base/pytest.log 2>&1; tail -3 gpurun_out/r02_base/pytest.log
DSA_FIM_VARIANTS=1,0 python3 tools/perf_probe.py 131 1024 1.25 smooth 256 > gpurun_out/r02_base/probe_variants.log 2>&1
DSA_LIB_PATH=dsurftomo_amd/build/ab/lib_clocks.so python3 tools/passa_probe.py 1024 256 > gpurun_out/r02_base/passa.log 2>&1
python3 tools/perf_probe.py 131 128 1.25 smooth 256 > gpurun_out/r02_base/probe_128units.log 2>&1
python3 bench.py --steps 3 --warmup 1 > gpurun_out/r02_base/bench.log 2>&1
cat gpurun_out/r02_base/*.log | tail -40
