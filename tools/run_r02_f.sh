O=gpurun_out/r02_clk; mkdir -p $O
for l in clk clkoc2; do echo "== $l"; DSA_LIB_PATH=dsurftomo_amd/build/ab/lib_$l.so python3 tools/passa_probe.py 1024 256; done 2>&1 | tee $O/passa.txt
for l in ph phoc2; do echo "== $l"; DSA_LIB_PATH=dsurftomo_amd/build/ab/lib_$l.so python3 tools/perf_probe.py 131 1024 1.25 smooth 256; done 2>&1 | cut -c1-230 | tee $O/phase.txt
