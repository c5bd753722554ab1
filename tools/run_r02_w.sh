O=gpurun_out/r02_w; mkdir -p $O
export TMPDIR=/tmp
python3 -m pytest tests -m gpu -x -q -k "spmv or lsmr or iteration" > $O/pytest.log 2>&1; tail -6 $O/pytest.log | cut -c1-200
timeout 900 rocprofv3 --kernel-trace --stats -d $O/spmv_trace -o r -- python3 tests/tools/headline_boundary.py 8 1000 --spmv --lsmr --device-rows > $O/headline.log 2>&1
python3 tools/rocpd_summary.py $(find $O/spmv_trace -name "*.db" | head -1) > $O/spmv_trace_summary.txt 2>&1; grep -E "spmv|fill_block" $O/spmv_trace_summary.txt | cut -c1-200; grep -E "aprod|LSMR|lsmr|iteration_system|identical|same" $O/headline.log | cut -c1-220
