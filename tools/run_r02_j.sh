O=gpurun_out/r02_j; mkdir -p $O
export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $O/spmv_trace -o r -- python3 tests/tools/headline_boundary.py 8 1000 --spmv > $O/spmv_trace.log 2>&1
python3 tools/rocpd_summary.py $(find $O/spmv_trace -name "*.db" | head -1) > $O/spmv_trace_summary.txt 2>&1; grep -E "spmv|fill_slices" $O/spmv_trace_summary.txt | cut -c1-200; grep "aprod" $O/spmv_trace.log
bash tools/collect_pmc.sh r02_j/disp_pmc flops64 - -- python3 tools/disp_roofline.py 1 > $O/disp_pmc.log 2>&1; grep -E "k_disp|rc=" gpurun_out/r02_j/disp_pmc/summary.txt | cut -c1-200
DSA_LIB_PATH=dsurftomo_amd/build/ab/lib_clk.so python3 tools/passa_probe.py 1024 256 2>&1 | tee $O/passa.txt
