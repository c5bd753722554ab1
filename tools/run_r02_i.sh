O=gpurun_out/r02_i; mkdir -p $O
export TMPDIR=/tmp
bash tools/collect_pmc.sh r02_i/disp_pmc fp64,busy,insts - -- python3 tools/disp_roofline.py 1 > $O/disp_pmc.log 2>&1; grep -E "k_disp|k_depth|rc=" gpurun_out/r02_i/disp_pmc/summary.txt | cut -c1-200
timeout 900 rocprofv3 --kernel-trace --stats -d $O/spmv_trace -o r -- python3 tests/tools/headline_boundary.py 8 1000 --spmv > $O/spmv_trace.log 2>&1
python3 tools/rocpd_summary.py $(find $O/spmv_trace -name "*.db" | head -1) > $O/spmv_trace_summary.txt 2>&1; head -30 $O/spmv_trace_summary.txt | cut -c1-200
timeout 1200 python3 tests/tools/config4_probe.py 512 8 2 > $O/config4.log 2>&1; cat $O/config4.log
