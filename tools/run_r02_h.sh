O=gpurun_out/r02_h; mkdir -p $O
export TMPDIR=/tmp
rocprofv3 -L > $O/counters.txt 2>&1
grep -iE "F64|TRANS|FLOP" $O/counters.txt | head -40
python3 tools/disp_roofline.py 2 > $O/disp.log 2>&1; cat $O/disp.log
bash tools/collect_pmc.sh r02_h/disp_pmc fp64,busy,insts dummy -- python3 tools/disp_roofline.py 1 > $O/disp_pmc.log 2>&1; grep -E "k_disp|k_depth|rc=" gpurun_out/r02_h/disp_pmc/summary.txt | cut -c1-200
timeout 1200 python3 tests/tools/headline_boundary.py 8 1000 --spmv --lsmr --device-rows > $O/headline.log 2>&1; cat $O/headline.log
