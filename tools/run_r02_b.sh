set -x
O=gpurun_out/r02_b; mkdir -p $O
python3 tests/tools/parity_table.py > $O/parity_table.log 2>&1
export TMPDIR=/tmp
timeout 300 rocprofv3 --pmc FETCH_SIZE -d $O/calib_f -o r -- tools/micro/fetch_calib > $O/calib_f.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE -d $O/calib_w -o r -- tools/micro/fetch_calib > $O/calib_w.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats -d $O/calib_t -o r -- tools/micro/fetch_calib > $O/calib_t.log 2>&1
for n in f w; do python3 tools/rocpd_pmc.py $(find $O/calib_$n -name "*.db" | head -1); done > $O/calib_summary.txt 2>&1
python3 tools/rocpd_summary.py $(find $O/calib_t -name "*.db" | head -1) >> $O/calib_summary.txt 2>&1
python3 bench.py --steps 2 --warmup 1 > $O/bench1.log 2>&1
timeout 600 python3 bench.py --gpus 2 --steps 2 --warmup 1 > $O/bench2.log 2>&1
tail -5 $O/parity_table.log; cat $O/calib_summary.txt | head -40; cat $O/bench1.log $O/bench2.log | cut -c1-1500
