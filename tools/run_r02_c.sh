set -x
O=gpurun_out/r02_c; mkdir -p $O
export TMPDIR=/tmp
python3 tools/perf_probe.py 131 1024 0.3,0.45,0.6,0.8,1.0,1.25,1.6 smooth 128,256 > $O/sweep.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE -d $O/calib_f -o r -- tools/micro/fetch_calib > $O/calib_f.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats -d $O/calib_t -o r -- tools/micro/fetch_calib > $O/calib_t.log 2>&1
python3 tools/rocpd_pmc.py $(find $O/calib_f -name "*.db" | head -1) > $O/calib_summary.txt 2>&1
python3 tools/rocpd_summary.py $(find $O/calib_t -name "*.db" | head -1) >> $O/calib_summary.txt 2>&1
python3 -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1
grep -v "phase share" $O/sweep.log | cut -c1-200; grep -v "^#" $O/calib_summary.txt | cut -c1-150 | head -30; tail -40 $O/pytest.log
