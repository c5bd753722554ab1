export TMPDIR=/tmp
bash tools/collect_pmc.sh r02_o/skip insts,busy dsurftomo_amd/build/ab/lib_skip.so > gpurun_out/r02_o_skip.log 2>&1
bash tools/collect_pmc.sh r02_o/solve2 insts,busy dsurftomo_amd/build/ab/lib_solve2.so > gpurun_out/r02_o_solve2.log 2>&1
grep -E "solves/s|SQ_INSTS_VALU |SQ_INSTS_SALU|SQ_ACTIVE_INST_VALU|SQ_BUSY_CYCLES" gpurun_out/r02_o/skip/summary.txt | grep -v "128, false" | cut -c1-180
grep -E "solves/s|SQ_INSTS_VALU |SQ_INSTS_SALU|SQ_ACTIVE_INST_VALU|SQ_BUSY_CYCLES" gpurun_out/r02_o/solve2/summary.txt | grep -v "128, false" | cut -c1-180
