#!/bin/bash
# round 5, GPU call 9: the final tree -- GPU suite, rocprofv3 trace + counters of the bench command, the two-point bytes experiment (times and bytes), the bench line
O=gpurun_out/r5i; mkdir -p $O
AB=dsurftomo_amd/build/ab
timeout 2400 python3 -m pytest tests -m gpu -q > $O/tests.log 2>&1; echo "tests rc=$?"; tail -4 $O/tests.log | cut -c1-300
DSA_PMC_STEPS=1 timeout 2000 bash tools/profile_bench.sh r5i_prof > $O/profile.log 2>&1; tail -22 gpurun_out/r5i_prof/summary.txt | cut -c1-260
timeout 300 python3 tools/ab_headline.py 1000 smooth stride1_nodetect:exact_ties=0,tie_detect=0 > $O/ab_stride.log 2>&1
DSA_LIB_PATH=$AB/lib_stride2.so timeout 300 python3 tools/ab_headline.py 1000 smooth stride2_nodetect:exact_ties=0,tie_detect=0 >> $O/ab_stride.log 2>&1
cat $O/ab_stride.log
export DSA_AB_REPS=1
timeout 500 bash tools/collect_pmc.sh r5i_bytes1 fetch,write - -- python3 tools/ab_headline.py 1000 smooth x:exact_ties=0,tie_detect=0 > $O/bytes1.log 2>&1
timeout 500 bash tools/collect_pmc.sh r5i_bytes2 fetch,write $AB/lib_stride2.so -- python3 tools/ab_headline.py 1000 smooth x:exact_ties=0,tie_detect=0 > $O/bytes2.log 2>&1
grep -h "k_fim_bundle" gpurun_out/r5i_bytes1/summary.txt gpurun_out/r5i_bytes2/summary.txt | cut -c1-200
unset DSA_AB_REPS
timeout 900 python3 bench.py --steps 5 --warmup 2 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; cut -c1-400 $O/bench.json
