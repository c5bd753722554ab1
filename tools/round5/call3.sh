#!/bin/bash
# round 5, GPU call 3: census cost after the unroll, the bench command's trace and counters, byte counts of the two-point experiment, the full bench line
O=gpurun_out/r5c; mkdir -p $O
AB=dsurftomo_amd/build/ab
timeout 500 python3 tools/ab_headline.py 1000 smooth mode0_nodetect:exact_ties=0,tie_detect=0 default: halves_default:bundle_tail=0 > $O/ab_census.log 2>&1
cat $O/ab_census.log
DSA_PMC_STEPS=1 timeout 1500 bash tools/profile_bench.sh r5c_prof > $O/profile.log 2>&1
tail -25 gpurun_out/r5c_prof/summary.txt | cut -c1-300
export DSA_AB_REPS=1
timeout 600 bash tools/collect_pmc.sh r5c_bytes_default fetch,write - -- python3 tools/ab_headline.py 1000 smooth x:exact_ties=0,tie_detect=0 > $O/bytes_default.log 2>&1
timeout 600 bash tools/collect_pmc.sh r5c_bytes_stride2 fetch,write $AB/lib_stride2.so -- python3 tools/ab_headline.py 1000 smooth x:exact_ties=0,tie_detect=0 > $O/bytes_stride2.log 2>&1
grep -h "k_fim_bundle\|solves/s" gpurun_out/r5c_bytes_default/summary.txt gpurun_out/r5c_bytes_stride2/summary.txt | cut -c1-220
timeout 900 python3 bench.py > $O/bench.json 2> $O/bench.err
echo "bench rc=$?"; cut -c1-1500 $O/bench.json; tail -3 $O/bench.err
