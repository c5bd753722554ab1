#!/bin/bash
# What the round's committed measurements come from (GPU box, repo root): the GPU test suite, the rocprofv3 kernel trace and --pmc passes
# of the bench command, and one plain bench run.  Copy gpurun_out/<tag>/{summary.txt -> profiles/rNN_rocprof_bench.txt, pmc_latest.json ->
# profiles/pmc_latest.json}, gpurun_out/parity_report.txt and the bench line into profiles/ afterwards.
TAG=${1:-final}
O=gpurun_out/$TAG; mkdir -p $O
python3 -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; tail -3 $O/pytest.log | cut -c1-200
bash tools/profile_bench.sh $TAG > $O/profile.log 2>&1; tail -14 $O/profile.log | cut -c1-200
python3 bench.py --steps 5 --warmup 2 > $O/bench.log 2>&1; tail -1 $O/bench.log
# the dispersion kernel's FP64 counters (secondary block of the bench line): copy gpurun_out/<tag>_disp/pmc_dispersion.json -> profiles/
bash tools/collect_pmc.sh ${TAG}_disp flops64,busy - -- python3 tools/disp_roofline.py 1 > $O/disp_pmc.log 2>&1
python3 tools/pmc_to_json.py --dispersion gpurun_out/${TAG}_disp 15101680 gpurun_out/${TAG}_disp/pmc_dispersion.json >> $O/disp_pmc.log 2>&1; tail -12 $O/disp_pmc.log
