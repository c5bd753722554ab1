#!/bin/bash
# round 5, GPU call 6: is the pipeline's second stream on a hardware queue of its own?
O=gpurun_out/r5f; mkdir -p $O
export DSA_DEBUG_PIPE=1 DSA_AB_REPS=2
timeout 400 python3 tools/ab_headline.py 1000 smooth nopipe:exact_ties=0,tie_detect=0,bundle_pipeline=0 pipe_prio:exact_ties=0,tie_detect=0 pipe_noprio:exact_ties=0,tie_detect=0,stream2_priority=0 > $O/ab_pipe.log 2>&1
GPU_MAX_HW_QUEUES=8 timeout 400 python3 tools/ab_headline.py 1000 smooth pipe_prio_8q:exact_ties=0,tie_detect=0 pipe_noprio_8q:exact_ties=0,tie_detect=0,stream2_priority=0 >> $O/ab_pipe.log 2>&1
timeout 300 python3 tools/ab_headline.py 700 smooth nopipe:exact_ties=0,tie_detect=0,bundle_pipeline=0 pipe_prio:exact_ties=0,tie_detect=0 >> $O/ab_pipe.log 2>&1
cat $O/ab_pipe.log
