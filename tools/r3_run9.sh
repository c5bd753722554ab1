#!/bin/bash
O=gpurun_out/r3_run9; mkdir -p $O
DSA_BENCH_FORCE_DIST=1 timeout 900 python3 bench.py --steps 2 --warmup 1 > $O/bench_rccl_one_rank.log 2> $O/bench_rccl.err; echo "rc $?"; tail -1 $O/bench_rccl_one_rank.log | cut -c1-700; tail -5 $O/bench_rccl.err | cut -c1-300
timeout 900 python3 -m pytest tests/test_gpu_sharded.py -x -q > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -3 $O/pytest.log | cut -c1-300
