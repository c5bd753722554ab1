#!/bin/bash
# round 5, GPU call 13: the GPU suite and the bench line once more on the final tree (the counters of profiles/pmc_latest.json now match it)
O=gpurun_out/r5m; mkdir -p $O
timeout 2400 python3 -m pytest tests -m gpu -q > $O/tests.log 2>&1; echo "tests rc=$?"; tail -3 $O/tests.log | cut -c1-300
timeout 900 python3 bench.py --steps 5 --warmup 2 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; cut -c1-300 $O/bench.json
python3 -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -3 $O/smoke.log | cut -c1-200
