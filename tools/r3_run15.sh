#!/bin/bash
O=gpurun_out/r3_run15; mkdir -p $O
bash tools/collect_pmc.sh r3_run15_disp flops64,busy - -- python3 tools/disp_roofline.py 1 > $O/disp_pmc.log 2>&1
python3 tools/pmc_to_json.py --dispersion gpurun_out/r3_run15_disp 15101680 gpurun_out/r3_run15_disp/pmc_dispersion.json >> $O/disp_pmc.log 2>&1; tail -11 $O/disp_pmc.log
cp gpurun_out/r3_run15_disp/pmc_dispersion.json profiles/pmc_dispersion.json
python3 bench.py --steps 3 --warmup 1 > $O/bench.log 2> $O/bench.err; tail -1 $O/bench.log | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['traffic'], d['secondary']['dispersion']['roofline'])"
