#!/bin/bash
O=gpurun_out/r03_final2; mkdir -p $O
timeout 2400 python3 -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -3 $O/pytest.log | cut -c1-200
bash tools/collect_pmc.sh r03_final2_disp flops64,busy - -- python3 tools/disp_roofline.py 1 > $O/disp_pmc.log 2>&1
python3 tools/pmc_to_json.py --dispersion gpurun_out/r03_final2_disp 15101680 gpurun_out/r03_final2_disp/pmc_dispersion.json >> $O/disp_pmc.log 2>&1; tail -12 $O/disp_pmc.log
timeout 600 python3 tests/tools/taipei_probe.py > $O/taipei_probe.log 2>&1; tail -12 $O/taipei_probe.log | cut -c1-250
timeout 900 python3 tests/tools/headline_boundary.py > $O/headline_boundary.log 2>&1; tail -14 $O/headline_boundary.log | cut -c1-250
