#!/bin/bash
# round 3: the committed measurements (pytest -m gpu, rocprofv3 kernel trace + pmc passes of the bench command, bench line, dispersion counters)
bash tools/final_round.sh r03_final 2>&1 | cut -c1-400
