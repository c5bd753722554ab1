#!/bin/bash
O=gpurun_out/r03_final4; mkdir -p $O
# exact mode against the oracle on random sources (some on nodes, some hugging an edge): every field must be bit-identical
DSA_EXACT=2 timeout 1500 python3 tests/tools/fuzz_parity.py 48 3 > $O/fuzz_exact_small.log 2>&1; grep -E "^nx|worst" $O/fuzz_exact_small.log | cut -c1-220
DSA_EXACT=2 timeout 900 python3 tests/tools/fuzz_parity.py 24 5 131:rough:8 131:checker:8 > $O/fuzz_exact_1025.log 2>&1; grep -E "^nx|worst" $O/fuzz_exact_1025.log | cut -c1-220
# the default mode on the same random sources, for the record
timeout 900 python3 tests/tools/fuzz_parity.py 24 5 131:rough:8 131:checker:8 > $O/fuzz_default_1025.log 2>&1; grep -E "^nx|worst" $O/fuzz_default_1025.log | cut -c1-220
# determinism of the final kernels
for c in "35 checker4 8 100" "35 rough 8 100" "18 homog 8 100"; do timeout 600 python3 tools/stress_determinism.py $c 2>&1 | tail -1; done | tee $O/determinism.log
timeout 600 python3 -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc $?"; tail -4 $O/smoke.log | cut -c1-200
timeout 2400 python3 -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -3 $O/pytest.log | cut -c1-200
